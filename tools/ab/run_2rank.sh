#!/bin/bash
# repeat the 2-rank shared-GPU bench: tools/ab/run_2rank.sh <runs> [lib letter] [extra bench args...]
R=/root/repo; cd $R
N=${1:-12}; L=$2; shift; shift
[ -n "$L" ] && [ "$L" != tree ] && export MI355PT_LIB=$R/tools/ab/lib$L.so
bad=0
for i in $(seq 1 $N); do
  timeout 200 python bench.py --gpus 2 --share-gpu --steps 64 --warmup 64 --no-4k --steady-ms 100 "$@" > /tmp/b2.out 2> /tmp/b2.err; rc=$?
  [ $rc != 0 ] && { bad=$((bad+1)); echo "  run $i rc=$rc: $(grep -h 'NativeError\|Error' /tmp/b2.err | head -2 | cut -c1-160)"; }
done
echo "lib${L:-tree} [$*]: $bad of $N runs failed"
