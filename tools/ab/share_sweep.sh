#!/bin/bash
# small-share sweep on one GPU: tools/ab/share_sweep.sh "<tune list>" ... (EMULATE_TUNE syntax, e.g. "frame_group=4,queue_chunk=4")
cd /root/repo
for t in "$@"; do
  for w in 8 4; do
    EMULATE_ONLY=1920x1080:$w EMULATE_TUNE="$t" python tools/emulate_strong.py /tmp/es.json 2>/dev/null | sed "s/^/[$t] /"
  done
done
