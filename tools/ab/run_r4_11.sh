#!/bin/bash
cd /root/repo
echo "== resume in a loop inside one iteration (multisample kernel)"; timeout 200 tools/handover_stress.bin tools/ab/libM2.so 1200 304 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -3 | cut -c1-400
