#!/bin/bash
# tools/unit_sweep.py over the middle sizes: which (tiles per row group, parts per pass, work items) fill the card
R=${GRAFT_REPO_ROOT:-$PWD}
for n in ${SIZES:-4096 8192 16384 32768 65536}; do
  python3 $R/tools/unit_sweep.py $n ${COMBOS:-0,0,0 1,1,16384 1,2,16384 2,1,16384 2,2,16384 2,4,16384 4,1,16384 4,2,16384 4,4,16384 4,1,65536 4,2,65536 2,2,65536 4,1,8192 4,2,8192 2,2,8192}
done
