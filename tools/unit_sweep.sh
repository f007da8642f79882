#!/bin/bash
# tools/unit_sweep.py over the middle sizes: which (tiles per row group, parts per pass, work items) fill the card
R=${GRAFT_REPO_ROOT:-$PWD}
for n in ${SIZES:-4096 8192 16384 32768 65536}; do
  python3 $R/tools/unit_sweep.py $n ${COMBOS:-0,0,0 1,0,16384 2,0,16384 4,0,16384 4,0,32768 4,0,65536 2,0,32768}
done
