#!/bin/bash
# full -m gpu suite under non-default engine knobs (robustness sweep; not part of the judged tests).
# Known, understood failures: LJMD_SORT_KD=0 (Morton order: looser tiles, the all-fp32 extreme of the mixed mode leaves its
# k-d-calibrated bound) and LJMD_N3_TARGET_WAVES=4096 (32x longer sequential energy sums per wave: d_epot 1.2e-11
# against the oracle, bound 1e-12).
mkdir -p gpurun_out/cross
for cfg in ${CONFIGS:-"LJMD_N3_WG_WAVES=4" "LJMD_N3_XCD_REMAP=0" "LJMD_N3_ROW_TILES=2" "LJMD_RESORT_EVERY=1" "LJMD_SORT_KD=0" "LJMD_N3_TARGET_WAVES=4096"}; do
  name=$(echo $cfg | tr '=' '_')
  env $cfg timeout -k 10 500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_parity.py::test_one_million_particles_single_gpu_indexing > gpurun_out/cross/$name.log 2>&1
  echo "$cfg -> rc $? : $(tail -1 gpurun_out/cross/$name.log)"
done
