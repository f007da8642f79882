#!/bin/bash
# full -m gpu suite under non-default engine knobs (robustness sweep; not part of the judged tests).
# Known, understood failures are listed in profiles/rNN_cross_config_tests.txt.
mkdir -p gpurun_out/cross
for cfg in ${CONFIGS:-"LJMD_N3_WG_WAVES=4" "LJMD_N3_XCD_REMAP=0" "LJMD_N3_ROW_TILES=2" "LJMD_RESORT_EVERY=1" "LJMD_N3_BOTH_TIES=0" "LJMD_FP32_VFAR=0" "LJMD_N3_TARGET_WAVES=4096" "LJMD_N3_CLUSTERS=0" "LJMD_N3_PERTILE=0" "LJMD_FUSE_TAIL=0" "LJMD_MULTI_THREADS=0" "LJMD_MIGRATE_DEAL=blocks" "LJMD_OVERLAP_EXCHANGE=0"}; do
  name=$(echo $cfg | tr '=' '_')
  env $cfg timeout -k 10 400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_parity.py::test_one_million_particles_single_gpu_indexing --deselect tests/test_gpu_sharded.py::test_config4_sharded_eight_ranks_n1048576 > gpurun_out/cross/$name.log 2>&1
  echo "$cfg -> rc $? : $(tail -1 gpurun_out/cross/$name.log)"
done
