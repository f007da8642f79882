#!/bin/bash
# rocprofv3 kernel stats + SQ counter passes of one bench.py command -> gpurun_out/pmc_<tag>/{kernel_stats.csv,pmc.json}
#   tools/pmc_collect.sh TAG "<bench.py arguments>"        e.g.  tools/pmc_collect.sh mixed "--mode mixed"
# Counters go in passes of their own with --kernel-trace only (the pool refuses --pmc beside the trace domains).
set -eo pipefail
TAG=$1; ARGS=$2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py $ARGS --steps 20 --warmup 3 --no-cpu-baseline --no-liquid > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "stats pass done"
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/pass$i" -o run -- python3 bench.py $ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pass$i.log"
  echo "pass $i done: $counters"
done <<'C'
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32
SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VSKIPPED
SQ_IFETCH SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES
FETCH_SIZE
WRITE_SIZE
C
python3 tools/pmc_generic.py "$OUT" "$ARGS"
