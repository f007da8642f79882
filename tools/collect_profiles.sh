#!/bin/bash
# Collects the judged profiles of bench.py on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats      -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE            -> HBM read traffic   } separate passes, --kernel-trace only,
#   3. rocprofv3 --pmc WRITE_SIZE            -> HBM write traffic  } as MI355X_MICROARCH.md prescribes
#   4. rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
#                                            -> VALU issue evidence of the fp64-bound pair kernel
#   5./6. rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 | SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU ...
#                                            -> executed fp64 instruction mix, LDS bank conflicts
# and summarises them into gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/); the bench line
# itself (bench.json) is produced last, after the summaries.
set -eo pipefail
TAG=${1:-final}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-liquid"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_valu" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pmc_valu.log"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_mix_fp64" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pmc_mix_fp64.log"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/pmc_mix_lds" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/pmc_mix_lds.log"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.log" 2>&1
# the bench line quotes roofline.traffic / valu_issue_frac / the rocprof K1 time from the committed summaries: put THIS
# session's summaries in place (the box's copy of the tree) before the line is produced, so that all of it is one session
PFX=${TAG%final}
if [ -n "$PFX" ] && [ "$PFX" != "$TAG" ]; then
  cp "$OUT/pmc_hbm_traffic.json" "profiles/${PFX}_final_pmc_hbm_traffic.json"
  cp "$OUT/pmc_valu.json" "profiles/${PFX}_final_pmc_valu.json"
  cp "$OUT/kernel_stats.csv" "profiles/${PFX}_final_kernel_stats.csv"
  cp "$OUT/pmc_instruction_mix.json" "profiles/${PFX}_final_pmc_instruction_mix.json"
fi
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -5 "$OUT/summary.log"
