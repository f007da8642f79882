#!/bin/bash
# Round-2 evidence in one GPU call (run through gpurun from the repo root): the judged profiles of the bench command
# (tools/collect_profiles.sh), the same for the mixed-precision mode (BASELINE config 5), the HBM traffic of the
# 4-wave-workgroup variant of the pair kernel, and one pass over BASELINE configs 2-5.
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/round2
mkdir -p "$OUT"
bash tools/collect_profiles.sh r02final > "$OUT/collect_final.log" 2>&1
echo "[round2] final profiles done"
# the same command WITH the liquid and sampled-segment legs: the forces-only instantiation pair_n3_kernel<3, 4, 1, false>
# shows up as its own row beside <3, 4, 1, true>
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sampled_stats" -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/sampled_bench_under_rocprof.json" 2> "$OUT/sampled_stats.log"
echo "[round2] sampled-segment stats done"
python3 bench.py --mode mixed --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/mixed_bench.json" 2> "$OUT/mixed_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mixed_stats" -o run -- python3 bench.py --mode mixed --steps 20 --warmup 3 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/mixed_stats.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mixed_pmc_valu" -o run -- python3 bench.py --mode mixed --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/mixed_pmc_valu.log"
echo "[round2] mixed mode done"
export LJMD_N3_WG_WAVES=4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/wg4_pmc_fetch" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > "$OUT/wg4_bench.json" 2> "$OUT/wg4_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/wg4_pmc_write" -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/wg4_write.log"
unset LJMD_N3_WG_WAVES
echo "[round2] 4-wave workgroup traffic done"
python3 tools/run_all_configs.py > "$OUT/all_configs.jsonl" 2> "$OUT/all_configs.err"
echo "[round2] all configs done"
tail -3 "$OUT/all_configs.jsonl" | cut -c1-600
