set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/round2
mkdir -p "$OUT"; rm -rf "$OUT/mixed_stats" "$OUT/mixed_pmc_valu"
python3 bench.py --mode mixed --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/mixed_bench.json" 2> "$OUT/mixed_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mixed_stats" -o run -- python3 bench.py --mode mixed --steps 20 --warmup 3 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/mixed_stats.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mixed_pmc_valu" -o run -- python3 bench.py --mode mixed --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> "$OUT/mixed_pmc_valu.log"
python3 -c "import json; m=json.load(open('$OUT/mixed_bench.json')); print('mixed', m['value'], m['steps_per_s_liquid'], m['roofline']['kernel_ms_avg'])"
