#!/bin/bash
# Round-4 evidence set in one gpurun call: the r0N_final_* set of the fp64 bench command (tools/collect_profiles.sh), the
# counter set of the mixed-precision command (tools/pmc_collect.sh), kernel trace of the n = 4096 step, mid-size step rates.
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
tools/collect_profiles.sh r04final || exit 1
echo "== fp64 set done"
tools/pmc_collect.sh mixed "--mode mixed" > gpurun_out/pmc_mixed.log 2>&1 || { tail -5 gpurun_out/pmc_mixed.log; exit 1; }
echo "== mixed set done"
python3 bench.py --mode mixed --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r04_mixed_bench.json 2> gpurun_out/r04_mixed_bench.err || exit 1
SMALL_N=4096 tools/small_n_trace.sh > gpurun_out/r04_small_n_trace_4096.txt 2>&1 || exit 1
python3 tools/mid_n_rate.py > gpurun_out/r04_mid_n_rates.txt 2>&1 || exit 1
python3 tools/n_sweep_rate.py 4096 6144 8192 12288 16384 24576 32768 49152 65536 98304 131072 262144 > gpurun_out/r04_n_sweep.txt 2>&1 || exit 1
tail -3 gpurun_out/r04_n_sweep.txt
