#!/bin/bash
# Copies the summaries of tools/collect_round2.sh (gpurun_out/, scratch) into profiles/ (tracked), named r02_*.
set -eo pipefail
F=gpurun_out/profiles_r02final
R=gpurun_out/round2
cp $F/bench.json profiles/r02_final_bench.json
cp $F/bench_under_rocprof.json profiles/r02_final_bench_under_rocprof.json
cp $F/kernel_stats.csv profiles/r02_final_kernel_stats.csv
cp $F/pmc_hbm_traffic.json profiles/r02_final_pmc_hbm_traffic.json
cp $F/pmc_valu.json profiles/r02_final_pmc_valu.json
cp $F/pmc_instruction_mix.json profiles/r02_final_pmc_instruction_mix.json
# mixed mode: its own --stats and VALU passes (pmc_summary.py wants the traffic passes too: the fp64 ones stand in and
# their summary is not copied)
T=$(mktemp -d)
mkdir -p $T/m $T/w $T/s
ln -s $PWD/$R/mixed_stats $T/m/stats; ln -s $PWD/$R/mixed_pmc_valu $T/m/pmc_valu
ln -s $PWD/$F/pmc_fetch $T/m/pmc_fetch; ln -s $PWD/$F/pmc_write $T/m/pmc_write
python3 tools/pmc_summary.py $T/m > /dev/null
cp $R/mixed_bench.json profiles/r02_mixed_bench.json
cp $T/m/kernel_stats.csv profiles/r02_mixed_kernel_stats.csv
cp $T/m/pmc_valu.json profiles/r02_mixed_pmc_valu.json
# 4-wave workgroups: traffic passes (the --stats pass of the default build stands in)
ln -s $PWD/$F/stats $T/w/stats; ln -s $PWD/$R/wg4_pmc_fetch $T/w/pmc_fetch; ln -s $PWD/$R/wg4_pmc_write $T/w/pmc_write
python3 tools/pmc_summary.py $T/w > /dev/null
cp $T/w/pmc_hbm_traffic.json profiles/r02_wg4_pmc_hbm_traffic.json
# bench command with the liquid and sampled-segment legs under rocprofv3 --stats (forces-only instantiation as its own row)
ln -s $PWD/$R/sampled_stats $T/s/stats; ln -s $PWD/$F/pmc_fetch $T/s/pmc_fetch; ln -s $PWD/$F/pmc_write $T/s/pmc_write
python3 tools/pmc_summary.py $T/s > /dev/null
cp $T/s/kernel_stats.csv profiles/r02_sampled_segment_kernel_stats.csv
cp $R/sampled_bench_under_rocprof.json profiles/r02_sampled_segment_bench_under_rocprof.json
cp $R/all_configs.jsonl profiles/r02_all_configs_one_gpu.jsonl
rm -rf $T
git status --short profiles | head -20
