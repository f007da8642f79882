#!/bin/bash
# Copies what tools/collect_round4.sh left under gpurun_out/ into profiles/ (the judged, committed copies) and prints the
# figures the README tables quote.  Run here, after the gpurun call has merged its output back.
set -e
S=gpurun_out/profiles_r04final; M=gpurun_out/pmc_mixed
cp $S/bench.json profiles/r04_final_bench.json
cp $S/bench_under_rocprof.json profiles/r04_final_bench_under_rocprof.json
cp $S/kernel_stats.csv profiles/r04_final_kernel_stats.csv
cp $S/pmc_hbm_traffic.json profiles/r04_final_pmc_hbm_traffic.json
cp $S/pmc_valu.json profiles/r04_final_pmc_valu.json
cp $S/pmc_instruction_mix.json profiles/r04_final_pmc_instruction_mix.json
cp $M/pmc.json profiles/r04_mixed_pmc.json
cp $M/kernel_stats.csv profiles/r04_mixed_kernel_stats.csv
cp $M/bench_under_rocprof.json profiles/r04_mixed_bench_under_rocprof.json
cp gpurun_out/r04_mixed_bench.json profiles/r04_mixed_bench.json
cp gpurun_out/r04_small_n_trace_4096.txt gpurun_out/r04_mid_n_rates.txt gpurun_out/r04_n_sweep.txt profiles/
[ -f gpurun_out/mixed_precision_parity_vs_oracle.json ] && cp gpurun_out/mixed_precision_parity_vs_oracle.json profiles/r04_mixed_precision_parity_vs_oracle.json
[ -f gpurun_out/r04_gpu_test_log.txt ] && cp gpurun_out/r04_gpu_test_log.txt profiles/r04_gpu_test_log.txt
python3 - <<'PY'
import json, hashlib
from pathlib import Path
h = hashlib.sha256()
for n in ("ljmd_kernels.hip", "ljmd_internal.h"):
    h.update((Path("molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/csrc") / n).read_bytes())
print("tree kernel sha", h.hexdigest()[:16])
for f in ("r04_final_pmc_valu.json", "r04_mixed_pmc.json", "r04_mixed_precision_parity_vs_oracle.json"):
    print(" ", f, json.load(open("profiles/" + f)).get("kernel_source_sha16"))
d = json.loads(open("profiles/r04_final_bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("fp64 %.2f steps/s | pair kernel avg %.3f min %.3f ms | frac %.3f shortest %.3f at clock %.3f (%.2f GHz) | VALU %.3e issue %.3f | liquid %.1f sampled %.1f mixed %.1f" % (
    d["value"], r["kernel_ms_avg"], r["kernel_ms_min"], r["frac"], r.get("frac_of_shortest_launch", 0), r.get("frac_at_observed_clock", 0),
    r.get("clock_ghz_observed", 0), r.get("valu_wave_instructions_per_launch", 0), r.get("valu_issue_frac", 0), d["steps_per_s_liquid"],
    d["steps_per_s_sampled_segment"], d["config5_mixed_precision"]["value"]))
m = json.loads(open("profiles/r04_mixed_bench.json").read().strip().splitlines()[-1])
print("mixed bench %.1f steps/s, liquid %.1f" % (m["value"], m.get("steps_per_s_liquid") or 0))
PY
head -3 profiles/r04_mixed_kernel_stats.csv | cut -c1-110
head -2 profiles/r04_final_kernel_stats.csv | cut -c1-110
cut -c1-120 profiles/r04_n_sweep.txt
