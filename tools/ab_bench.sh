#!/bin/bash
# Same-box A/B of two builds of libljmd.so: alternating bench.py runs (steps/s | pair kernel, reduce+kick, geometry ms).
# usage: tools/ab_bench.sh OLD.so [rounds]; the tree's own library is the other side.  Measurement tool.
old=$1; rounds=${2:-2}
show() { python3 - "$1" "$2" <<'P'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d["roofline"]
m=(d.get("config5_mixed_precision") or {}).get("value") or 0
print("%-4s %.2f steps/s | liquid %.2f sampled %.2f mixed %.2f | pair %.3f reduce+kick %.3f geometry %.3f drift %.3f ms" % (sys.argv[1], d["value"], d.get("steps_per_s_liquid") or 0, d.get("steps_per_s_sampled_segment") or 0, m, r["kernel_ms_avg"], r["reduce_kick_finalize_ms_avg"], r["geometry_prepass_ms_avg"], r["drift_kick_resort_ms_avg"]))
P
}
for i in $(seq $rounds); do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_new.json 2>gpurun_out/ab_new.err && show new gpurun_out/ab_new.json
  LJMD_LIBRARY=$old python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_old.json 2>gpurun_out/ab_old.err && show old gpurun_out/ab_old.json
done
