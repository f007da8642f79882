#!/bin/bash
# Same-box, alternating A/B of the pair kernel's optional features by the MINIMUM launch time over the timed launches
# (bench.py: roofline.kernel_ms_min / kernel_ms_median) -- a 1-2 % effect is inside the box-to-box scatter of the average
# but not of the minimum on one box.  usage: tools/feature_ab.sh [rounds]    Measurement tool.
rounds=${1:-3}
show() { python3 - "$1" "$2" <<'P'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-28s %.2f steps/s | pair kernel min %.3f median %.3f avg %.3f ms" % (sys.argv[1], d["value"], r.get("kernel_ms_min") or 0, r.get("kernel_ms_median") or 0, r["kernel_ms_avg"]), flush=True)
P
}
for i in $(seq $rounds); do
  for v in "default" "LJMD_N3_CLUSTERS=0" "LJMD_N3_PERTILE=0" "LJMD_N3_CLUSTERS=0 LJMD_N3_PERTILE=0"; do
    ( [ "$v" = default ] || export $v; python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-liquid > gpurun_out/fab.json 2>gpurun_out/fab.err ) && show "$v" gpurun_out/fab.json
  done
done
