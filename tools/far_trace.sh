#!/bin/bash
# kernel times of the mixed-precision step (rocprofv3 --kernel-trace --stats), optionally with another build of the
# library: tools/far_trace.sh [OLD.so].  Measurement tool.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
run() {  # tag
  rm -rf $R/gpurun_out/far_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/far_$1 -o t -- python3 $R/bench.py --mode mixed --no-liquid --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/far_$1.json 2> $R/gpurun_out/far_$1.err
  f=$(find $R/gpurun_out/far_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1"; python3 - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("%-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
P
}
run new
if [ -n "$1" ]; then export LJMD_LIBRARY=$R/$1; run old; fi
