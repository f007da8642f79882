#!/bin/bash
# kernel durations and gaps of the small-system step (rocprofv3 kernel trace of tools/small_n_trace.py)
export TMPDIR=/tmp
OUT=gpurun_out/smalln_${SMALL_N:-4096}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 tools/small_n_trace.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:90], r["Calls"], "%.2f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%")
t = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))[-300:-280]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s dur %6.2f us  gap %6.2f us" % (r["Kernel_Name"][:44], (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0))
    prev = e
PY
