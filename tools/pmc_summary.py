#!/usr/bin/env python3
"""Summarises the rocprofv3 output of tools/collect_profiles.sh:
   <dir>/kernel_stats.csv       per-kernel launch statistics (copied from the --stats pass)
   <dir>/pmc_hbm_traffic.json   per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B read request and is
doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])


def find(sub, suffix):
    hits = sorted((out / sub).rglob(f"*{suffix}"))
    if not hits:
        raise SystemExit(f"no {suffix} under {out / sub}")
    return hits[0]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0] if name.startswith("ljmdk::") else name.split("<")[0][:80]


stats_src = find("stats", "kernel_stats.csv")
rows = list(csv.DictReader(open(stats_src)))
with open(out / "kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])

acc = defaultdict(lambda: defaultdict(list))
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for r in csv.DictReader(open(find(sub, "counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])][counter].append(float(r["Counter_Value"]))
kernels = {}
for name, c in acc.items():
    if not name.startswith("ljmdk::"):
        continue
    fetch = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    write = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    kernels[name] = {"FETCH_SIZE_KiB_mean": fetch, "launches_FETCH_SIZE": len(c["FETCH_SIZE"]),
                     "WRITE_SIZE_KiB_mean": write, "launches_WRITE_SIZE": len(c["WRITE_SIZE"]),
                     "hbm_bytes_per_launch": 1024.0 * (2.0 * fetch + write)}
doc = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 5 --warmup 1 "
                  "--no-cpu-baseline (two separate passes; tools/collect_profiles.sh)",
       "units": "FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); per-launch means",
       "gfx950_correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md: reads tallied at 64 B per 128-B request)",
       "kernels": kernels}
(out / "pmc_hbm_traffic.json").write_text(json.dumps(doc, indent=1))
for k in kernels:
    if k.startswith(("ljmdk::pair_n3_kernel<", "ljmdk::drift_kick_kernel<", "ljmdk::reduce_forces_kernel<")):
        print(k, f"{kernels[k]['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch")
for r in rows[:6]:
    print(short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e6:.4f} ms avg", r["Percentage"], "%")
