#!/usr/bin/env python3
"""Summary of tools/pmc_collect.sh: <dir>/kernel_stats.csv (from the --stats pass) and <dir>/pmc.json = per-kernel means
per launch of every counter collected, plus derived figures (kernel cycles, cycles per VALU wave-instruction per SIMD, HBM
bytes with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  Carries the kernel sources' sha like pmc_summary.py."""
import csv
import hashlib
import json
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
args = sys.argv[2] if len(sys.argv) > 2 else ""
root = Path(__file__).resolve().parent.parent
csrc = root / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "csrc"
hsh = hashlib.sha256()
for name in ("ljmd_kernels.hip", "ljmd_internal.h"):
    hsh.update((csrc / name).read_bytes())
SHA = hsh.hexdigest()[:16]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0] if name.startswith("ljmdk::") else name.split("<")[0][:80]


stats = sorted((out / "stats").rglob("*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(out / "kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
acc = defaultdict(lambda: defaultdict(list))
passes = []
for d in sorted(out.glob("pass*")):
    if not d.is_dir():
        continue
    names = set()
    for f in d.rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Counter_Name"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
                acc[short(r["Kernel_Name"])]["dispatch_ns_in_counter_pass"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    passes.append(sorted(names))
N_SIMD, N_XCD = 1024, 8
kernels = {}
for name, c in acc.items():
    if not name.startswith("ljmdk::"):
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    m["launches_per_pass"] = len(next(iter(c.values())))
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    if cyc > 0:
        m["kernel_cycles"] = cyc
        if m.get("dispatch_ns_in_counter_pass", 0.0) > 0:
            m["clock_ghz_observed"] = cyc / m["dispatch_ns_in_counter_pass"]
        if m.get("SQ_INSTS_VALU"):
            m["cycles_per_valu_instruction_per_simd"] = N_SIMD * cyc / m["SQ_INSTS_VALU"]
        if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
            m["active_lane_share_of_valu"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        m["hbm_bytes_per_launch"] = 1024.0 * (2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0))
    kernels[name] = m
doc = {"kernel_source_sha16": SHA,
       "command": "rocprofv3 --pmc <one pass per list below> --kernel-trace -- python3 bench.py %s --steps 5 --warmup 1 "
                  "--no-cpu-baseline --no-liquid (tools/pmc_collect.sh)" % args,
       "passes": passes,
       "units": "per-launch means; SQ_* summed over the chip; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles; "
                "GRBM_GUI_ACTIVE summed over the 8 XCDs (kernel_cycles = / 8); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE "
                "doubled in hbm_bytes_per_launch (gfx950: 64 B tallied per 128-B read)",
       "kernels": kernels}
(out / "pmc.json").write_text(json.dumps(doc, indent=1))
for k, m in sorted(kernels.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:4]:
    print(k, {x: ("%.4g" % m[x]) for x in ("SQ_INSTS_VALU", "kernel_cycles", "cycles_per_valu_instruction_per_simd") if x in m})
