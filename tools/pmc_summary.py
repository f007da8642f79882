#!/usr/bin/env python3
"""Summarises the rocprofv3 output of tools/collect_profiles.sh:
   <dir>/kernel_stats.csv       per-kernel launch statistics (copied from the --stats pass)
   <dir>/pmc_hbm_traffic.json   per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
   <dir>/pmc_valu.json          per-kernel SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE
                                per launch and the derived VALU issue fraction (wave-instructions x 4 cycles / (1024 SIMDs x
                                kernel cycles)) -- the evidence behind "the fp64 pipe is saturated" (MEASUREMENTS.md 3.2)
rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B read request and is
doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])


def kernel_source_sha16():
    """the kernel sources these counters were collected with; bench.py quotes a summary as evidence for a run only when
    this matches the sources the run was built from"""
    import hashlib
    root = Path(__file__).resolve().parent.parent
    csrc = root / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "csrc"
    hsh = hashlib.sha256()
    for name in ("ljmd_kernels.hip", "ljmd_internal.h"):
        hsh.update((csrc / name).read_bytes())
    return hsh.hexdigest()[:16]


SHA = kernel_source_sha16()


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
doc = {"kernel_source_sha16": SHA,
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 5 --warmup 1 "
                  "--no-cpu-baseline (two separate passes; tools/collect_profiles.sh)",
       "units": "FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); per-launch means",
       "gfx950_correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md: reads tallied at 64 B per 128-B request)",
       "kernels": kernels}
(out / "pmc_hbm_traffic.json").write_text(json.dumps(doc, indent=1))
# ---- VALU issue evidence (optional pass) ----
valu_dir = out / "pmc_valu"
if valu_dir.exists() and list(valu_dir.rglob("*counter_collection.csv")):
    N_SIMD = 256 * 4
    N_XCD = 8          # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs of an MI355X (19.8 ms kernel -> 3.64e8)
    vacc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(find("pmc_valu", "counter_collection.csv"))):
        vacc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
            # the dispatch's own duration in THIS pass: kernel cycles / it = the shader clock the kernel ran at
            vacc[short(r["Kernel_Name"])]["dispatch_ns_in_counter_pass"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    vk = {}
    for name, c in vacc.items():
        if not name.startswith("ljmdk::"):
            continue
        m = {k: sum(v) / len(v) for k, v in c.items()}
        m["launches"] = len(next(iter(c.values())))
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        m["kernel_cycles"] = cyc
        if cyc > 0 and "SQ_INSTS_VALU" in m:
            m["valu_issue_frac"] = m["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cyc)
            m["cycles_per_valu_instruction_per_simd"] = N_SIMD * cyc / m["SQ_INSTS_VALU"]
        if cyc > 0 and m.get("dispatch_ns_in_counter_pass", 0.0) > 0:
            m["clock_ghz_observed"] = cyc / m["dispatch_ns_in_counter_pass"]
        if cyc > 0 and "SQ_ACTIVE_INST_VALU" in m:
            # SQ_ACTIVE_INST_VALU: quad-cycles with a VALU instruction executing, summed over the SIMDs
            m["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc)
        vk[name] = m
    vdoc = {"kernel_source_sha16": SHA,
            "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                       "--kernel-trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid",
            "units": "per-launch means; SQ_* summed over all shader engines / XCDs by rocprofv3; GRBM_GUI_ACTIVE = GPU-busy "
                     "cycles of the dispatch summed over the 8 XCDs (kernel_cycles = GRBM_GUI_ACTIVE / 8)",
            "valu_issue_frac": "SQ_INSTS_VALU (wave-instructions) x 4 cycles / (1024 SIMDs x kernel_cycles)",
            "valu_busy_frac": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel_cycles)",
            "kernels": vk}
    (out / "pmc_valu.json").write_text(json.dumps(vdoc, indent=1))
    for k, m in vk.items():
        if k.startswith("ljmdk::pair_n3_kernel<"):
            print(k, "SQ_INSTS_VALU %.4g" % m.get("SQ_INSTS_VALU", 0), "valu_issue_frac %.3f" % m.get("valu_issue_frac", 0))
# ---- instruction mix / LDS evidence (optional passes 5 and 6) ----
mix = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_mix_fp64", "pmc_mix_lds"):
    d = out / sub
    if d.exists() and list(d.rglob("*counter_collection.csv")):
        for r in csv.DictReader(open(find(sub, "counter_collection.csv"))):
            mix[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if mix:
    mk = {}
    for name, c in mix.items():
        if not name.startswith("ljmdk::"):
            continue
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if "SQ_INSTS_VALU_FMA_F64" in m:
            m["fp64_arith_wave_instructions"] = sum(m.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                                                           "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
            # executed flop per launch: 64 lanes per wave-instruction (lanes switched off by EXEC included), FMA = 2
            m["executed_fp64_flop"] = 64.0 * (m.get("SQ_INSTS_VALU_ADD_F64", 0.0) + m.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                                              2.0 * m.get("SQ_INSTS_VALU_FMA_F64", 0.0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        mk[name] = m
    mdoc = {"kernel_source_sha16": SHA,
            "command": "rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 "
                       "SQ_INSTS_VALU | SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM "
                       "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -- python3 bench.py --steps 5 --warmup 1 "
                       "--no-cpu-baseline --no-liquid (two passes)",
            "units": "wave-instructions per launch (means); executed_fp64_flop = 64 x (ADD + MUL + 2 FMA + TRANS)",
            "kernels": mk}
    (out / "pmc_instruction_mix.json").write_text(json.dumps(mdoc, indent=1))
    for k, m in mk.items():
        if k.startswith("ljmdk::pair_n3_kernel<"):
            print(k, "fp64 add/mul/fma/trans %.3g/%.3g/%.3g/%.3g" % tuple(m.get("SQ_INSTS_VALU_" + x, 0) for x in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")),
                  "LDS instr %.3g bank conflicts %.3g" % (m.get("SQ_INSTS_LDS", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)))
for k in kernels:
    if k.startswith(("ljmdk::pair_n3_kernel<", "ljmdk::drift_kick_kernel<", "ljmdk::reduce_forces_kernel<")):
        print(k, f"{kernels[k]['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch")
for r in rows[:6]:
    print(short(r["Name"]), r["Calls"], f"{float(r['AverageNs']) / 1e6:.4f} ms avg", r["Percentage"], "%")
