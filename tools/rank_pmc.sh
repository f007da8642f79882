#!/bin/bash
# VALU counters of the pair kernel of rank 0 of a G-rank decomposition (tools/probe_rank.py, one GPU): is the loss
# against 1/G of the single-rank kernel in the instruction count or in the issue rate?
export TMPDIR=/tmp
mkdir -p gpurun_out/rankpmc
for G in ${RANKS:-1 8}; do
  PROBE_G=$G rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/rankpmc/g$G -o run -- python3 tools/probe_rank.py > gpurun_out/rankpmc/g$G.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
for G in [int(x) for x in os.environ.get("RANKS", "1 8").split()]:
    f = glob.glob(f"gpurun_out/rankpmc/g{G}/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair_n3_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[-8:]) / len(v[-8:]) for k, v in acc.items()}      # the timed steps of rank 0 (the last launches)
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"G = {G}: SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.4e} (x G = {m['SQ_INSTS_VALU'] * G:.4e})  kernel cycles {cyc:.4e}  valu_issue_frac {m['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f}  valu_busy_frac {m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f}")
PY
