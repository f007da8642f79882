#!/bin/bash
# Pair-kernel counters with / without the cluster passes (LJMD_N3_CLUSTERS): instruction counts, issue fraction and
# kernel time of the same bench command (measurement tool; VERDICT r02 item 6 asks for these three per variant).
export TMPDIR=/tmp
mkdir -p gpurun_out/clupmc
for V in ${VARIANTS:-1 0}; do
  LJMD_N3_CLUSTERS=$V rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/clupmc/v$V -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> gpurun_out/clupmc/v$V.log
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
for V in os.environ.get("VARIANTS", "1 0").split():
    f = glob.glob(f"gpurun_out/clupmc/v{V}/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair_n3_kernel" in r["Kernel_Name"] and "Lb1E" not in r["Kernel_Name"].split("(")[0][-1:]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"LJMD_N3_CLUSTERS={V}: SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.4e}  SQ_INSTS_LDS {m['SQ_INSTS_LDS']:.4e}  SQ_INSTS_SALU {m['SQ_INSTS_SALU']:.4e}  "
          f"kernel cycles {cyc:.4e}  valu_issue_frac {m['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f}  valu_busy_frac {m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f}")
PY
