export TMPDIR=/tmp
mkdir -p gpurun_out/wgpmc
for W in 1 2 4; do
  LJMD_N3_WG_WAVES=$W rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/wgpmc/w$W -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> gpurun_out/wgpmc/w$W.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for W in (1, 2, 4):
    f = glob.glob(f"gpurun_out/wgpmc/w{W}/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair_n3_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"W = {W}: SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.4e}  kernel cycles {cyc:.4e}  valu_issue_frac {m['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f}  valu_busy_frac {m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f}")
PY
