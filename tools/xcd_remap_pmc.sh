export TMPDIR=/tmp
mkdir -p gpurun_out/pmcx
for R in 0 4; do
  LJMD_N3_XCD_REMAP=$R rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmcx/a$R -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> gpurun_out/pmcx/a$R.log
  LJMD_N3_XCD_REMAP=$R rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmcx/b$R -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> gpurun_out/pmcx/b$R.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for R in (0, 4):
    m = defaultdict(list)
    for sub in ("a", "b"):
        fs = glob.glob(f"gpurun_out/pmcx/{sub}{R}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        for r in csv.DictReader(open(fs[0])):
            if "pair_n3_kernel" in r["Kernel_Name"]:
                m[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("remap", R, {k: f"{sum(v)/len(v):.4g}" for k, v in m.items()})
PY
