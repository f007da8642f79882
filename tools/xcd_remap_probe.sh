export TMPDIR=/tmp
PROBE_ROUNDS=3 python3 tools/probe_force.py "LJMD_N3_XCD_REMAP=0" "LJMD_N3_XCD_REMAP=1"
for R in 0 1; do
  LJMD_N3_XCD_REMAP=$R rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/xcd/f$R -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-liquid > /dev/null 2> gpurun_out/xcd/f$R.log
done
python3 - <<'PY'
import csv, glob
for R in (0, 1):
    f = glob.glob(f"gpurun_out/xcd/f{R}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "pair_n3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print(f"remap {R}: pair kernel FETCH_SIZE {sum(v)/len(v):.0f} KiB per launch -> {2*1024*sum(v)/len(v)/1e9:.2f} GB read (x2 gfx950 correction)")
PY
