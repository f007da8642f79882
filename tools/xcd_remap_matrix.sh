# XCD chunk mapping across sizes / ranks / liquid
for N in 131072 524288; do
  PROBE_N=$N PROBE_ROUNDS=2 python3 tools/probe_force.py "LJMD_N3_XCD_REMAP=0" "LJMD_N3_XCD_REMAP=4" 2>&1 | grep LJMD_ | sed "s/^/n=$N /"
done
for R in 0 4; do LJMD_N3_XCD_REMAP=$R python3 tools/probe_rank.py 2>&1 | grep "^G=" | sed "s/^/remap=$R /"; done
for R in 0 4; do LJMD_N3_XCD_REMAP=$R python3 bench.py --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('remap=$R bench', d['value'], 'liquid', d['steps_per_s_liquid'], 'pair ms', d['roofline']['kernel_ms_avg'], d['liquid']['pair_kernel_ms_avg'])"; done
