#!/bin/bash
# One-box sweep of the launch-shape knobs of the bench configuration (steps/s, pair kernel ms); measurement tool.
run() { env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-liquid 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.2f steps/s  pair %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms_avg']))" "$*"; }
run LJMD_DUMMY=0
run LJMD_N3_XCD_REMAP=2
run LJMD_N3_XCD_REMAP=8
run LJMD_N3_XCD_REMAP=16
run LJMD_DUMMY=0
run LJMD_N3_TARGET_WAVES=65536
run LJMD_N3_TARGET_WAVES=262144
run LJMD_RESORT_EVERY=5
run LJMD_RESORT_EVERY=20
run LJMD_DUMMY=0
