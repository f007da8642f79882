#!/usr/bin/env python3
"""A/B timing of the pair-force path at N = 262144 under different LJMD_* knobs, one process,
interleaved rounds (measurement tool).  usage: probe_force.py "K1=V1,K2=V2" "K1=V3" ..."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(os.environ.get("PROBE_N", "262144"))
steps = int(os.environ.get("PROBE_STEPS", "6"))
variants = sys.argv[1:] or [""]
p, r, v = synthetic.make_config(n)
results = {var: [] for var in variants}
for rnd in range(int(os.environ.get("PROBE_ROUNDS", "2"))):
    for var in variants:
        keys = []
        for kv in filter(None, var.split(",")):
            k, val = kv.split("=")
            os.environ[k] = val
            keys.append(k)
        with Engine(p, precision_mode=int(os.environ.get('PROBE_MODE', '0'))) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            eng.verlet_steps(2)
            eng.profile_enable(True)
            e = eng.verlet_steps(steps)
            prof = eng.profile_read()
        for k in keys:
            del os.environ[k]
        results[var].append((prof["pair_ms"], prof["geometry_ms"], prof["drift_ms"], prof["reduce_ms"], e[0][-1]))
for var, rows in results.items():
    a = np.array(rows)
    print(f"{var or 'default':40s} pair {a[:,0].min():8.3f} ms (med {np.median(a[:,0]):8.3f})  geom {a[:,1].min():6.3f}  "
          f"drift {a[:,2].min():6.3f}  reduce {a[:,3].min():6.3f}  epot {a[0,4]:.12e}", flush=True)
