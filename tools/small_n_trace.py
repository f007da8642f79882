#!/usr/bin/env python3
"""2000 steps at n = SMALL_N (default 4096; SMALL_RC = rc / L, default 0.49) for a rocprofv3 --kernel-trace --stats run (measurement tool)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(os.environ.get("SMALL_N", "4096"))
p, r, v = synthetic.make_config(n, rc_over_L=float(os.environ.get("SMALL_RC", "0.49")))
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    eng.verlet_steps(200)
    eng.enqueue_steps(2000)
    eng.synchronize()
    eng.collect_steps(2000)
