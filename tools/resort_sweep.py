#!/usr/bin/env python3
"""Steps/s in the developed liquid (after `MELT` steps from the jittered lattice) as a function of the re-sort
interval LJMD_RESORT_EVERY (read at engine creation).  Measurement tool."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(os.environ.get("SWEEP_N", "262144"))
melt = int(os.environ.get("MELT", "300"))
p, r, v = synthetic.make_config(n)
for every in (int(x) for x in os.environ.get("EVERY", "5,10,20,40").split(",")):
    os.environ["LJMD_RESORT_EVERY"] = str(every)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.verlet_steps(melt)
        eng.profile_enable(True)
        eng.synchronize()
        t0 = time.perf_counter()
        e, k, _d, _dd = eng.verlet_steps(200)
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
    print(f"re-sort every {every:3d} steps: {200 / dt:7.2f} steps/s | pair {prof['pair_ms']:.3f} ms  drift+resort "
          f"{prof['drift_ms']:.3f} ms  reduce {prof['reduce_ms']:.3f} ms | Etot {e[-1] + k[-1]:.6f}", flush=True)
