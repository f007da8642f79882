#!/usr/bin/env python3
"""Step rate and pair-kernel time of single-rank systems of the sizes given on the command line (default: the reference's
own range up to the bench size), in the liquid (200 steps from the jittered lattice first).  Measurement tool.
usage: n_sweep_rate.py [n ...]      env: SWEEP_MODE=1 mixed precision"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536, 131072, 262144]
mode = int(os.environ.get("SWEEP_MODE", "0"))
for n in sizes:
    p, r, v = synthetic.make_config(n)
    with Engine(p, precision_mode=mode) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.verlet_steps(200 if n <= 65536 else 40)
        nst = min(4000, max(20, int(1e5 * 4096 / n / 8)))
        best = 0.0
        for _ in range(3):
            eng.synchronize()
            t0 = time.perf_counter()
            eng.enqueue_steps(nst)
            eng.synchronize()
            best = max(best, nst / (time.perf_counter() - t0))
            e = eng.collect_steps(nst)
        eng.profile_enable(True)
        eng.verlet_steps(min(nst, 40))
        prof = eng.profile_read_rank(0)
    pairs = n * (n - 1) / 2 * best
    print(f"n = {n:7d}  {best:9.1f} steps/s ({1e6 / best:8.1f} us/step)  {pairs:.3e} pairs/s  pair kernel {1e3 * prof['pair_ms_min']:8.1f} us min "
          f"{1e3 * prof['pair_ms_median']:8.1f} med  (with events on)  epot {e[0][-1]:.10e}", flush=True)
