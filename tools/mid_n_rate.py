#!/usr/bin/env python3
"""Step rate of mid-size single-rank systems (between the two-launch path, n <= 8192, and the bench size).  Measurement tool."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402
for n in (8192, 12288, 16384, 32768, 65536, 131072):
    p, r, v = synthetic.make_config(n)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.verlet_steps(50)
        nst = min(4000, max(50, int(2e5 * 4096 / n / 8)))
        best = 0.0
        for _ in range(3):
            eng.synchronize(); t0 = time.perf_counter()
            eng.enqueue_steps(nst); eng.synchronize()
            best = max(best, nst / (time.perf_counter() - t0))
            eng.collect_steps(nst)
    pairs = n * (n - 1) / 2 * best
    print(f"n = {n:6d}  {best:9.1f} steps/s  ({1e6 / best:8.1f} us per step)  {pairs:.3e} pair interactions/s", flush=True)
