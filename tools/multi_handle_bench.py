#!/usr/bin/env python3
"""Whole-system step time of the single-process multi-device handle with G rank engines on ONE card (copy
exchange) against the single engine: what the index-range decomposition + the two exchanges + G-fold launch count
cost when they cannot buy any parallelism (measurement tool).  usage: multi_handle_bench.py [n] [steps]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p, r, v = synthetic.make_config(n)
ref = None
for G in (1, 2, 4, 8):
    with Engine(p, devices=[0] * G) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = eng.compute_forces()
        eng.verlet_steps(3)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.enqueue_steps(steps)                     # returns when everything is enqueued: the host thread's share
        host = time.perf_counter() - t0
        e, k, d, dd = eng.collect_steps(steps)
        secs = time.perf_counter() - t0
    et = e[-1] + k[-1]
    ref = et if ref is None else ref
    print(f"G = {G}: host enqueue {1e3 * host / steps:6.3f} ms per step ({1e3 * host / steps / G:6.3f} per rank), "
          f"{1e3 * secs / steps:8.3f} ms per step for the whole system on one card ({steps / secs:6.2f} steps/s), "
          f"epot(t=0) = {e0[0]:.12e}, Etot after {steps + 3} steps rel. to G = 1: {abs(et - ref) / abs(ref):.1e}", flush=True)
