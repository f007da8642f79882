#!/usr/bin/env python3
"""Measured deviation of ONE force evaluation of the bench workload (n = 262144) from the CPU oracle over all
6.9e10 ordered pairs (same computation as tests/test_gpu_parity.py::test_bench_configuration_full_parity_vs_oracle_n262144,
printing the numbers instead of asserting)."""
import ctypes
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

n = 262144
p, r, v = synthetic.make_config(n)
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    e, d, dd = eng.compute_forces()
    a = np.stack(eng.get_state(("a",))["a"])
cores = len(os.sched_getaffinity(0))
try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if quota != "max":
        cores = min(cores, max(1, int(quota) // int(period)))
except (OSError, ValueError):
    pass
ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
x, y, z = (np.ascontiguousarray(q) for q in r)
ao = np.empty((3, n))
se = sd = sdd = 0.0
t0 = time.perf_counter()
for i0 in range(0, n, 32768):
    ax, ay, az, pe, pd, pdd = oracle.rows_raw(po, i0, i0 + 32768, x, y, z)
    ao[0, i0:i0 + 32768], ao[1, i0:i0 + 32768], ao[2, i0:i0 + 32768] = ax, ay, az
    se, sd, sdd = se + pe, sd + pd, sdd + pdd
secs = time.perf_counter() - t0
te, td, tdd = oracle.tail_corrections(po)
ref = (4.0 * (0.5 * se) + te, 24.0 * (0.5 * sd) + td, 24.0 * (0.5 * sdd) + tdd)
ao *= 24.0
print(f"oracle: {secs:.1f} s on {cores} threads")
for name, mine, want in zip(("epot", "d_epot", "dd_epot"), (e, d, dd), ref):
    print(f"{name}: gpu {mine:.16e} oracle {want:.16e} rel.diff {abs(mine - want) / abs(want):.2e}")
print(f"accelerations: max|a_gpu - a_oracle| / max|a| = {np.abs(a - ao).max() / np.abs(ao).max():.2e}, "
      f"rms rel = {np.sqrt(np.mean((a - ao) ** 2)) / np.sqrt(np.mean(ao ** 2)):.2e}")
