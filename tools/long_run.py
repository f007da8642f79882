#!/usr/bin/env python3
"""Energy conservation over a long NVE run at the bench size (north-star: 10 000 steps at N = 262144):
total energy, temperature, pressure and total momentum in blocks of 500 steps, fp64 engine.
Prints one JSON line per block (progress) and a summary line.  LONG_N, LONG_STEPS, LONG_MODE=fp64|mixed, LONG_LATTICE=auto|sc|fcc."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic, _lib  # noqa: E402
from ljmd_amd.physics import observables  # noqa: E402

n = int(os.environ.get("LONG_N", "262144"))
steps = int(os.environ.get("LONG_STEPS", "10000"))
mode = os.environ.get("LONG_MODE", "fp64")
block = 500
p, r, v = synthetic.make_config(n, lattice=os.environ.get("LONG_LATTICE", "auto"))
with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE if mode == "mixed" else _lib.PRECISION_FP64) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    e0 = eng.compute_forces()[0]
    k0 = eng.kinetic_energy()
    etot0 = e0 + k0
    blocks, t_all = [], time.perf_counter()
    for b in range(steps // block):
        t0 = time.perf_counter()
        e, k, d, _dd = eng.verlet_steps(block)
        dt = time.perf_counter() - t0
        etot = e + k
        temp = 2.0 * k / (3.0 * n)
        press = np.array([observables(p, ee, kk, ddd)[2] for ee, kk, ddd in zip(e, k, d)])
        vel = np.stack(eng.get_state(("v",))["v"])
        rec = {"steps": (b + 1) * block, "steps_per_s": block / dt, "etot_mean": float(etot.mean()),
               "etot_std_rel": float(etot.std() / abs(etot.mean())), "etot_last": float(etot[-1]),
               "T_mean": float(temp.mean()), "P_mean": float(press.mean()),
               "momentum_per_particle": [float(x) for x in vel.sum(axis=1) / n]}
        blocks.append(rec)
        print(json.dumps(rec), flush=True)
    wall = time.perf_counter() - t_all
m = np.array([b["etot_mean"] for b in blocks])
# equilibrated part (after the lattice has melted): drift of the block means, least-squares slope per step
half = len(m) // 2
slope = np.polyfit(np.arange(half, len(m)) * block, m[half:], 1)[0] if len(m) - half >= 2 else float("nan")
print(json.dumps({"summary": True, "n": n, "steps": steps, "mode": mode, "dt": p.dt, "wall_s": wall,
                  "steps_per_s": steps / wall, "etot_t0": etot0,
                  "block_mean_range_rel": float((m.max() - m.min()) / abs(m.mean())),
                  "second_half_block_mean_range_rel": float((m[half:].max() - m[half:].min()) / abs(m.mean())),
                  "second_half_drift_per_step_rel": float(slope / abs(m.mean())),
                  "momentum_per_particle_last": blocks[-1]["momentum_per_particle"]}))
