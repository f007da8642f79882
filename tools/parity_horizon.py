#!/usr/bin/env python3
"""First step at which the GPU trajectory leaves the reference's raw fp64 dump by more than a given
relative tolerance (N=108, 10 000 steps; N=4096, 200 steps) -- the chaos horizon of SURVEY fact #6."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ljmd_amd  # noqa: E402
from ljmd_amd import Engine, init_params  # noqa: E402

for name in ("traj_n108", "traj_n4096_200"):
    g = np.load(ROOT / "tests" / "golden" / f"{name}.npz")
    n = int(g["n"])
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    ref = g["scalars"]
    nsteps = ref.shape[0] - 1
    with Engine(p) as eng:
        r0, v0 = g["r0"], g["v0"]
        eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        e0, d0, dd0 = eng.compute_forces()
        k0 = eng.kinetic_energy()
        e, k, d, dd = eng.verlet_steps(nsteps)
    mine = np.vstack([[e0, k0, d0, dd0], np.stack([e, k, d, dd], axis=1)])

    def series(sc):
        etot = sc[:, 0] + sc[:, 1]
        temp = 2.0 * sc[:, 1] / (3.0 * n)
        press = (n / p.volume) * temp - sc[:, 2] / (3.0 * p.volume)
        return {"Etot": etot, "T": temp, "P": press}
    a, b = series(mine), series(ref)
    for key in a:
        err = np.abs(a[key] - b[key]) / np.abs(b[key])
        out = []
        for tol in (1e-12, 1e-10, 1e-6, 1e-3):
            idx = np.nonzero(err > tol)[0]
            out.append(f">{tol:g}: {'never' if len(idx) == 0 else 'step %d' % idx[0]}")
        print(f"{name} {key:5s} max over first 200 = {err[:201].max():.2e} | " + ", ".join(out), flush=True)
