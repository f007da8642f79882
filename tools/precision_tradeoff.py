#!/usr/bin/env python3
"""BASELINE config 5: throughput vs energy-drift trade-off of the mixed-precision mode at N = 262144.
Runs the same start for `steps` steps in fp64 and in LJMD_PRECISION_FP32_FORCE with several r_split
values; prints one JSON object (measurement tool, results committed under profiles/)."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402
from ljmd_amd import Engine, synthetic, _lib  # noqa: E402

n = int(os.environ.get("TRADEOFF_N", "262144"))
steps = int(os.environ.get("TRADEOFF_STEPS", "1000"))
p, r, v = synthetic.make_config(n)
runs = [("fp64", _lib.PRECISION_FP64, None)] + [(f"mixed r_split={s}", _lib.PRECISION_FP32_FORCE, s) for s in ("8", "5", "3", "0")]
out = {"n": n, "steps": steps, "dt": p.dt, "runs": []}
ref = None
for name, mode, split in runs:
    if split is not None:
        os.environ["LJMD_FP32_SPLIT"] = split
    with Engine(p, precision_mode=mode) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = eng.compute_forces()[0]
        k0 = eng.kinetic_energy()
        eng.verlet_steps(3)
        eng.synchronize()
        t0 = time.perf_counter()
        e, k, d, dd = eng.verlet_steps(steps)
        dtw = time.perf_counter() - t0
    etot = e + k
    rec = {"name": name, "steps_per_s": steps / dtw, "ms_per_step": 1e3 * dtw / steps,
           "etot_t0": e0 + k0, "etot_mean": float(etot.mean()), "etot_std_rel": float(etot.std() / abs(etot.mean())),
           "etot_range_rel": float((etot.max() - etot.min()) / abs(etot.mean())),
           "etot_last": float(etot[-1])}
    if ref is None:
        ref = etot
    else:
        rec["max_rel_dev_from_fp64_first_200"] = float(np.max(np.abs(etot[:200] - ref[:200]) / np.abs(ref[:200])))
        rec["rel_dev_of_mean_from_fp64"] = float(abs(etot.mean() - ref.mean()) / abs(ref.mean()))
    out["runs"].append(rec)
    print(json.dumps(rec), flush=True)
print(json.dumps(out))
