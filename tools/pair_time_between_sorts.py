#!/usr/bin/env python3
"""Pair-kernel time of every single step over three re-sort intervals (n = 262144) after PROBE_ADVANCE steps (default 400: the equilibrated liquid; 3: the bench's timed region): does the
kernel slow down between two k-d sorts (tiles loosening as the particles move)?  Measurement tool."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

p, r, v = synthetic.make_config(262144)
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    eng.advance(int(os.environ.get("PROBE_ADVANCE", "400")))
    rows = []
    for s in range(30):
        eng.profile_enable(True)
        eng.verlet_steps(1)
        prof = eng.profile_read()
        eng.profile_enable(False)
        rows.append((s, prof["pair_ms"], prof["drift_ms"]))
    for s, pm, dm in rows:
        print(f"step {s:2d}: pair {pm:7.3f} ms   drift(+re-sort) {dm:6.3f} ms")
