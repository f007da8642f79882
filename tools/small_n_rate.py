#!/usr/bin/env python3
"""Step rate of small single-rank systems (the reference's own range; BASELINE configs 1-2) with the launch-fusion
knobs: LJMD_FUSE_TAIL=1 (default for n <= 8192: two launches per step -- pair kernel with in-kernel pass descriptors,
tile_tail_kernel), LJMD_FUSE_TAIL=0 (round 2: five launches), LJMD_FUSE=0 (seven).  Measurement tool."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

for n in (108, 1024, 2048, 4096, 6144, 8192):
    p, r, v = synthetic.make_config(n)
    for label, env in (("two launches", {}), ("LJMD_FUSE_TAIL=0", {"LJMD_FUSE_TAIL": "0"}), ("LJMD_FUSE=0", {"LJMD_FUSE": "0"})):
        os.environ.update(env)
        with Engine(p) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            eng.verlet_steps(200)
            best = 0.0
            for _ in range(3):
                eng.synchronize()
                t0 = time.perf_counter()
                eng.enqueue_steps(2000)
                eng.synchronize()
                best = max(best, 2000 / (time.perf_counter() - t0))
                eng.collect_steps(2000)
        for k in env:
            del os.environ[k]
        print(f"n = {n:5d}  {label:18s} {best:9.0f} steps/s  ({1e6 / best:6.1f} us per step)", flush=True)
