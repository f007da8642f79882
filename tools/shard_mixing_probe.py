#!/usr/bin/env python3
"""Multi-rank runs shard the particles by INDEX range: a rank owns a fixed set of particles, and in a liquid that set
diffuses out of the slab it filled at t = 0.  How fast do a rank's tiles (64 of ITS particles each) loosen, i.e. how does
the step time of a G-rank run grow over a long run?  G ranks on one card through the multi-device handle (peer-copy
exchange), n = PROBE_N; prints steps/s over blocks of 1000 steps.  Measurement tool."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(os.environ.get("PROBE_N", "65536"))
G = int(os.environ.get("PROBE_G", "8"))
blocks = int(os.environ.get("PROBE_BLOCKS", "10"))
p, r, v = synthetic.make_config(n)
for label, kw in ((f"{G} ranks on one card", dict(devices=[0] * G)), ("one rank", dict())):
    with Engine(p, **kw) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        for b in range(blocks):
            eng.synchronize()
            t0 = time.perf_counter()
            eng.advance(1000)
            eng.synchronize()
            dt = time.perf_counter() - t0
            print(f"{label}: steps {1000 * b:6d}..{1000 * (b + 1):6d}: {1000 / dt:8.1f} steps/s", flush=True)
