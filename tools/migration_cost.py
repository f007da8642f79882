#!/usr/bin/env python3
"""Cost of one ownership migration (ljmd_migrate: pack, all-gather of ru/v/a/ids, k-d deal of all n particles on every
rank, select, re-sort of every shard, position exchange) with eight rank engines on ONE card (peer-copy exchange: every
rank's sorts run one after the other here; on eight devices they run side by side).  Measurement tool."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

G = int(os.environ.get("PROBE_G", "8"))
for n in (65536, 262144, 1048576):
    p, r, v = synthetic.make_config(n)
    os.environ["LJMD_MULTI_MIGRATE_EVERY"] = "1000000"
    with Engine(p, devices=[0] * G) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.advance(10)
        eng.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.migrate()
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
            eng.advance(2)
            eng.synchronize()
        t0 = time.perf_counter()
        eng.advance(5)
        eng.synchronize()
        step = (time.perf_counter() - t0) / 5
        print(f"n={n} G={G} on one card: migration {1e3 * min(ts):.2f} ms (of 3: {', '.join('%.2f' % (1e3 * t) for t in ts)}), "
              f"one step of all ranks {1e3 * step:.2f} ms, migrations {eng.migrations()}", flush=True)
