#!/usr/bin/env python3
"""Per-rank pair-kernel times of an 8-rank decomposition on one card: the caller's index ranges (LJMD_MULTI_MIGRATE_EVERY=0)
against the k-d blocks of the on-device deal, at the start and in the liquid.  Measurement tool."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

G = int(os.environ.get("PROBE_G", "8"))
for n in [int(x) for x in os.environ.get("PROBE_NS", "65536,262144").split(",")]:
    p, r, v = synthetic.make_config(n)
    for label, env in (("index ranges", {"LJMD_MULTI_MIGRATE_EVERY": "0"}), ("k-d blocks", {"LJMD_MULTI_MIGRATE_EVERY": "100000"})):
        os.environ.update(env)
        with Engine(p, devices=[0] * G) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            for phase, warm in (("lattice", 5), ("liquid", 400)):
                eng.advance(warm)
                eng.profile_enable(True)
                eng.enqueue_steps(10)
                eng.collect_steps(10)
                pr = [eng.profile_read_rank(g) for g in range(G)]
                eng.profile_enable(False)
                pm = [q["pair_ms"] for q in pr]
                print(f"n={n} {label:13s} {phase:8s} pair ms per rank: " + " ".join(f"{x:.3f}" for x in pm) +
                      f" | sum {sum(pm):.3f} max {max(pm):.3f}  reduce+kick sum {sum(q['reduce_ms'] for q in pr):.3f}", flush=True)
