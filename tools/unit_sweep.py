#!/usr/bin/env python3
"""Work-item shapes of the Newton-3 pair kernel for single-rank systems: tiles per row group (LJMD_N3_ROW_TILES) x parts per
pass (LJMD_N3_PARTS: read only by the round-4 measurement builds, the knob is no longer in the tree) x work items aimed at
(LJMD_N3_TARGET_WAVES), in the liquid (200 steps from the jittered lattice).
Prints step rate and the pair kernel's min / median launch (HIP events).  Measurement tool.
usage: unit_sweep.py n [rt,parts,target ...]      ("0,0,0" = the library's own choice)"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(sys.argv[1])
combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(0, 0, 0)]
p, r, v = synthetic.make_config(n)
# the liquid once, with the library's own choice; every variant starts from the same state
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    eng.verlet_steps(200 if n <= 65536 else 40)
    st = eng.get_state(("r", "v"))
for rt, parts, target in combos:
    for k, val in (("LJMD_N3_ROW_TILES", rt), ("LJMD_N3_PARTS", parts), ("LJMD_N3_TARGET_WAVES", target)):
        if val:
            os.environ[k] = str(val)
        else:
            os.environ.pop(k, None)
    with Engine(p) as eng:
        eng.set_state(st["r"][0], st["r"][1], st["r"][2], st["v"][0], st["v"][1], st["v"][2])
        eng.compute_forces()
        eng.verlet_steps(20)
        nst = min(4000, max(20, int(1e5 * 4096 / n / 8)))
        best = 0.0
        for _ in range(3):
            eng.synchronize()
            t0 = time.perf_counter()
            eng.enqueue_steps(nst)
            eng.synchronize()
            best = max(best, nst / (time.perf_counter() - t0))
            e = eng.collect_steps(nst)
        eng.profile_enable(True)
        eng.verlet_steps(min(nst, 40))
        prof = eng.profile_read_rank(0)
    pairs = n * (n - 1) / 2 * best
    print(f"n = {n:7d} rt {rt} parts {parts} target {target:7d}  {best:9.1f} steps/s ({1e6 / best:8.1f} us/step)  {pairs:.3e} pairs/s  "
          f"pair kernel {1e3 * prof['pair_ms_min']:8.1f} us min {1e3 * prof['pair_ms_median']:8.1f} med  reduce+kick "
          f"{1e3 * prof['reduce_ms_median']:7.1f} us  epot {e[0][-1]:.10e}", flush=True)
