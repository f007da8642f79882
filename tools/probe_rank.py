#!/usr/bin/env python3
"""Per-rank kernel times of a G-rank run, measured on ONE GPU: rank 0's engine of an n_ranks = G
decomposition with the collectives left out (exchange buffer pre-filled by set_state, force exchange
marked external).  Shows what does and does not shrink with G (measurement tool)."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402
from ljmd_amd import Engine, synthetic  # noqa: E402

n = int(os.environ.get("PROBE_N", "262144"))
p, r, v = synthetic.make_config(n)
def allgather(engines):
    for e in engines:
        e.synchronize()
    for src in engines:
        sp, _tot, off, cnt = src.exchange_buffer()
        for dst in engines:
            if dst is not src:
                dst.memcpy(dst.exchange_buffer()[0] + 8 * off, sp + 8 * off, 8 * cnt, 3)


for G in [int(x) for x in os.environ.get("PROBE_G", "1,2,4,8").split(",")]:
    engines = [Engine(p, rank=g, n_ranks=G) for g in range(G)]
    for e in engines:
        e.force_buffers(True)
        e.set_state(r[0], r[1], r[2], v[0], v[1], v[2])      # every rank k-d sorts its own block
    allgather(engines)                                        # ... and everybody sees the sorted blocks
    if os.environ.get("PROBE_MIGRATE", "0") == "1" and G > 1:
        # the ownership migration of the one-process-per-GPU form with this script as the collective: k-d blocks by position
        for e in engines:
            e.migrate_pack()
        for e in engines:
            e.synchronize()
        bufs = [e.migrate_buffer() for e in engines]
        for g, (sp, _tot, off, cnt) in enumerate(bufs):
            for d, dst in enumerate(engines):
                if d != g:
                    dst.memcpy(bufs[d][0] + 8 * off, sp + 8 * off, 8 * cnt, 3)
        for e in engines:
            e.migrate_deal()
        allgather(engines)
    which = [int(x) for x in os.environ.get("PROBE_RANKS", "0").split(",")]
    for rk in [w for w in which if w < G]:
      eng = engines[rk]
      eng.forces_partial()
      for _ in range(2):
          eng.step_begin(); eng.step_forces(); eng.step_finish()
      eng.read_partials(2)
      eng.profile_enable(True)
      eng.synchronize()
      t0 = time.perf_counter()
      steps = 8                                                 # < resort interval: rank 0's block order stays valid
      for _ in range(steps):
          eng.step_begin(); eng.step_forces(); eng.step_finish()
      eng.synchronize()
      wall = (time.perf_counter() - t0) / steps * 1e3
      prof = eng.profile_read()
      if G == 1 or "pair1" not in globals():
          pair1 = prof["pair_ms"] * G
      print(f"G={G} rank {rk}: wall {wall:7.3f} ms/step | pair {prof['pair_ms']:7.3f} geometry {prof['geometry_ms']:6.3f} "
            f"drift(+resort) {prof['drift_ms']:6.3f} reduce+kick {prof['reduce_ms']:6.3f}  -> 1/G of the G=1 pair time would be {pair1 / G:6.3f}", flush=True)
    for e in engines:
        e.close()
