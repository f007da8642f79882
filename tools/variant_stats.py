#!/usr/bin/env python3
"""Distribution of the Newton-3 kernel's loop variants over the (row tile, column tile) passes of the bench workload,
from a measurement build of the library (make -C .../csrc OUT=.../libljmd_stats.so EXTRA=-DLJMD_VARIANT_STATS).
Also counts, on the host, how many of those passes' pairs are really inside the cutoff."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["LJMD_LIBRARY"] = str(ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "libljmd_stats.so")
sys.path.insert(0, str(ROOT))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic, _lib  # noqa: E402

n = int(os.environ.get("STATS_N", "262144"))
p, r, v = synthetic.make_config(n)
lib = _lib.load()
buf = (C.c_ulonglong * 80)()
names = {8: "no image", 16: "common image, x only", 17: "common image, y only", 18: "common image, z only",
         0: "common image, several axes", 24: "x general, others no image", 25: "y general, others no image", 26: "z general, others no image", 1: "x general", 2: "y general", 4: "z general", 7: "all general"}
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    for label, steps in (("t = 0 (jittered lattice)", 0), ("after 300 steps (liquid, right after a re-sort)", 300),
                         ("after 309 steps (liquid, 9 steps after the re-sort)", 9)):
        if steps:
            eng.verlet_steps(steps)
        lib.ljmd_debug_variant_stats(buf, 1)
        eng.compute_forces()
        eng.synchronize()
        lib.ljmd_debug_variant_stats(buf, 1)
        by_rows = [buf[56 + k] for k in range(5)]
        tot = sum(buf[k] for k in range(56))
        pairs_all = n * (n - 1) / 2
        clu = [buf[64 + k] for k in range(5)]                   # cluster passes: (cluster, active row tiles) counts
        clu_rows = sum(k * clu[k] for k in range(5))
        print(f"== {label}: {tot} (row tile, column tile) passes in the 64-step loops + {clu_rows} (row tile, 16-cluster) "
              f"passes = {(tot * 4096 + clu_rows * 1024) / pairs_all:.3f} of all unordered pairs "
              f"(diagonal-tile passes are not counted)")
        if sum(clu):
            print(f"  cluster passes: {sum(clu) // 4} column tiles; clusters by active row tiles (0..4): "
                  + ", ".join(f"{k}: {100.0 * clu[k] / sum(clu):.1f} %" for k in range(5))
                  + f"; pair evaluations in clusters: {100.0 * clu_rows * 1024 / (tot * 4096 + clu_rows * 1024):.1f} % of all")
        rows_tot = sum(k * by_rows[k] for k in range(5))
        if rows_tot:
            print("  column-tile passes by number of active row tiles (1..4): "
                  + ", ".join(f"{k}: {100.0 * by_rows[k] / sum(by_rows):.1f} %" for k in range(1, 5))
                  + f"; share of the pair evaluations in passes with all four active: {100.0 * 4 * by_rows[4] / rows_tot:.1f} %")
        for nu, name in names.items():
            a, b = buf[nu * 2], buf[nu * 2 + 1]
            if a + b:
                print(f"  {name:30s} {100.0 * (a + b) / tot:6.2f} %   of which INNER (no cutoff test) {100.0 * b / tot:6.2f} %")
