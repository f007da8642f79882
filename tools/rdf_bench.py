#!/usr/bin/env python3
"""Timing of the RDF pair pass (ljmd_rdf_histogram) at N = 262144, all particles (measurement tool)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402
from ljmd_amd import analysis, synthetic  # noqa: E402

for n in (32768, 262144):
    p, r, _ = synthetic.make_config(n)
    h = np.zeros(400, dtype=np.uint64)
    analysis.rdf_histogram(r[0], r[1], r[2], p.box_length, 400, 0.5 * p.box_length, h)      # warm-up (module load)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        analysis.rdf_histogram(r[0], r[1], r[2], p.box_length, 400, 0.5 * p.box_length, h)
    dt = (time.perf_counter() - t0) / reps
    pairs = n * (n - 1) / 2
    print(f"n={n}: {dt * 1e3:8.2f} ms per snapshot incl. H2D/D2H = {pairs / dt:.3e} unordered pairs/s", flush=True)
