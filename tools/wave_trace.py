#!/usr/bin/env python3
"""Where the waves of the small-system pair kernel sit and when (measurement build -DLJMD_WAVE_TRACE: tools/build_variant.sh
trace -DLJMD_WAVE_TRACE; run with LJMD_LIBRARY=variants/libljmd_trace.so).  usage: wave_trace.py [n]   env LJMD_N3_QUAD=0|1"""
import ctypes as C
import os
import sys
from collections import Counter, defaultdict
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p, r, v = synthetic.make_config(n)
lib = _lib.load()
lib.ljmd_debug_wave_trace.argtypes = [C.c_void_p, C.c_int]
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    eng.verlet_steps(200)
    eng.verlet_steps(1)                      # the last launch's trace is what is read
    eng.synchronize()
    name = eng.pair_kernel_name()
    nw = 1 << 16
    buf = np.zeros(nw * 8, dtype=np.uint64)
    rc = lib.ljmd_debug_wave_trace(buf.ctypes.data_as(C.c_void_p), nw)
    assert rc == 0, rc
t = buf.reshape(nw, 8)
t = t[t[:, 0] > 0]
tick = 10.0                                  # ns per s_memrealtime tick (100 MHz)
t0 = t[:, 0].min()
start, desc, loop, end = [(t[:, k].astype(np.int64) - int(t0)) * tick * 1e-3 for k in range(4)]   # us
hw, xcc, heavy = t[:, 4].astype(np.int64), t[:, 5].astype(np.int64) & 15, t[:, 6].astype(np.int64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print(f"n = {n}  kernel {name}  waves traced {len(t)}  span {end.max():.2f} us  (first start 0, last start {start.max():.2f}, last end {end.max():.2f})")
for label, sel in (("heavy", heavy > 0), ("light", heavy == 0)):
    if sel.any():
        print(f"  {label:5s} waves {sel.sum():5d}: start {start[sel].mean():6.2f}  descriptor {np.mean(desc[sel] - start[sel]):5.2f}  "
              f"passes {np.mean(loop[sel] - desc[sel]):6.2f} (max {np.max(loop[sel] - desc[sel]):6.2f})  epilogue {np.mean(end[sel] - loop[sel]):5.2f}  "
              f"lifetime {np.mean(end[sel] - start[sel]):6.2f} us")
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = key * 4 + simd
per_simd = defaultdict(lambda: [0, 0, 0.0])
for k, h_, a, b in zip(simd_key, heavy, desc, loop):
    per_simd[k][0] += 1
    per_simd[k][1] += int(h_ > 0)
    per_simd[k][2] += (b - a)
print(f"  SIMDs used {len(per_simd)} (CUs {len(set(key))}); waves per SIMD: {dict(sorted(Counter(v[0] for v in per_simd.values()).items()))}")
print(f"  heavy waves per SIMD: {dict(sorted(Counter(v[1] for v in per_simd.values()).items()))}")
busy = np.array([v[2] for v in per_simd.values()])
print(f"  summed pass time per SIMD: mean {busy.mean():.2f} us, max {busy.max():.2f}, p90 {np.percentile(busy, 90):.2f}")
edges = np.linspace(0.0, end.max(), 21)
alive = [(int(((start <= x) & (end > x)).sum()), int(((desc <= x) & (loop > x) & (heavy > 0)).sum())) for x in edges]
print("  time us : waves alive / heavy waves inside their passes")
print("  " + "  ".join(f"{x:5.1f}:{a}/{b}" for x, (a, b) in zip(edges, alive)))
