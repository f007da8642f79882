#!/usr/bin/env python3
"""Wall-clock of the thin Fortran production driver with snapshot I/O overlapped with compute
(default) vs serialised (LJMD_ASYNC_IO=0): N = 4 k^3 particles from a jittered start, `steps`
steps, a snapshot every `oi` steps.   python tools/io_overlap_bench.py [k=40] [steps=60] [oi=5]"""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import io_formats, synthetic  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
oi = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 4 * k ** 3
p, r, v = synthetic.make_config(n)
exe = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "bin" / "md_simulation_gpu"
for mode in ("1", "0", "1", "0"):
    with tempfile.TemporaryDirectory() as t:
        t = Path(t)
        (t / "inputs").mkdir()
        (t / "outputs" / "one_run").mkdir(parents=True)
        (t / "inputs" / "input_simulation_parameters.txt").write_text(
            f"{k} {steps} {oi} 0\n{p.dt!r} {p.box_length!r} 0.49d0\n-1.d0\n")
        io_formats.write_rv_init(t / "outputs" / "rv_init.dat", r[0], r[1], r[2], v[0], v[1], v[2])
        t0 = time.perf_counter()
        out = subprocess.run([str(exe)], cwd=t, check=True, capture_output=True, text=True,
                             env=dict(os.environ, LJMD_ASYNC_IO=mode))
        dt = time.perf_counter() - t0
        size = (t / "outputs" / "one_run" / "rva.dat").stat().st_size
        print(f"N={n} steps={steps} oi={oi} LJMD_ASYNC_IO={mode}: {out.stdout.strip().splitlines()[-1]}  "
              f"(process wall {dt:.2f} s, rva.dat {size / 1e6:.0f} MB)", flush=True)
