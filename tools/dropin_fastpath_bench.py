#!/usr/bin/env python3
"""The REFERENCE's own production program linked against the drop-in shim (oracle/_ref/md_simulation_program_gpu:
its unmodified main program + base/stats modules, lj_potential_energy / verlet replaced by ours), N = 4000 (k = 10):
wall time per verlet_step call with the resident fast path of the stateless entry point (default) and without
(LJMD_STATELESS_FASTPATH=0: upload + spatial re-sort on every call).  Measurement tool."""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
REF = ROOT / "oracle" / "_ref"
INPUT = """k total_steps output_interval warmup_steps
{k} {steps} 100 0
dt L rc_over_L
5.d-3 {L}d0 0.49d0
target_total_energy
{e}d0
"""


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    n = 4 * k ** 3
    L = (n / 0.8) ** (1.0 / 3.0)
    res = {}
    for mode in ("1", "0"):
        with tempfile.TemporaryDirectory() as t:
            t = Path(t)
            (t / "inputs").mkdir()
            (t / "outputs" / "one_run").mkdir(parents=True)
            (t / "inputs" / "input_simulation_parameters.txt").write_text(INPUT.format(k=k, steps=steps, L=repr(L), e=-4.6 * n))
            env = dict(os.environ, LJMD_STATELESS_FASTPATH=mode)
            subprocess.run([str(PKG / "bin" / "md_initial_config_gpu")], cwd=t, check=True, env=env)
            t0 = time.perf_counter()
            subprocess.run([str(REF / "md_simulation_program_gpu")], cwd=t, check=True, env=env)
            res[mode] = time.perf_counter() - t0
            rows = (t / "outputs" / "one_run" / "instantaneous_energies.dat").read_text().splitlines()
            print(f"fastpath={mode}: {res[mode]:.3f} s for {steps} steps at N={n} -> {res[mode] / steps * 1e3:.3f} ms per "
                  f"verlet_step call (incl. process start, t=0 force call, statistics, file output); last row: {rows[-1]}")
    print(f"speed-up of the reference's unmodified caller loop: {res['0'] / res['1']:.2f}x")


if __name__ == "__main__":
    main()
