"""Production-run loop around the resident engine: the host-side mirror of
`program md_simulation` (scripts/md_simulation_program.f90:208-391) restricted to what
touches the hot path -- parameter + rv_init input, t=0 forces, the MD loop, the sampling
condition, instantaneous_energies.dat and rva.dat output -- plus the end-of-run statistics
files (:401-560) through `stats`.

Between two sampling steps the state never leaves HBM, and the host's sample I/O is
pipelined against the GPU: at a sampling step the state is frozen on the device
(`snapshot_begin`), the next segment of steps is enqueued at once, and only then does the
host wait for the snapshot transfer and write the files.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from . import io_formats, stats
from .physics import Engine
from .read_input_files import RunControl, read_simulation_parameters

MAX_PENDING_STEPS = 4096      # LJMD_MAX_PENDING_STEPS (include/ljmd.h)


@dataclass
class RunResult:
    n_samples: int = 0
    time: list = field(default_factory=list)
    epot: list = field(default_factory=list)
    ekin: list = field(default_factory=list)
    etot: list = field(default_factory=list)
    temp: list = field(default_factory=list)
    press: list = field(default_factory=list)
    steps_per_second: float = 0.0
    summary: dict = field(default_factory=dict)     # means, stds, thermodynamic coefficients


def run_md_simulation(root_dir, device: int = 0, write_rva: bool = True) -> RunResult:
    """Runs `root_dir/inputs/input_simulation_parameters.txt` from `root_dir/outputs/rv_init.dat`
    and writes root_dir/outputs/one_run/{instantaneous_energies.dat, rva.dat, corr_*.dat,
    corrmean_*.dat, md_final_results.txt}."""
    import time as _time

    root = Path(root_dir)
    ctl: RunControl = read_simulation_parameters(root / "inputs" / "input_simulation_parameters.txt")
    p = ctl.params
    r, v = io_formats.read_rv_init(root / "outputs" / "rv_init.dat", p.n)
    out_dir = root / "outputs" / "one_run"
    if not out_dir.is_dir():
        # md_simulation_program.f90:250 stops when the directory is missing
        raise ValueError("md_simulation: cannot open outputs/one_run/rva.dat")

    n_snap = max(0, ctl.total_steps // ctl.output_interval - ctl.warmup_steps // ctl.output_interval)  # :254-255
    res = RunResult()
    acc = stats.RunStatistics(p.n, p.volume)                     # md_means_init (:272)
    with Engine(p, device=device) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])        # :221-231 (ru <- r)
        eng.compute_forces()                                     # :236
        rva = io_formats.RvaWriter(out_dir / "rva.dat", p.n, p.box_length, p.dt, ctl.output_interval,
                                   n_snap) if write_rva else None
        f_en = open(out_dir / "instantaneous_energies.dat", "w")
        f_en.write(io_formats.ENERGIES_HEADER + "\n")
        t = 0.0
        step = 0
        t0 = _time.perf_counter()
        def segment_length(from_step: int) -> int:
            """steps to the next sampling instant of :361 (or the end), capped by the pending-step limit"""
            nxt = (from_step // ctl.output_interval + 1) * ctl.output_interval
            while nxt <= ctl.warmup_steps:
                nxt += ctl.output_interval
            return min(max(min(nxt, ctl.total_steps) - from_step, 0), MAX_PENDING_STEPS)

        # software pipeline: the GPU runs the next segment while the host writes this sample; only the last step of a
        # segment is read below (:361), so the steps in between run the forces-only pair kernel (LJMD_SAMPLED_STEPS=0:
        # energy sums on every step, as in the Fortran driver)
        sampled = os.environ.get("LJMD_SAMPLED_STEPS", "1") != "0"
        count = segment_length(step)
        eng.enqueue_steps(count, sampled=sampled)
        while step < ctl.total_steps:
            epot, ekin, d_epot, dd_epot = eng.collect_steps(count)
            for _ in range(count):
                t = t + p.dt                                      # :356 accumulated, not step*dt
            step += count
            sample_now = step > ctl.warmup_steps and step % ctl.output_interval == 0
            if sample_now and rva is not None:
                eng.snapshot_begin()
            count = segment_length(step)
            if count > 0:
                eng.enqueue_steps(count, sampled=sampled)
            if sample_now:
                e, k = float(epot[-1]), float(ekin[-1])
                temp, press = acc.push(e, k, d_epot[-1], dd_epot[-1])                   # :371-372
                etot = e + k
                f_en.write(io_formats.energies_row(t, e, k, etot, temp, press) + "\n")   # :374
                res.n_samples += 1
                for lst, val in ((res.time, t), (res.epot, e), (res.ekin, k), (res.etot, etot),
                                 (res.temp, temp), (res.press, press)):
                    lst.append(val)
                if rva is not None:
                    st = eng.snapshot_end()
                    rva.write_snapshot(st["r"], st["ru"], st["v"], st["a"])           # :384-387
        res.steps_per_second = ctl.total_steps / max(_time.perf_counter() - t0, 1e-12)
        f_en.close()
        if rva is not None:
            rva.close()
    if res.n_samples <= 0:
        raise ValueError("md_simulation: no samples were taken (check warmup_steps/output_interval).")  # :399
    res.summary = stats.write_run_statistics(out_dir, p, ctl.total_steps, ctl.output_interval,
                                             ctl.warmup_steps, acc)
    return res
