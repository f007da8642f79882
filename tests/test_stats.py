"""End-of-run statistics -- CPU tests, no GPU.

The reference's raw per-step scalars for BASELINE config 1 are in tests/golden/traj_n108.npz
(written by oracle/ref_harness over the reference's own modules, from the same rv_init.dat the
reference's production program started from), and the files the reference's production
program wrote for that run are in tests/golden/ref_run_n108_oi{10,100}/.  Feeding the former
into our statistics code must reproduce the latter: byte for byte (13 significant digits).
"""
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from ljmd_amd import stats
from ljmd_amd.read_input_files import read_simulation_parameters

PKG = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
REPLAY = PKG / "bin" / "md_stats_replay"
FILES = [f"{kind}_{obs}.dat" for kind in ("corr", "corrmean") for obs in stats.OBSERVABLES] + ["md_final_results.txt"]


# (fixture directory, raw scalars, output_interval, warmup_steps, total_steps, samples)
CASES = {"oi10": ("ref_run_n108_oi10", "traj_n108.npz", 10, 100, 1000, 90),
         "oi100": ("ref_run_n108_oi100", "traj_n108.npz", 100, 100, 1000, 9),
         # k = 4 (N = 256), rc = 0.35 L, dt = 0.002, L = 6.5, sampling every 20 after 40: differs from config 1 everywhere
         "k4": ("ref_run_n256_k4", "traj_n256_k4.npz", 20, 40, 600, 28)}


def _samples(case):
    _d, traj, oi, warm, total, _n = CASES[case]
    sc = np.load(GOLDEN / traj)["scalars"]                       # [steps + 1, 4]: epot, ekin, d_epot, dd_epot after k steps
    first = (warm // oi + 1) * oi                                # sampling condition: step > warmup and step % oi == 0
    return sc[first:total + 1:oi]


@pytest.mark.parametrize("case", list(CASES))
def test_python_mirror_reproduces_reference_files(tmp_path, case):
    src = GOLDEN / CASES[case][0]
    ctl = read_simulation_parameters(src / "input_simulation_parameters.txt")
    s = _samples(case)
    assert s.shape == (CASES[case][5], 4)
    acc = stats.RunStatistics(ctl.params.n, ctl.params.volume)
    energies = [ln.split() for ln in (src / "instantaneous_energies.dat").read_text().splitlines()[1:]]
    for row, ref_row in zip(s, energies):
        temp, press = acc.push(*row)
        # T and P of the time series (7 digits in the text file)
        assert abs(temp - float(ref_row[4])) <= 1e-6 * abs(temp)
        assert abs(press - float(ref_row[5])) <= 2e-6 * max(abs(press), 0.1)
    out = stats.write_run_statistics(tmp_path, ctl.params, ctl.total_steps, ctl.output_interval, ctl.warmup_steps, acc)
    for name in FILES:
        assert (tmp_path / name).read_text() == (src / name).read_text(), name
    if case == "oi10":
        assert abs(out["coefficients"]["gamma"] - 2.848590910736) < 1e-11


@pytest.mark.skipif(not REPLAY.exists(), reason="run __graft_entry__.build() first (needs amdflang)")
@pytest.mark.parametrize("case", list(CASES))
def test_fortran_statistics_modules_reproduce_reference_files(tmp_path, case):
    """fortran/md_stats.f90 + md_run_outputs.f90 (what md_simulation_gpu links), driven by the
    CPU-only replay tool."""
    src = GOLDEN / CASES[case][0]
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs" / "one_run").mkdir(parents=True)
    shutil.copy(src / "input_simulation_parameters.txt", tmp_path / "inputs")
    _samples(case).astype("<f8").tofile(tmp_path / "outputs" / "one_run" / "samples.bin")
    subprocess.run([str(REPLAY)], cwd=tmp_path, check=True, timeout=60)
    for name in FILES:
        assert (tmp_path / "outputs" / "one_run" / name).read_bytes() == (src / name).read_bytes(), name
    # the summary is appended, not replaced (md_simulation_program.f90:532)
    subprocess.run([str(REPLAY)], cwd=tmp_path, check=True, timeout=60)
    twice = (tmp_path / "outputs" / "one_run" / "md_final_results.txt").read_text()
    assert twice == (src / "md_final_results.txt").read_text() * 2


def test_estimators_against_numpy():
    rng = np.random.default_rng(5)
    x = rng.normal(size=200).cumsum()
    c = stats.autocovariance(list(x), 50)
    m = x.mean()
    for lag in (0, 1, 17, 50):
        ref = np.dot(x[:200 - lag] - m, x[lag:] - m) / (200 - lag)
        assert abs(c[lag] - ref) <= 1e-12 * abs(c[0])
    cn = stats.normalise_by_lag0(c)
    assert cn[0] == 1.0
    cm, cnm = stats.block_mean_autocovariance(list(x), 3, 50)
    blocks = [stats.autocovariance(list(x[b * 66:(b + 1) * 66]), 50) for b in range(3)]   # 200 // 3 = 66, tail unused
    assert abs(cm[7] - sum(b[7] for b in blocks) / 3) <= 1e-12 * abs(cm[0])
    assert abs(cnm[0] - 1.0) < 1e-15


def test_edge_cases_follow_the_reference():
    assert stats.lag_limit(0) == -1 and stats.lag_limit(1) == -1          # no correlations below 2 samples
    assert stats.lag_limit(2) == 1 and stats.lag_limit(9) == 4 and stats.lag_limit(90) == 45
    assert stats.lag_limit(5000) == 1000
    assert stats.normalise_by_lag0([0.0, 1.0]) == [0.0, 0.0]              # |C(0)| <= 1e-14 -> zeros
    with pytest.raises(ValueError, match="lag_max must be < n_samples"):
        stats.autocovariance([1.0, 2.0], 2)
    with pytest.raises(ValueError, match="max_lag must be < block_len"):
        stats.block_mean_autocovariance([1.0, 2.0, 3.0, 4.0], 2, 2)
    acc = stats.RunStatistics(108, 135.0)
    with pytest.raises(ValueError, match="no samples"):
        acc.mean_std("U")
    with pytest.raises(ValueError, match="ekin must be > 0"):
        acc.push(-600.0, 0.0, 1.0, 1.0)
    with pytest.raises(ValueError, match="params%n must be > 0"):
        stats.RunStatistics(0, 1.0)
    assert stats.fortran_1pe(-8.049620348676e-1, 19, 12) == "-8.049620348676E-01"
    assert stats.fortran_1pe(1.0e-105, 13, 6) == " 1.000000-105"
