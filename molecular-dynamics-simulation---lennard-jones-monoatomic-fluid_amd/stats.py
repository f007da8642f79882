"""Scalar statistics of a production run -- Python mirror of fortran/md_stats.f90 and
fortran/md_run_outputs.f90, used by `simulation.run_md_simulation`.

Reference behaviour reproduced (all `scripts/...` paths are the reference's):
  running means / population std of 11 per-sample quantities   stats/md_means.f90:215-270, :311-364
  centred autocovariance C(lag), C(lag)/C(0)                   stats/stats_math.f90:129-149, :160-190
  block-averaged curves over <= 5 contiguous blocks            stats/md_correlations.f90:692-799
  microcanonical thermodynamic coefficients                    physics/thermodynamic_coefs.f90:104-203
  corr_*.dat / corrmean_*.dat / md_final_results.txt           md_simulation_program.f90:419-560, :594-634

Sums are accumulated left to right in Python floats (IEEE fp64), like the reference's
scalar accumulators; the dot products use the same left-to-right order, so the files come out
digit for digit equal to the reference's when fed the same samples (tests/test_stats.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import sqrt
from pathlib import Path

OBSERVABLES = ("epot", "ekin", "etot", "temp", "press")
QUANTITIES = ("U", "K", "E", "T", "P", "Kinv", "dU", "ddU", "dU_Kinv", "ddU_Kinv", "dU2_Kinv")


def fortran_1pe(x: float, width: int, digits: int) -> str:
    """Edit descriptor 1pe<width>.<digits>."""
    mant, exp = f"{x:.{digits}E}".split("E")
    e = int(exp)
    s = f"{mant}E{e:+03d}" if abs(e) < 100 else f"{mant}{e:+04d}"
    return s.rjust(width) if len(s) <= width else "*" * width


@dataclass
class RunStatistics:
    n_particles: int
    volume: float
    n_samples: int = 0
    s1: dict = field(default_factory=lambda: dict.fromkeys(QUANTITIES, 0.0))
    s2: dict = field(default_factory=lambda: dict.fromkeys(QUANTITIES, 0.0))
    series: dict = field(default_factory=lambda: {k: [] for k in OBSERVABLES})

    def __post_init__(self):
        if self.n_particles <= 0:
            raise ValueError("md_means_init(): params%n must be > 0.")
        if not self.volume > 0.0:
            raise ValueError("md_means_init(): params%volume must be > 0.")

    def push(self, epot: float, ekin: float, d_epot: float, dd_epot: float):
        """One sampling instant -> (T, P); T = 2K/(3N), P = rho T + W/(3V), W = -d_epot."""
        epot, ekin, d_epot, dd_epot = float(epot), float(ekin), float(d_epot), float(dd_epot)
        npd = float(self.n_particles)
        rho = npd / self.volume
        temp = 2.0 * ekin / (3.0 * npd)
        press = rho * temp + (-d_epot) / (3.0 * self.volume)
        if ekin <= 0.0:
            raise ValueError("md_means_add_sample(): ekin must be > 0 to accumulate 1/ekin terms.")
        kinv = 1.0 / ekin
        q = dict(U=epot, K=ekin, E=epot + ekin, T=temp, P=press, Kinv=kinv, dU=d_epot, ddU=dd_epot,
                 dU_Kinv=d_epot * kinv, ddU_Kinv=dd_epot * kinv, dU2_Kinv=(d_epot * d_epot) * kinv)
        for k, val in q.items():
            self.s1[k] = self.s1[k] + val
            self.s2[k] = self.s2[k] + val * val
        self.n_samples += 1
        for k, val in zip(OBSERVABLES, (epot, ekin, epot + ekin, temp, press)):
            self.series[k].append(val)
        return temp, press

    def mean_std(self, which: str):
        if self.n_samples <= 0:
            raise ValueError("md_means_get(): no samples accumulated.")
        inv_ns = 1.0 / float(self.n_samples)
        mean = self.s1[which] * inv_ns
        m2 = self.s2[which] * inv_ns
        return mean, sqrt(max(0.0, m2 - mean * mean))


def lag_limit(n_samples: int) -> int:
    """min(1000, n-1, n/2); -1 = fewer than 2 samples (md_simulation_program.f90:280-288)."""
    return -1 if n_samples < 2 else min(1000, n_samples - 1, n_samples // 2)


def autocovariance(x, lag_max: int):
    n = len(x)
    if n <= 0:
        raise ValueError("autocorr_scalar_centered(): n_samples must be > 0.")
    if lag_max < 0:
        raise ValueError("autocorr_scalar_centered(): lag_max must be >= 0.")
    if lag_max >= n:
        raise ValueError("autocorr_scalar_centered(): lag_max must be < n_samples.")
    total = 0.0
    for v in x:
        total = total + v
    m = total / float(n)
    d = [v - m for v in x]
    out = []
    for lag in range(lag_max + 1):
        nv = n - lag
        acc = 0.0
        for k in range(nv):
            acc = acc + d[k] * d[k + lag]
        out.append(acc / float(nv))
    return out


def normalise_by_lag0(c):
    if abs(c[0]) <= 1.0e-14:
        return [0.0] * len(c)
    return [v / c[0] for v in c]


def block_mean_autocovariance(x, n_blocks: int, lag_max: int):
    if n_blocks <= 0:
        raise ValueError("md_corr_cm_compute(): invalid number of blocks.")
    block_len = len(x) // n_blocks
    if block_len <= 0:
        raise ValueError("md_corr_cm_compute(): block_len <= 0 (too many blocks).")
    if lag_max >= block_len:
        raise ValueError("md_corr_cm_compute(): max_lag must be < block_len.")
    c_sum = [0.0] * (lag_max + 1)
    cn_sum = [0.0] * (lag_max + 1)
    for b in range(n_blocks):
        c = autocovariance(x[b * block_len:(b + 1) * block_len], lag_max)
        cn = normalise_by_lag0(c)
        c_sum = [a + v for a, v in zip(c_sum, c)]
        cn_sum = [a + v for a, v in zip(cn_sum, cn)]
    inv_nb = 1.0 / float(n_blocks)
    return [v * inv_nb for v in c_sum], [v * inv_nb for v in cn_sum]


def thermo_coefficients(st: RunStatistics) -> dict:
    """thermodynamic_coefs.f90:104-203; f = 3N - 3."""
    k_mean, p_mean = st.mean_std("K")[0], st.mean_std("P")[0]
    kinv, du, ddu = st.mean_std("Kinv")[0], st.mean_std("dU")[0], st.mean_std("ddU")[0]
    du_kinv, du2_kinv = st.mean_std("dU_Kinv")[0], st.mean_std("dU2_Kinv")[0]
    vol, npd = st.volume, float(st.n_particles)
    f = 3.0 * npd - 3.0
    if f <= 0.0:
        raise ValueError("thermodynamic_compute(): degrees_of_freedom <= 0 (check N).")
    a1 = 1.0 - 2.0 / f
    a2 = f / 2.0 - 1.0

    def inv(denom, what):
        if abs(denom) < 1.0e-14:
            raise ValueError(f"thermodynamic_compute(): {what}")
        return 1.0 / denom

    o = {}
    o["temperature"] = 2.0 * k_mean / f
    o["pressure"] = p_mean
    o["Ca_v"] = inv(1.0 - a1 * k_mean * kinv, "Ca_v denominator ~ 0 (numerical instability).")
    o["Ce_v"] = o["Ca_v"] / npd
    if abs(o["Ce_v"]) < 1.0e-14:
        raise ValueError("thermodynamic_compute(): Ce_v ~ 0 (check inputs).")
    o["gamma"] = 1.0 / o["Ce_v"] + (a2 / 3.0) * (du * kinv - du_kinv)
    ks_aux = ((npd * o["temperature"] * (1.0 + 2.0 * o["gamma"] - 1.0 / o["Ce_v"])) / vol) \
        + (ddu - 2.0 * du) / (9.0 * vol)
    o["K_S"] = ks_aux - (a2 * (du2_kinv - 2.0 * du * du_kinv + (du * du) * kinv)) / (9.0 * vol * vol)
    o["K_S_inv"] = inv(o["K_S"], "K_S ~ 0 (cannot invert).")
    o["K_T"] = o["K_S"] - (o["temperature"] * o["Ca_v"] * (o["gamma"] * o["gamma"])) / vol
    o["K_T_inv"] = inv(o["K_T"], "K_T ~ 0 (cannot invert / compute Cp, alpha_P).")
    o["Ca_p"] = o["Ca_v"] * (o["K_S"] / o["K_T"])
    o["Ce_p"] = o["Ca_p"] / npd
    o["alpha_E1"] = inv((o["pressure"] * vol / o["Ca_v"]) - (o["gamma"] * o["temperature"]), "alpha_E1 denominator ~ 0.")
    o["alpha_E2"] = inv((1.0 / 3.0) * (a1 * k_mean * du_kinv - du), "alpha_E2 denominator ~ 0.")
    o["alpha_S"] = -inv(o["gamma"] * o["temperature"], "gamma*T ~ 0 (alpha_S undefined).")
    o["alpha_P"] = (o["Ca_v"] * o["gamma"]) / vol * o["K_T_inv"]
    return o


def _write_curve(path: Path, header: str, c, cn) -> None:
    with open(path, "w") as f:
        f.write(header + "\n")
        for lag, (a, b) in enumerate(zip(c, cn)):
            f.write(f"{lag:8d}  {fortran_1pe(a, 19, 12)}  {fortran_1pe(b, 19, 12)}\n")


def write_run_statistics(out_dir, params, total_steps: int, output_interval: int, warmup_steps: int,
                         st: RunStatistics) -> dict:
    """Writes corr_*.dat, corrmean_*.dat and appends the md_final_results.txt block; returns the
    numbers written (means, stds, coefficients)."""
    out_dir = Path(out_dir)
    if st.n_samples <= 0:
        raise ValueError("md_simulation: no samples were taken (check warmup_steps/output_interval).")
    tc = thermo_coefficients(st)
    lag_max = lag_limit(st.n_samples)
    if lag_max >= 0:
        for k in OBSERVABLES:
            c = autocovariance(st.series[k], lag_max)
            _write_curve(out_dir / f"corr_{k}.dat", "# lag   C(lag)   C_norm(lag)", c, normalise_by_lag0(c))
        n_blocks = min(5, st.n_samples // (lag_max + 1))
        if n_blocks >= 1:
            for k in OBSERVABLES:
                c, cn = block_mean_autocovariance(st.series[k], n_blocks, lag_max)
                _write_curve(out_dir / f"corrmean_{k}.dat", "# lag   <C(lag)>_blocks   <C_norm(lag)>_blocks", c, cn)

    ms = {k: st.mean_std(k) for k in ("U", "K", "E", "T", "P")}
    e19 = lambda v: fortran_1pe(v, 19, 12)  # noqa: E731
    pair = lambda a, x, b, y: f"{a} {e19(x)}  {b} {e19(y)}"  # noqa: E731
    lines = ["************** MD PRODUCTION RESULTS **************",
             f"num_particles: {params.n:8d}", f"num_cells: {params.num_cells:8d}",
             f"box_length: {e19(params.box_length)}", f"volume: {e19(params.volume)}",
             f"density: {e19(float(params.n) / params.volume)}", f"time_step: {e19(params.dt)}",
             f"output_interval: {output_interval:8d}", f"total_steps: {total_steps:10d}",
             f"warmup_steps: {warmup_steps:10d}",
             "-------------------- Averages --------------------"]
    for label, k in (("<Epot>:", "U"), ("<Ekin>:", "K"), ("<Etot>:", "E"), ("<T>   :", "T"), ("<P>   :", "P")):
        lines.append(pair(label, ms[k][0], "std:", ms[k][1]))
    lines += ["-------------- Thermodynamic coefficients --------------",
              pair("Temperature:", tc["temperature"], "Pressure:", tc["pressure"]),
              pair("Ca_v:", tc["Ca_v"], "Ce_v:", tc["Ce_v"]),
              pair("Ca_p:", tc["Ca_p"], "Ce_p:", tc["Ce_p"]),
              pair("kappa_S:", tc["K_S_inv"], "kappa_T:", tc["K_T_inv"]),
              f"Gamma: {e19(tc['gamma'])}",          # format reversion of the reference's 3-item write (:555)
              pair("Alpha_E1:", tc["alpha_E1"], "Alpha_E2:", tc["alpha_E2"]),
              pair("Alpha_S:", tc["alpha_S"], "Alpha_P:", tc["alpha_P"]),
              "--------------------------------------------------------", ""]
    with open(out_dir / "md_final_results.txt", "a") as f:     # access='append' (:532)
        f.write("\n".join(lines) + "\n")
    return dict(means={k: v[0] for k, v in ms.items()}, stds={k: v[1] for k, v in ms.items()}, coefficients=tc)
