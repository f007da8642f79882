"""Trajectory analysis on rva.dat snapshots: the host-side mirror of the reference's
`compute_rdf`, `compute_msd_tau_timeorig`, `compute_vacf_tau_timeorig`
(scripts/md_one_run_analysis.py:404-595).  SURVEY.md 8(f) #3.

The O(n^2) pair pass of the RDF runs on the GPU (`ljmd_rdf_histogram`, integer histogram,
bit-exact); with `subsample=True` the reference's sub-sampling (<= 200 snapshots, <= 800
particles, chosen with np.linspace) is applied first, so the result equals the reference's
for the same input; `subsample=False` uses every particle of every snapshot -- the case the
reference cannot afford in numpy.  MSD / VACF: `time_origin_average_gpu` (the O(n_snap^2 n) sums in a HIP kernel) beside
the numpy mirror of the reference's functions (bit-identical to the reference module; the CPU tests' checker).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import c_double_p


def rdf_histogram(x, y, z, L: float, nbins: int, rmax: float, hist: np.ndarray) -> None:
    """Adds one snapshot's ordered-pair distance counts to hist (uint64[nbins]) on the GPU."""
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z)]
    assert hist.dtype == np.uint64 and hist.shape == (nbins,) and hist.flags.c_contiguous
    _lib.check(_lib.load().ljmd_rdf_histogram(len(arrs[0]), *[a.ctypes.data_as(c_double_p) for a in arrs],
                                              float(L), int(nbins), float(rmax),
                                              hist.ctypes.data_as(C.POINTER(C.c_uint64))))


def compute_rdf(rx: np.ndarray, ry: np.ndarray, rz: np.ndarray, L: float, nbins: int = 200,
                rmax: float | None = None, subsample: bool = True, histogram=rdf_histogram):
    """-> (r_centers, g_r); arrays rx, ry, rz of shape (n_snap, n), wrapped positions."""
    n_snap, n = rx.shape
    if rmax is None:
        rmax = 0.5 * L
    snap_idx = np.arange(n_snap)
    part_idx = np.arange(n)
    if subsample:
        if n_snap > 200:
            snap_idx = np.linspace(0, n_snap - 1, 200, dtype=int)
        if n > 800:
            part_idx = np.linspace(0, n - 1, 800, dtype=int)
    n_eff = len(part_idx)
    if n_eff < 2:
        raise ValueError("Not enough particles for RDF after subsampling.")
    counts = np.zeros(nbins, dtype=np.uint64)
    for s in snap_idx:
        histogram(rx[s, part_idx], ry[s, part_idx], rz[s, part_idx], L, nbins, rmax, counts)
    hist = counts.astype(np.float64)               # each unordered pair contributed 2 (i-j and j-i)
    vol = L ** 3
    rho = n_eff / vol
    r_edges = np.linspace(0.0, rmax, nbins + 1)
    r_centers = 0.5 * (r_edges[:-1] + r_edges[1:])
    shell_vol = (4.0 / 3.0) * math.pi * (r_edges[1:] ** 3 - r_edges[:-1] ** 3)
    norm = len(snap_idx) * n_eff * rho * shell_vol
    g = np.zeros_like(r_centers)
    mask = norm > 0
    g[mask] = hist[mask] / norm[mask]
    return r_centers, g


def time_origin_average_gpu(kind: int, x, y, z, max_lag=None, origin_stride: int = 1) -> np.ndarray:
    """MSD (kind 0: unwrapped positions) / VACF (kind 1: velocities) on the GPU (`ljmd_time_origin_average`): the
    O(n_snap^2 n) particle means per (origin, lag) in a HIP kernel, the origins added on the host in the reference's order.
    Equal to compute_msd_tau_timeorig / compute_vacf_tau_timeorig below to rounding (numpy's mean is a pairwise sum)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z)]
    n_snap, n = arrs[0].shape
    if n_snap < 2:
        return compute_msd_tau_timeorig(*arrs) if kind == 0 else compute_vacf_tau_timeorig(*arrs)
    max_lag = n_snap - 1 if max_lag is None else int(min(max_lag, n_snap - 1))
    out = np.empty(max_lag + 1, dtype=np.float64)
    _lib.check(_lib.load().ljmd_time_origin_average(kind, n_snap, n, *[a.ctypes.data_as(c_double_p) for a in arrs], max_lag,
                                                    max(1, int(origin_stride)), out.ctypes.data_as(c_double_p)))
    return out


def _time_origin_average(series, max_lag, origin_stride, term):
    n_snap = series[0].shape[0]
    if max_lag is None:
        max_lag = n_snap - 1
    max_lag = int(min(max_lag, n_snap - 1))
    origin_stride = max(1, int(origin_stride))
    acc = np.zeros(max_lag + 1, dtype=np.float64)
    counts = np.zeros(max_lag + 1, dtype=np.int64)
    for t0 in range(0, n_snap - 1, origin_stride):
        lag = min(max_lag, (n_snap - 1) - t0)
        if lag <= 0:
            continue
        acc[:lag + 1] += term(t0, lag)
        counts[:lag + 1] += 1
    mask = counts > 0
    acc[mask] /= counts[mask]
    return acc


def compute_msd_tau_timeorig(rux, ruy, ruz, max_lag=None, origin_stride: int = 1) -> np.ndarray:
    """MSD(tau) = <|ru(t+tau) - ru(t)|^2> over particles and time origins (md_one_run_analysis.py:404-441)."""
    if rux.shape[0] < 2:
        return np.array([0.0], dtype=np.float64)

    def term(t0, lag):
        dx = rux[t0:t0 + lag + 1, :] - rux[t0, :][None, :]
        dy = ruy[t0:t0 + lag + 1, :] - ruy[t0, :][None, :]
        dz = ruz[t0:t0 + lag + 1, :] - ruz[t0, :][None, :]
        return np.mean(dx * dx + dy * dy + dz * dz, axis=1)

    return _time_origin_average((rux,), max_lag, origin_stride, term)


def compute_vacf_tau_timeorig(vx, vy, vz, max_lag=None, origin_stride: int = 1) -> np.ndarray:
    """VACF(tau) = <v(t) . v(t+tau)> over particles and time origins (md_one_run_analysis.py:444-489)."""
    if vx.shape[0] < 2:
        return np.array([np.mean(vx[0] * vx[0] + vy[0] * vy[0] + vz[0] * vz[0])], dtype=np.float64)

    def term(t0, lag):
        return np.mean(vx[t0:t0 + lag + 1, :] * vx[t0, :][None, :] + vy[t0:t0 + lag + 1, :] * vy[t0, :][None, :]
                       + vz[t0:t0 + lag + 1, :] * vz[t0, :][None, :], axis=1)

    return _time_origin_average((vx,), max_lag, origin_stride, term)
