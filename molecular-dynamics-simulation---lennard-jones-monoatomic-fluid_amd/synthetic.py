"""Synthetic initial configurations of the BASELINE.json configs.

All at rho = 0.8 (L = (N/0.8)^(1/3)), rc = 0.49 L, dt = 0.005, reduced units.
Positions: a lattice plus a seeded uniform jitter of +-5 % of the lattice spacing (a perfect
lattice has |a| ~ 1e-14, so the jitter makes the forces non-trivial from step 0);
velocities uniform(-0.5, 0.5), centre-of-mass velocity removed, scaled to T = 1 (K = 1.5 N).
Deterministic: numpy PCG64 stream with a fixed seed.
"""
from __future__ import annotations

import numpy as np

from .md_types import SimParams, init_params

RHO = 0.8
RC_OVER_L = 0.49
DT = 0.005
SEED = 20240601


def box_length(n: int, rho: float = RHO) -> float:
    return float((n / rho) ** (1.0 / 3.0))


def simple_cubic(n: int, L: float) -> np.ndarray:
    """First n sites of the smallest m^3 >= n simple-cubic lattice (cell centres), [3, n]."""
    m = int(np.ceil(n ** (1.0 / 3.0) - 1e-9))
    idx = np.arange(n)
    ix, iy, iz = idx // (m * m), (idx // m) % m, idx % m
    a = L / m
    return np.stack([(ix + 0.5) * a, (iy + 0.5) * a, (iz + 0.5) * a]).astype(np.float64), a


def fcc(num_cells: int, L: float) -> np.ndarray:
    """FCC sites in the reference's order (md_initial_config_program.f90:144-178), [3, 4k^3]."""
    a = L / num_cells
    g = np.arange(num_cells, dtype=np.float64)
    x0, y0, z0 = np.meshgrid(g * a, g * a, g * a, indexing="ij")
    cell = np.stack([x0.ravel(), y0.ravel(), z0.ravel()])            # ix outer ... iz inner
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], dtype=np.float64).T * a
    pos = cell[:, :, None] + basis[:, None, :]                        # [3, cells, 4]
    return pos.reshape(3, -1), a / np.sqrt(2.0)


def make_config(n: int, lattice: str = "auto", seed: int = SEED, temperature: float = 1.0,
                jitter: float = 0.05, rho: float = RHO, rc_over_L: float = RC_OVER_L, dt: float = DT):
    """-> (SimParams, r[3, n], v[3, n]) ; r wrapped into [0, L)."""
    L = box_length(n, rho)
    params: SimParams = init_params(n, L, dt, rc_over_L * L)
    rng = np.random.Generator(np.random.PCG64(seed))
    if lattice == "auto":
        k = round((n / 4) ** (1.0 / 3.0))
        m = round(n ** (1.0 / 3.0))
        lattice = "sc" if m ** 3 == n else ("fcc" if 4 * k ** 3 == n else "sc")
    if lattice == "fcc":
        k = round((n / 4) ** (1.0 / 3.0))
        assert 4 * k ** 3 == n, "fcc needs n = 4 k^3"
        r, spacing = fcc(k, L)
    else:
        r, spacing = simple_cubic(n, L)
    r = r + (rng.random((3, n)) - 0.5) * (2.0 * jitter * spacing)
    r = r - L * np.floor(r / L)
    r[r >= L] = 0.0
    v = rng.random((3, n)) - 0.5
    v -= v.mean(axis=1, keepdims=True)
    ke = 0.5 * np.sum(v * v)
    v *= np.sqrt(1.5 * n * temperature / ke)
    return params, np.ascontiguousarray(r), np.ascontiguousarray(v)
