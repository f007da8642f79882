"""-m gpu: known-answer tests of the rounding edge cases ON THE DEVICE KERNELS.

geometry_pbc.f90:54  wrap      x - L*floor(x*invL)      -> floor, result in [0, L] (x = -1e-20 gives exactly L)
geometry_pbc.f90:86  min image d - L*dnint(d*invL)      -> dnint = round half AWAY from zero (HIP round(), not rint())
The CPU oracle is pinned to the reference on these cases by tests/test_oracle_vs_golden.py (kat.json); here the
same inputs go through drift_kick_kernel (wrap + the unwrapped update's minimum image, which is NOT protected
by a cutoff) and through all three pair kernels (ties |d| = L/2, r = rc exactly, coordinates equal to 0, L,
-1e-20).
"""
import json

import numpy as np
import pytest

from ljmd_amd import Engine, init_params
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_drift_kick_kernel_wrap_and_half_box_ties_bit_exact(oracle):
    """One Verlet step with a = 0 and dt = 0.5, L = 10: r_new = r + v*dt is exact, so the wrap and the unwrapped
    update see exactly the kat.json arguments.  r and ru after the step depend on r0, v0, a0 only: bit for bit
    equal to the oracle's (= the reference's) verlet_step + caller-side unwrapped update."""
    k = json.loads((GOLDEN / "kat.json").read_text())
    L, dt, rc = 10.0, 0.5, 4.0
    # x: the edge cases; y, z keep the particles apart so that the forces evaluated inside the step stay finite
    x0 = np.array([0.0, 7.5, 2.5, -1e-20, 10.0, 9.999999999999999, -0.4, 10.2, 20.7, -10.0, 9.8, 5.0])
    vx = np.array([10.0, -10.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -10.0])
    n = len(x0)
    y0 = (np.arange(n) % 4) * 2.5 + 0.25
    z0 = (np.arange(n) // 4) * 3.0 + 0.5
    zero = np.zeros(n)
    p = init_params(n, L, dt, rc)
    with Engine(p) as eng:
        eng.set_state(x0, y0, z0, vx, zero, zero)
        eng.set_accel(zero, zero, zero)
        eng.verlet_steps(1)
        st = eng.get_state(("r", "ru"))
    po = oracle.derive_params(n, L, dt, rc)
    so = {"rx": x0.copy(), "ry": y0.copy(), "rz": z0.copy(), "ux": x0.copy(), "uy": y0.copy(), "uz": z0.copy(),
          "vx": vx.copy(), "vy": zero.copy(), "vz": zero.copy(), "ax": zero.copy(), "ay": zero.copy(), "az": zero.copy()}
    oracle.run_steps(po, 1, so)
    for name, mine, want in (("rx", st["r"][0], so["rx"]), ("ry", st["r"][1], so["ry"]), ("rz", st["r"][2], so["rz"]),
                             ("rux", st["ru"][0], so["ux"]), ("ruy", st["ru"][1], so["uy"]), ("ruz", st["ru"][2], so["uz"])):
        assert np.array_equal(np.asarray(mine).view(np.uint64), want.view(np.uint64)), (name, mine, want)
    rx, rux = np.asarray(st["r"][0]), np.asarray(st["ru"][0])
    # the reference's own known answers (kat.json, produced by the reference's geometry_pbc module)
    assert rx[3] == k["wrapz"][1] == L                      # x = -1e-20 wraps to exactly L
    assert rx[4] == k["wrapy"][2] == 0.0                    # x = L wraps to 0
    assert rx[5] == k["wrapz"][2]                           # 9.999999999999999 -> 9.999999999999998
    assert rx[6:9].tolist() == k["wrapx"]                   # -0.4, 10.2, 20.7
    assert rx[9] == k["wrapz"][0] and rx[10] == k["wrapy"][0]
    # displacement exactly +L/2 / -L/2: dnint(+-0.5) = +-1 (half away from zero), so ru moves by -+L/2 -- rint()
    # (half to even) would give the opposite sign.  kat.json "mic": minimum_image(5) = -5, minimum_image(-5) = +5
    assert rx[0] == 5.0 and rux[0] - x0[0] == k["mic"][1] == -5.0
    assert rx[1] == 2.5 and rux[1] - x0[1] == k["mic"][2] == 5.0
    assert rx[2] == 7.5 and rux[2] - x0[2] == -5.0
    assert rx[11] == 0.0 and rux[11] - x0[11] == 5.0


def _edge_configuration():
    """L = 10, rc = 4.5 (exact in binary): pairs at |d| = L/2 exactly (ties of d/L), a pair at r = rc exactly
    (strict `<`, lj_potential_energy.f90:132: excluded), coordinates equal to 0, L (a legitimate wrap result) and
    -1e-20, plus a seeded gas around them."""
    L, rc = 10.0, 4.5
    special = np.array([
        [0.0, 0.0, 0.0], [5.0, 0.0, 0.0],            # +L/2 along x
        [1.25, 5.0, 2.5], [1.25, 0.0, 2.5],          # L/2 along y
        [7.5, 7.5, 1.0], [7.5, 7.5, 6.0],            # L/2 along z
        [2.0, 3.0, 8.0], [6.5, 3.0, 8.0],            # r = rc exactly: not inside
        [10.0, 4.0, 4.0], [1.0, 4.0, 4.0],           # x = L (wrap output) next to x = 1: image distance 1
        [-1e-20, 8.0, 8.0], [9.0, 8.0, 8.0],         # x = -1e-20 next to x = 9
        [2.5, 2.5, 5.0], [7.5, 7.5, 5.0],            # L/2 on two axes at once
    ]).T
    rng = np.random.Generator(np.random.PCG64(77))
    m = 6
    g = (np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij")).reshape(3, -1) + 0.5) * (L / m)
    g = g + rng.uniform(-0.25, 0.25, g.shape) + 0.013           # keeps every pair farther apart than ~0.9
    r = np.concatenate([special, g], axis=1)
    d = r[:, :, None] - r[:, None, :]
    d -= L * np.round(d / L)
    r2 = (d * d).sum(axis=0) + np.eye(r.shape[1]) * 100
    keep = np.ones(r.shape[1], dtype=bool)
    for j in range(special.shape[1], r.shape[1]):                # drop gas particles that sit on a special one
        if r2[j, :special.shape[1]].min() < 0.64:
            keep[j] = False
    return L, rc, np.ascontiguousarray(r[:, keep])


@pytest.mark.parametrize("kernel", ["generic", "tiles", "n3_1", "n3_2", "n3_4"])
def test_pair_kernels_on_half_box_ties_and_box_edges(oracle, kernel, monkeypatch):
    L, rc, r = _edge_configuration()
    n = r.shape[1]
    if kernel == "generic":
        monkeypatch.setenv("LJMD_FORCE_GENERIC", "1")
    elif kernel == "tiles":
        monkeypatch.setenv("LJMD_N3", "0")
    else:
        monkeypatch.setenv("LJMD_N3_MIN_N", "1")
        monkeypatch.setenv("LJMD_N3_ROW_TILES", kernel[-1])
    p = init_params(n, L, 0.005, rc)
    po = oracle.derive_params(n, L, 0.005, rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    ao = np.stack([ax, ay, az])
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], r[0], r[1], r[2])
        want = {"generic": "pair_rows_generic_kernel", "tiles": "pair_tiles_kernel"}.get(kernel, "pair_n3_kernel")
        assert eng.pair_kernel_name() == want
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    assert np.isfinite(a).all()
    for name, mine, ref in (("epot", e, e_o), ("d_epot", d, d_o), ("dd_epot", dd, dd_o)):
        assert abs(mine - ref) <= 1e-13 * abs(ref), (name, mine, ref)
    assert np.abs(a - ao).max() <= 1e-12 * np.abs(ao).max()
    # the pair at r = rc exactly contributes nothing: removing particle 7 changes particle 6's force only through
    # its other neighbours -- checked through the oracle, which is bit-pinned to the reference
    assert abs(r[0, 7] - r[0, 6]) == rc


@pytest.mark.parametrize("kernel", ["generic", "tiles", "n3_1", "n3_2", "n3_4"])
def test_ragged_and_tiny_systems_vs_oracle(oracle, kernel, monkeypatch):
    """Edge sizes of every pair kernel against the oracle (= the reference's loop): n = 1 (no pair at all: the
    reference's `do i = 1, n-1` does not execute, the scalars are the tail corrections alone), n = 2, 3, tile sizes
    +-1 (63 / 64 / 65, 127 / 129, 255 / 257: padding slots in the last tile, partially filled row groups), a cutoff so
    short that no pair is inside, a cutoff just below L/2, a dilute and a dense box -- seeded random gases with a
    minimum separation so that no force overflows."""
    if kernel == "generic":
        monkeypatch.setenv("LJMD_FORCE_GENERIC", "1")
    elif kernel == "tiles":
        monkeypatch.setenv("LJMD_N3", "0")
    else:
        monkeypatch.setenv("LJMD_N3_MIN_N", "1")
        monkeypatch.setenv("LJMD_N3_ROW_TILES", kernel[-1])
    rng = np.random.Generator(np.random.PCG64(2024))
    cases = [(1, 6.0, 0.4), (2, 6.0, 0.45), (3, 5.0, 0.49), (63, 9.0, 0.49), (64, 9.0, 0.3), (65, 9.0, 0.49),
             (127, 11.0, 0.49), (129, 11.0, 0.1), (255, 14.0, 0.49), (257, 14.0, 0.4999), (300, 30.0, 0.02),
             (511, 9.5, 0.49), (513, 17.0, 0.49), (1000, 21.0, 0.35)]
    for n, L, rc_over_L in cases:
        # rejection-free placement with a minimum separation: jittered sub-lattice
        m = int(np.ceil(n ** (1.0 / 3.0)))
        cells = rng.permutation(m ** 3)[:n]
        g = np.stack(np.unravel_index(cells, (m, m, m))).astype(np.float64)
        r = np.ascontiguousarray((g + 0.5 + rng.uniform(-0.2, 0.2, g.shape)) * (L / m))
        rc = rc_over_L * L
        p = init_params(n, L, 0.005, rc)
        po = oracle.derive_params(n, L, 0.005, rc)
        e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
        ao = np.stack([ax, ay, az])
        with Engine(p) as eng:
            eng.set_state(r[0], r[1], r[2], r[0], r[1], r[2])
            e, d, dd = eng.compute_forces()
            a = np.stack(eng.get_state(("a",))["a"])
            # two steps as well: drift / wrap / kick on ragged shards, scalars against the oracle's verlet_step
            v = rng.uniform(-0.5, 0.5, r.shape)
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            sc = np.stack(eng.verlet_steps(2), axis=1)
        st = {"rx": r[0].copy(), "ry": r[1].copy(), "rz": r[2].copy(), "ux": r[0].copy(), "uy": r[1].copy(),
              "uz": r[2].copy(), "vx": v[0].copy(), "vy": v[1].copy(), "vz": v[2].copy(),
              "ax": ax.copy(), "ay": ay.copy(), "az": az.copy()}
        sc_o = oracle.run_steps(po, 2, st)
        tag = (kernel, n, L, rc_over_L)
        for name, mine, ref in (("epot", e, e_o), ("d_epot", d, d_o), ("dd_epot", dd, dd_o)):
            assert abs(mine - ref) <= 1e-13 * max(abs(ref), 1e-300) + 1e-15, (tag, name, mine, ref)
        assert np.abs(a - ao).max() <= 1e-12 * max(np.abs(ao).max(), 1e-3), tag
        assert np.max(np.abs(sc - sc_o) / np.maximum(np.abs(sc_o), 1e-12)) < 1e-10, (tag, sc, sc_o)
    if kernel == "tiles":
        # n = 1: nothing but the tail terms (lj_potential_energy.f90:205-223)
        p1 = init_params(1, 6.0, 0.005, 2.4)
        with Engine(p1) as eng:
            z = np.array([1.0])
            eng.set_state(z, z, z, z, z, z)
            assert eng.compute_forces() == oracle.tail_corrections(oracle.derive_params(1, 6.0, 0.005, 2.4))
