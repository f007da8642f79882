"""-m gpu: the HIP path, called through the C ABI (ctypes -> libljmd.so), against
(1) golden vectors produced by the real reference and (2) the pinned C oracle.

Tolerances:
  single force call : scalars <= 1e-13 relative, per-particle a <= 1e-12 * max|a|
  short trajectory  : Etot, T, P <= 1e-10 relative over the first 200 steps (inside the
                      chaos horizon; the reference diverges from ITSELF beyond that when
                      only the summation order / FMA contraction changes -- SURVEY fact #6)
  integrator        : with identical accelerations the drift/kick arithmetic is bit-exact
The GPU sums pairs in a different order than the sequential i<j loop, so bitwise identity
with the reference is not attainable and not claimed.
"""
import os

import numpy as np
import pytest

import ljmd_amd
from ljmd_amd import Engine, init_params, init_state, synthetic
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

REL_SCALAR = 1e-13
REL_ACCEL = 1e-12
REL_TRAJ = 1e-10


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def check_force(scalars, a, g):
    gs = g["scalars"]
    for name, x, y in zip(("epot", "d_epot", "dd_epot"), scalars, gs):
        assert rel(x, y) <= REL_SCALAR, (name, x, y, rel(x, y))
    amax = np.max(np.abs(g["a"]))
    # a perfect lattice has |a| ~ 1e-14 (pure cancellation noise): floor the scale at 1
    assert np.max(np.abs(a - g["a"])) <= REL_ACCEL * max(amax, 1.0), np.max(np.abs(a - g["a"])) / amax


@pytest.mark.parametrize("name", ["force_n108", "force_n500", "force_n4000", "force_n4096"])
def test_force_call_vs_reference_golden(golden, name, monkeypatch):
    """The gather (full ordered matrix) tile kernel, which serves systems below 4096 particles."""
    monkeypatch.setenv("LJMD_N3", "0")
    g = golden(name)
    n = int(g["n"])
    p = init_params(n, float(g["L"]), 0.005, float(g["rc"]))
    r = g["r"]
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], r[0], r[1], r[2])
        sc = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    check_force(sc, a, g)


@pytest.mark.parametrize("target", ["0", "1000000"])
@pytest.mark.parametrize("row_tiles", ["1", "2", "4"])
@pytest.mark.parametrize("name", ["force_n108", "force_n500", "force_n4000", "force_n4096"])
def test_newton3_kernel_vs_reference_golden(golden, name, row_tiles, target, monkeypatch):
    """The Newton-3 rotation kernel normally engages at N >= 4096; force it at the golden sizes in all three
    instantiations (1, 2, 4 tiles per row group), with the library's own work items and with one pass per work item
    (N3Args::uchunk = 1).  With 4 tiles per group: N=108 -> one row group (diagonal only), N=500 -> two groups (the
    d = NG/2 tie, worked from both sides), N=4000/4096 -> 16 groups, 9 offsets."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    monkeypatch.setenv("LJMD_N3_ROW_TILES", row_tiles)
    if target != "0":
        monkeypatch.setenv("LJMD_N3_TARGET_WAVES", target)
    g = golden(name)
    n = int(g["n"])
    p = init_params(n, float(g["L"]), 0.005, float(g["rc"]))
    r = g["r"]
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], r[0], r[1], r[2])
        sc = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
        sc2 = eng.compute_forces()                      # second call: slab flags must be rewritten
        a2 = np.stack(eng.get_state(("a",))["a"])
    check_force(sc, a, g)
    assert sc == sc2 and np.array_equal(a, a2)


def test_newton3_short_trajectory(golden, monkeypatch):
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    g = golden("traj_n4096_200")
    p = init_params(4096, float(g["L"]), float(g["dt"]), float(g["rc"]))
    ref = g["scalars"][:201]
    with Engine(p) as eng:
        s0 = _start(eng, g)
        e, k, d, dd = eng.verlet_steps(200)
    mine = np.vstack([s0, np.stack([e, k, d, dd], axis=1)])
    for nm, a, b in zip(("etot", "T", "P"), _series(p, mine), _series(p, ref)):
        assert np.max(np.abs(a - b) / np.abs(b)) <= REL_TRAJ, nm


def test_production_kernel_configuration_vs_oracle_n32768(oracle):
    """The default kernel set exactly as bench.py runs it (k-d ordering, tile-pair skipping -- 512 tiles,
    rc = 16.9 in a 34.5 box, so the mask really drops tile pairs --, uniform-image variants, Newton-3,
    Halley reciprocal) against the pinned C oracle on EVERY particle; the oracle takes ~5 s here."""
    n = 32768
    p, r, v = synthetic.make_config(n, seed=77)
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.pair_kernel_name() == "pair_n3_kernel"
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    a_o = np.stack([ax, ay, az])
    # 2.7e8 in-cutoff terms: the oracle's own sequential running sums carry ~sqrt(n_terms) * 1e-16 = 2e-12
    # of rounding noise, so the single-call scalar bound of the N <= 4096 fixtures (1e-13) scales to 1e-12
    assert rel(e, e_o) <= 1e-12 and rel(d, d_o) <= 1e-12 and rel(dd, dd_o) <= 1e-12
    assert np.abs(a - a_o).max() <= REL_ACCEL * np.abs(a_o).max()


@pytest.mark.parametrize("knobs", [{}, {"LJMD_N3_CLUSTERS": "0"}, {"LJMD_N3_PERTILE": "0"}])
def test_cluster_passes_and_per_tile_images_vs_oracle_n32768(oracle, knobs, monkeypatch):
    """The boundary machinery of the production kernel at a size the oracle checks on EVERY particle: 4-tile row groups
    forced at n = 32768 (they are the default from 65536), so that cluster passes (column tile sorted along the
    row->column direction, (row tile, cluster) skips by projection) and per-tile periodic images (row tiles shifted by
    +-L for a pass) are on; 12 steps WITHOUT re-sort from a start with lattice planes on the box faces, so that tiles
    straddle faces and the image logic is exercised, then forces and scalars at the engine's own positions.  A wrongly
    skipped pair inside the cutoff would show as an error of 1e-3; the bounds are the usual 1e-12.  Knobs: both
    features on (default), each one off (the answers must agree with the oracle either way)."""
    monkeypatch.setenv("LJMD_N3_ROW_TILES", "4")
    monkeypatch.setenv("LJMD_RESORT_EVERY", "1000")
    for k, val in knobs.items():
        monkeypatch.setenv(k, val)
    n = 32768
    p, r, v = synthetic.make_config(n, seed=5)
    L = p.box_length
    r = (r + 0.5 * L / 32.0) % L
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0, d0, dd0 = eng.compute_forces()
        a0 = np.stack(eng.get_state(("a",))["a"])
        e, k_, d, dd = eng.verlet_steps(12)
        st = eng.get_state(("r", "a"))
    eo, do, ddo, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    assert rel(e0, eo) <= 1e-12 and rel(d0, do) <= 1e-12 and rel(dd0, ddo) <= 1e-12
    assert np.abs(a0 - np.stack([ax, ay, az])).max() <= REL_ACCEL * np.abs(ax).max()
    rr, a = np.stack(st["r"]), np.stack(st["a"])
    eo, do, ddo, ax, ay, az = oracle.compute_forces(po, rr[0].copy(), rr[1].copy(), rr[2].copy())
    assert rel(e[-1], eo) <= 1e-12 and rel(d[-1], do) <= 1e-12 and rel(dd[-1], ddo) <= 1e-12
    assert np.abs(a - np.stack([ax, ay, az])).max() <= REL_ACCEL * np.abs(ax).max()


@pytest.mark.parametrize("mode", [0, 1])
def test_tiles_that_straddle_a_box_face_vs_oracle_n32768(oracle, mode, monkeypatch):
    """Between two re-sorts particles cross the faces of the box and reappear at the other end of the wrapped interval
    while they stay in their tile.  The Newton-3 kernels read a tile-coherent copy of the positions (every tile in the
    periodic image of its first particle, tile_boxes_kernel) so that such a tile keeps a compact box.  12 steps without
    any re-sort from a start shifted so that a lattice plane lies on every face, then the resident accelerations and
    scalars against the pinned oracle evaluated at the engine's own (wrapped) positions -- fp64 and mixed precision."""
    monkeypatch.setenv("LJMD_RESORT_EVERY", "1000")
    n = 32768
    p, r, v = synthetic.make_config(n, seed=5)
    L = p.box_length
    r = (r + 0.5 * L / 32.0) % L                     # simple-cubic planes (spacing L / 32) onto the faces
    with Engine(p, precision_mode=mode) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        e, k, d, dd = eng.verlet_steps(12)
        st = eng.get_state(("r", "ru", "a"))
    rr, ru, a = np.stack(st["r"]), np.stack(st["ru"]), np.stack(st["a"])
    crossed = np.count_nonzero(np.abs(rr - ru) > 0.5 * L)        # wrapped != unwrapped: went through a face
    assert crossed > 500, crossed                                # ... spread over most of the 96 face tiles
    assert rr.min() >= 0.0 and rr.max() < L
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, rr[0].copy(), rr[1].copy(), rr[2].copy())
    a_o = np.stack([ax, ay, az])
    if mode == 0:
        assert rel(e[-1], e_o) <= 1e-12 and rel(d[-1], d_o) <= 1e-12 and rel(dd[-1], dd_o) <= 1e-12
        assert np.abs(a - a_o).max() <= REL_ACCEL * np.abs(a_o).max()
    else:       # the bounds of the mixed-precision parity test at n = 262144
        assert rel(e[-1], e_o) <= 5e-9 and rel(d[-1], d_o) <= 5e-9 and rel(dd[-1], dd_o) <= 5e-9
        assert np.abs(a - a_o).max() <= 1e-9 * np.abs(a_o).max()


def test_liquid_state_forces_vs_oracle_n65536(oracle):
    """One force evaluation in the equilibrated LIQUID (300 steps from the jittered lattice: about a fifth of the passes with a
    straddling axis, boundary passes cluster by cluster, rc = 21.3 sigma so that the fp32 far kernel has very far passes, most
    of them next to passes that are not) against the pinned oracle on every particle at the engine's own wrapped positions:
    fp64 at the single-call bounds, then the same configuration in the mixed-precision mode at its written bounds."""
    from ljmd_amd import _lib
    n = 65536
    p, r, v = synthetic.make_config(n, seed=23)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        e, k, d, dd = eng.verlet_steps(300)
        st = eng.get_state(("r", "v", "a"))
    rr, vv, a = np.stack(st["r"]), np.stack(st["v"]), np.stack(st["a"])
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, rr[0].copy(), rr[1].copy(), rr[2].copy())
    a_o = np.stack([ax, ay, az])
    assert rel(e[-1], e_o) <= 1e-12 and rel(d[-1], d_o) <= 1e-12 and rel(dd[-1], dd_o) <= 1e-12
    assert np.abs(a - a_o).max() <= REL_ACCEL * np.abs(a_o).max()
    with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
        eng.set_state(rr[0], rr[1], rr[2], vv[0], vv[1], vv[2])
        em, dm, ddm = eng.compute_forces()
        am = np.stack(eng.get_state(("a",))["a"])
    # (cutoff-edge pairs, see test_mixed_precision_far_pass_beside_lds_combining_workgroups: 24 rc^-7 = 1.2e-8 at rc = 21.3)
    edge = 24.0 * abs(p.rc ** -7 - 2.0 * p.rc ** -13)
    print("liquid n = 65536, mixed vs oracle:", rel(em, e_o), rel(dm, d_o), rel(ddm, dd_o), np.abs(am - a_o).max() / np.abs(a_o).max())
    assert rel(em, e_o) <= 5e-9 and rel(dm, d_o) <= 5e-9 and rel(ddm, dd_o) <= 5e-9
    assert np.abs(am - a_o).max() <= 1e-9 * np.abs(a_o).max() + 2.0 * edge


@pytest.mark.parametrize("n,rc_over_L,n3", [(4096, 0.15, True), (4096, 0.15, False), (32768, 0.08, True), (2500, 0.30, True)])
def test_short_cutoff_most_tile_pairs_skipped(oracle, n, rc_over_L, n3, monkeypatch):
    """rc well below L/2 (allowed by the reference: 0 < rc_over_L <= 0.5): most tile pairs are masked
    out, whole offset ranges of the Newton-3 kernel see no work.  Against the pinned oracle."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1" if n3 else "100000000")
    p, r, v = synthetic.make_config(n, seed=31, rc_over_L=rc_over_L)
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    st = {"rx": r[0].copy(), "ry": r[1].copy(), "rz": r[2].copy(), "ux": r[0].copy(), "uy": r[1].copy(),
          "uz": r[2].copy(), "vx": v[0].copy(), "vy": v[1].copy(), "vz": v[2].copy(), "ax": ax.copy(), "ay": ay.copy(), "az": az.copy()}
    nsteps = 12 if n <= 4096 else 2
    sc_o = oracle.run_steps(po, nsteps, st)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
        sc = np.stack(eng.verlet_steps(nsteps), axis=1)
    assert rel(e, e_o) <= 1e-12 and rel(d, d_o) <= 1e-12 and rel(dd, dd_o) <= 1e-12
    assert np.abs(a - np.stack([ax, ay, az])).max() <= REL_ACCEL * np.abs(ax).max()
    assert np.max(np.abs(sc - sc_o) / np.abs(sc_o)) < 1e-10


@pytest.mark.parametrize("n", [1, 2, 4, 10, 65, 257])
def test_tiny_systems_vs_oracle(oracle, n):
    """Edge sizes: one particle (no pair, tail terms only), fewer particles than a tile, one more than a
    tile / a workgroup (padding slots in the middle of the machinery)."""
    p, r, v = synthetic.make_config(max(n, 2), seed=100 + n, rho=0.5, dt=0.002)   # jittered lattice sites, no overlaps
    if n == 1:                                     # a single particle: no pairs at all, only the tail constants
        p = init_params(1, p.box_length, p.dt, p.rc)
        r, v = np.ascontiguousarray(r[:, :1]), np.zeros((3, 1))
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    st = {"rx": r[0].copy(), "ry": r[1].copy(), "rz": r[2].copy(), "ux": r[0].copy(), "uy": r[1].copy(),
          "uz": r[2].copy(), "vx": v[0].copy(), "vy": v[1].copy(), "vz": v[2].copy(), "ax": ax.copy(), "ay": ay.copy(), "az": az.copy()}
    sc_o = oracle.run_steps(po, 5, st)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
        sc = np.stack(eng.verlet_steps(5), axis=1)
    scale = max(np.abs(ax).max(), 1.0)
    assert abs(e - e_o) <= 1e-12 * max(abs(e_o), 1.0) and abs(d - d_o) <= 1e-12 * max(abs(d_o), 1.0)
    assert np.abs(a - np.stack([ax, ay, az])).max() <= 1e-12 * scale
    assert np.max(np.abs(sc - sc_o) / np.maximum(np.abs(sc_o), 1.0)) < 1e-9


def test_force_fcc108_known_answer(golden, oracle):
    g = golden("force_fcc108")
    L = float(g["L"])
    p = init_params(108, L, 0.005, float(g["rc"]), num_cells=3)
    rx, ry, rz = oracle.fcc_lattice(3, L)
    st = init_state(p)
    st.rx[:], st.ry[:], st.rz[:] = rx, ry, rz
    sc = ljmd_amd.compute_lj_potential_energy(p, st)          # stateless drop-in entry point
    check_force(sc, np.stack([st.ax, st.ay, st.az]), g)
    assert np.max(np.abs(np.stack([st.ax, st.ay, st.az]))) < 1e-12


@pytest.mark.parametrize("n", [500, 4096, 20000])
def test_tail_corrections_switch_off_vs_oracle(oracle, n):
    """The reference's compile-time switch use_tail_corrections (lj_potential_energy.f90:36, :205-219): with it off the
    three scalars are the bare pair sums -- ljmd_set_tail_corrections on a handle (gather kernel, one-tile and two-tile
    Newton-3 kernels), ljmd_stateless_set_tail_corrections for the drop-ins the Fortran shim binds; accelerations and ekin
    never contain the constants.  Checked against the oracle's else-branch (the C restatement pinned to the reference
    with the switch on: off, it adds 0.0 instead of the constants, :214-219)."""
    from ljmd_amd import physics
    p, r, v = synthetic.make_config(n, seed=5)
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    on = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    oracle.set_tail_corrections(False)
    try:
        off = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    finally:
        oracle.set_tail_corrections(True)
    te = oracle.tail_corrections(po)
    assert all(abs(t) > 1e-6 * abs(x) for t, x in zip(te, on[:3]))          # (far above the tolerances below: the test can see them)
    tol = 1e-13 if n <= 4096 else 1e-12                                       # (the oracle's own running sums: test_production_kernel_...)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        sc_on = eng.compute_forces()
        a_on = np.stack(eng.get_state(("a",))["a"])
        eng.set_tail_corrections(False)
        sc_off = eng.compute_forces()
        a_off = np.stack(eng.get_state(("a",))["a"])
        steps_off = np.stack(eng.verlet_steps(3))
        eng.set_tail_corrections(True)
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        steps_on = np.stack(eng.verlet_steps(3))
    for k in range(3):
        assert rel(sc_on[k], on[k]) <= tol and rel(sc_off[k], off[k]) <= tol, (k, sc_on[k], sc_off[k], on[k], off[k])
    assert np.array_equal(a_on, a_off)
    assert np.array_equal(steps_on[1], steps_off[1])                          # ekin: the same trajectory
    for row, t in zip((0, 2, 3), te):
        assert np.allclose(steps_on[row] - steps_off[row], t, rtol=1e-9, atol=0.0)
    # the stateless drop-ins (what fortran/lj_potential_energy.f90 and verlet.f90 call)
    lib = physics._lib.load()
    st = init_state(p)
    st.rx[:], st.ry[:], st.rz[:] = r
    st.vx[:], st.vy[:], st.vz[:] = v
    try:
        lib.ljmd_stateless_set_tail_corrections(0)
        sc = ljmd_amd.compute_lj_potential_energy(p, st)
        for k in range(3):
            assert rel(sc[k], off[k]) <= tol, (k, sc[k], off[k])
        e1 = ljmd_amd.verlet_step(p, st)
        lib.ljmd_stateless_set_tail_corrections(1)
        sc = ljmd_amd.compute_lj_potential_energy(p, st)
        e2 = ljmd_amd.verlet_step(p, st)
    finally:
        lib.ljmd_stateless_set_tail_corrections(1)
        physics.stateless_reset()
    assert abs((e1[0] - steps_off[0][0])) <= 1e-9 * abs(e1[0])              # step 1 of the same start, switch off
    assert abs(sc[0] - e1[0] - te[0]) <= 1e-9 * abs(te[0])                   # the same configuration, switch on


def test_force_unwrapped_positions_generic_minimum_image(golden):
    """Positions up to +-3 box lengths outside the box: the library must notice and use the
    exact dnint path (the fast rndne+fma form is only valid for |d/L| < 2.5)."""
    g = golden("force_n500_unwrapped")
    p = init_params(500, float(g["L"]), 0.005, float(g["rc"]))
    r = g["r"]
    assert np.ptp(r[0]) > 2.5 * p.box_length
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], r[0], r[1], r[2])
        sc = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    check_force(sc, a, g)


def _start(eng, g):
    r0, v0 = g["r0"], g["v0"]
    eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
    e, d, dd = eng.compute_forces()
    return e, eng.kinetic_energy(), d, dd


def _series(p, sc):
    etot = sc[:, 0] + sc[:, 1]
    temp = 2.0 * sc[:, 1] / (3.0 * p.n)
    press = (p.n / p.volume) * temp + (-sc[:, 2]) / (3.0 * p.volume)
    return etot, temp, press


@pytest.mark.parametrize("name,nsteps", [("traj_n108", 200), ("traj_n4096_200", 200)])
def test_short_trajectory_vs_reference_golden(golden, name, nsteps):
    """T2: Etot, T, P within 1e-10 relative of the reference's raw fp64 dump for 200 steps."""
    g = golden(name)
    n = int(g["n"])
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    ref = g["scalars"][:nsteps + 1]
    with Engine(p) as eng:
        s0 = _start(eng, g)
        e, k, d, dd = eng.verlet_steps(nsteps)
        mine = np.vstack([s0, np.stack([e, k, d, dd], axis=1)])
        fin = eng.get_state()
    for nm, a, b in zip(("etot", "T", "P"), _series(p, mine), _series(p, ref)):
        err = np.max(np.abs(a - b) / np.abs(b))
        assert err <= REL_TRAJ, (name, nm, err)
    if nsteps == g["scalars"].shape[0] - 1:      # golden final state is at this step
        gf = g["final"]
        mine_f = np.concatenate([np.stack(fin[k]) for k in ("r", "ru", "v", "a")])
        scale = np.array([p.box_length] * 6 + [np.max(np.abs(gf[6:9]))] * 3 + [np.max(np.abs(gf[9:12]))] * 3)
        dr = mine_f[:3] - gf[:3]
        dr -= p.box_length * np.round(dr / p.box_length)     # a particle may sit on the other side of the wrap
        assert np.max(np.abs(dr)) < 1e-8
        assert np.max(np.abs(mine_f[3:] - gf[3:]) / scale[3:, None]) < 1e-7


def test_verlet_step_drop_in_strict_mode(golden, oracle):
    """The stateless verlet_step entry point (what the Fortran shim binds): nine arrays in,
    nine arrays out, one step -- compared with the oracle stepping the same arrays."""
    g = golden("traj_n108")
    n = 108
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    r0, v0 = g["r0"], g["v0"]
    st = init_state(p)
    st.rx[:], st.ry[:], st.rz[:] = r0
    st.vx[:], st.vy[:], st.vz[:] = v0
    e0 = ljmd_amd.compute_lj_potential_energy(p, st)
    o = {k: getattr(st, k).copy() for k in st.FIELDS}
    # identical accelerations in -> positions and half-kicked velocities must agree BIT FOR BIT
    for step in range(5):
        r_before = np.stack([st.rx, st.ry, st.rz]).copy()
        mine = ljmd_amd.verlet_step(p, st)
        ref = oracle.verlet_step(po, o)
        if step == 0:
            assert np.array_equal(np.stack([st.rx, st.ry, st.rz]), np.stack([o["rx"], o["ry"], o["rz"]]))
        for a, b in zip(mine, ref):
            assert rel(a, b) < 1e-12
        assert not np.array_equal(r_before, np.stack([st.rx, st.ry, st.rz]))
    assert rel(e0[0], g["scalars"][0, 0]) < REL_SCALAR


def test_integrator_arithmetic_bit_exact(golden, oracle):
    """Feed the oracle's accelerations to the GPU and take ONE step: r (drift+wrap), ru and the
    first half-kick do not depend on the pair sum, so they must match the oracle bit for bit."""
    g = golden("traj_n4096_200")
    n = 4096
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    r0, v0 = g["r0"], g["v0"]
    _, _, _, ax, ay, az = oracle.compute_forces(po, r0[0].copy(), r0[1].copy(), r0[2].copy())
    st = {"rx": r0[0].copy(), "ry": r0[1].copy(), "rz": r0[2].copy(),
          "ux": r0[0].copy(), "uy": r0[1].copy(), "uz": r0[2].copy(),
          "vx": v0[0].copy(), "vy": v0[1].copy(), "vz": v0[2].copy(), "ax": ax.copy(), "ay": ay.copy(), "az": az.copy()}
    oracle.run_steps(po, 1, st)
    with Engine(p) as eng:
        eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        eng.set_accel(ax, ay, az)
        eng.verlet_steps(1)
        fin = eng.get_state()
    assert np.array_equal(np.stack(fin["r"]), np.stack([st["rx"], st["ry"], st["rz"]]))
    assert np.array_equal(np.stack(fin["ru"]), np.stack([st["ux"], st["uy"], st["uz"]]))
    # v after the step = (v + a*dt/2) + a_new*dt/2 ; a_new differs at the 1e-13 level
    assert np.max(np.abs(np.stack(fin["v"]) - np.stack([st["vx"], st["vy"], st["vz"]]))) < 1e-12


def test_rerun_is_bitwise_deterministic(golden):
    """No atomics, fixed reduction order: two runs give identical bits (T4)."""
    g = golden("traj_n4096_200")
    p = init_params(4096, float(g["L"]), float(g["dt"]), float(g["rc"]))
    outs = []
    for _ in range(2):
        with Engine(p) as eng:
            _start(eng, g)
            sc = np.stack(eng.verlet_steps(50))
            outs.append((sc, np.stack(eng.get_state(("a",))["a"])))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_long_run_n4096_against_oracle_envelope(golden, oracle):
    """BASELINE config 2 (N=4096, 10 000 steps).  Beyond the chaos horizon trajectories decorrelate,
    so the comparison is statistical: the GPU's total-energy wander and <T>, <P> must sit inside
    the envelope of the pinned oracle run from the same start (oracle limited to 300 steps to
    keep the CPU leg short), and momentum must stay conserved."""
    g = golden("traj_n4096_200")
    n = 4096
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    with Engine(p) as eng:
        s0 = _start(eng, g)
        e, k, d, dd = eng.verlet_steps(10000)
        fin = eng.get_state(("v",))
    etot = e + k
    ref = g["scalars"]
    ref_etot = ref[:, 0] + ref[:, 1]
    # the reference itself wanders (truncated, unshifted LJ): compare like with like
    assert abs(etot[:200] - ref_etot[1:201]).max() / abs(ref_etot[0]) < 1e-9
    drift = (etot.max() - etot.min()) / abs(etot.mean())
    assert drift < 5e-3, drift                               # reference N=108: 3.4e-3 peak-to-peak
    assert np.all(np.isfinite(etot))
    vsum = np.abs(np.stack(fin["v"]).sum(axis=1)).max()
    assert vsum < 1e-8, vsum                                 # total momentum stays ~0 (Newton 3)
    temp = 2.0 * k / (3.0 * n)
    assert 0.3 < temp[-2000:].mean() < 3.0


def test_large_n_properties_262144():
    """BASELINE config 3 size: no oracle finishes here, so size-independent properties:
    sum of forces = 0 (Newton 3), translation invariance, permutation invariance, and a
    sampled-row check against a direct numpy evaluation of rows."""
    n = 262144
    p, r, v = synthetic.make_config(n)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e1, d1, dd1 = eng.compute_forces()
        a1 = np.stack(eng.get_state(("a",))["a"])
        # translation by an arbitrary vector + re-wrap
        shift = np.array([0.3, -1.7, 11.1])[:, None]
        r2 = r + shift
        r2 -= p.box_length * np.floor(r2 / p.box_length)
        eng.set_state(r2[0], r2[1], r2[2], v[0], v[1], v[2])
        e2, d2, dd2 = eng.compute_forces()
        a2 = np.stack(eng.get_state(("a",))["a"])
    amax = np.abs(a1).max()
    assert np.abs(a1.sum(axis=1)).max() < 1e-9 * amax * np.sqrt(n)
    assert rel(e2, e1) < 1e-11 and rel(d2, d1) < 1e-10
    assert np.abs(a2 - a1).max() < 1e-9 * amax
    # sampled rows vs numpy (same formulas, fp64)
    rows = np.random.Generator(np.random.PCG64(5)).choice(n, size=8, replace=False)
    L = p.box_length
    for i in rows:
        dvec = r[:, i:i + 1] - r
        dvec -= L * np.round(dvec / L)
        r2_ = (dvec * dvec).sum(axis=0)
        m = (r2_ < p.rc_square) & (np.arange(n) != i)
        u = 1.0 / r2_[m]
        u3 = u * u * u
        gfac = (2.0 * u3 * u3 - u3) * u
        a_np = 24.0 * (gfac * dvec[:, m]).sum(axis=1)
        assert np.abs(a_np - a1[:, i]).max() < 1e-10 * max(np.abs(a_np).max(), 1.0)


@pytest.fixture(scope="module")
def oracle_rows_n262144(oracle):
    """The CPU oracle over ALL 6.9e10 ordered pairs of the bench workload (OpenMP full-matrix form: the reference's
    per-pair arithmetic, rows independent): (params, r, v, accelerations[3, n], (epot, d_epot, dd_epot), threads).
    ~30-40 s of host time on the GPU box's 16-core CPU share; shared by the fp64 and the mixed-precision test."""
    import ctypes
    import os
    n = 262144
    p, r, v = synthetic.make_config(n)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    x, y, z = (np.ascontiguousarray(q) for q in r)
    ao = np.empty((3, n))
    se = sd = sdd = 0.0
    for i0 in range(0, n, 32768):
        ax, ay, az, pe, pd, pdd = oracle.rows_raw(po, i0, i0 + 32768, x, y, z)
        ao[0, i0:i0 + 32768], ao[1, i0:i0 + 32768], ao[2, i0:i0 + 32768] = ax, ay, az
        se, sd, sdd = se + pe, sd + pd, sdd + pdd
    te, td, tdd = oracle.tail_corrections(po)
    ref = (4.0 * (0.5 * se) + te, 24.0 * (0.5 * sd) + td, 24.0 * (0.5 * sdd) + tdd)   # every unordered pair seen twice
    ao *= 24.0
    return p, r, v, ao, ref, cores


def test_bench_configuration_full_parity_vs_oracle_n262144(oracle_rows_n262144):
    """The EXACT bench workload (BASELINE config 3: n = 262144, production kernel configuration), one force
    evaluation, against the CPU oracle over all ordered pairs -- every acceleration and the three scalars."""
    p, r, v, ao, ref, cores = oracle_rows_n262144
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    # `pytest -s` shows the measured deviations (profiles/r0N_full_parity_n262144.txt is this output)
    for name, mine, want in zip(("epot", "d_epot", "dd_epot"), (e, d, dd), ref):
        print(f"{name}: gpu {mine:.16e} oracle {want:.16e} rel.diff {rel(mine, want):.2e}")
    print(f"accelerations: max|a_gpu - a_oracle| / max|a| = {np.abs(a - ao).max() / np.abs(ao).max():.2e}, "
          f"rms rel = {np.sqrt(np.mean((a - ao) ** 2)) / np.sqrt(np.mean(ao ** 2)):.2e} ({cores} oracle threads)")
    for name, mine, want in zip(("epot", "d_epot", "dd_epot"), (e, d, dd), ref):
        assert rel(mine, want) <= 1e-12, (name, mine, want, rel(mine, want))
    assert np.abs(a - ao).max() <= REL_ACCEL * np.abs(ao).max(), np.abs(a - ao).max() / np.abs(ao).max()


def kernel_source_sha16() -> str:
    """the kernel sources a measured summary belongs to (bench.py / tools/pmc_summary.py store the same)"""
    import hashlib
    csrc = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "csrc"
    hsh = hashlib.sha256()
    for name in ("ljmd_kernels.hip", "ljmd_internal.h"):
        hsh.update((csrc / name).read_bytes())
    return hsh.hexdigest()[:16]


@pytest.mark.parametrize("split", ["5", "0"])
def test_mixed_precision_full_parity_vs_oracle_n262144(oracle_rows_n262144, split, monkeypatch):
    """BASELINE config 5 AT ITS OWN SIZE (n = 262144, LJMD_PRECISION_FP32_FORCE), one force evaluation against the
    oracle over all ordered pairs.  Written bounds (the reference has no mixed mode; the anchor is its fp64 loop,
    lj_potential_energy.f90:109-183, through the oracle):
      r_split = 5 sigma (default): every acceleration within 1e-9 max|a| (the large, near forces stay fp64; a far
        pair contributes |f| < 24 * 5^-7 with fp32 relative error), scalars within 5e-9 relative
        (measured 1.8e-10 and 1.0e-9: profiles/r02_mixed_precision_parity_vs_oracle.txt);
      r_split = 0 (every pair outside the own row group in fp32): 2e-5 max|a|, scalars 5e-6
        (measured 6.9e-6 and 9.9e-7)."""
    from ljmd_amd import _lib
    monkeypatch.setenv("LJMD_FP32_SPLIT", split)
    p, r, v, ao, ref, _cores = oracle_rows_n262144
    with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.pair_kernel_name() == "pair_n3_f32_kernel"
        e, d, dd = eng.compute_forces()
        a = np.stack(eng.get_state(("a",))["a"])
    amax = np.abs(ao).max()
    for name, mine, want in zip(("epot", "d_epot", "dd_epot"), (e, d, dd), ref):
        print(f"split {split} {name}: gpu {mine:.16e} oracle {want:.16e} rel.diff {rel(mine, want):.2e}")
    print(f"split {split} accelerations: max|a_gpu - a_oracle| / max|a| = {np.abs(a - ao).max() / amax:.2e}, "
          f"rms rel = {np.sqrt(np.mean((a - ao) ** 2)) / np.sqrt(np.mean(ao ** 2)):.2e}; "
          f"total force / (n max|a|) = {np.abs(a.sum(axis=1)).max() / (p.n * amax):.2e}")
    tol_a, tol_s = (1e-9, 5e-9) if split == "5" else (2e-5, 5e-6)
    if split == "5":
        # the summary bench.py quotes beside the mixed-precision rate (copied to profiles/rNN_mixed_precision_parity_vs_oracle.json)
        import json
        out = ROOT / "gpurun_out"
        out.mkdir(exist_ok=True)
        (out / "mixed_precision_parity_vs_oracle.json").write_text(json.dumps({
            "what": "one force call, n = 262144 bench configuration, LJMD_PRECISION_FP32_FORCE at the default r_split = 5 sigma, "
                    "against the CPU oracle over all 6.9e10 ordered pairs (tests/test_gpu_parity.py)",
            "epot_rel_dev": rel(e, ref[0]), "d_epot_rel_dev": rel(d, ref[1]), "dd_epot_rel_dev": rel(dd, ref[2]),
            "max_abs_accel_dev_over_max_accel": float(np.abs(a - ao).max() / amax),
            "rms_accel_rel_dev": float(np.sqrt(np.mean((a - ao) ** 2)) / np.sqrt(np.mean(ao ** 2))),
            "total_force_over_n_max_accel": float(np.abs(a.sum(axis=1)).max() / (p.n * amax)),
            "test_bounds": {"scalars": tol_s, "accelerations": tol_a},
            # bench.py quotes this file only beside kernels built from the same sources
            "r_split_sigma": float(split), "kernel_source_sha16": kernel_source_sha16()}, indent=1))
    for name, mine, want in zip(("epot", "d_epot", "dd_epot"), (e, d, dd), ref):
        assert rel(mine, want) <= tol_s, (name, mine, want, rel(mine, want))
    assert np.abs(a - ao).max() <= tol_a * amax, np.abs(a - ao).max() / amax


def test_mixed_precision_very_far_passes_keep_the_force_bits(monkeypatch):
    """The cheaper form of the fp32 far loop (pair_n3_f32<., VFAR>) at the bench size: passes whose boxes are farther apart
    than 20.16 sigma (r^-6 < 2^-26) leave u^6 out -- in fp32 2 u^6 - u^3 IS -u^3 there.  Every bit of the accelerations
    stays; the energy sums lose sum u^6 beyond 20.16 sigma (3e-13 of epot, measured 8e-14 / 2.5e-12 / 1.6e-13 on the three
    scalars)."""
    from ljmd_amd import _lib
    n = 262144
    p, r, v = synthetic.make_config(n, seed=5)
    out = {}
    for key in ("0", "1"):
        monkeypatch.setenv("LJMD_FP32_VFAR", key)
        with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            sc = np.array(eng.compute_forces())
            a = np.stack(eng.get_state(("a",))["a"])
        out[key] = (sc, a)
    assert np.array_equal(out["1"][1], out["0"][1])
    dev = np.abs(out["1"][0] - out["0"][0]) / np.abs(out["0"][0])
    print("very far passes: relative change of epot, d_epot, dd_epot", dev)
    assert 0.0 < dev.max() <= 1e-11
    # the far pass on the engine's own stream instead of beside the near pass (LJMD_FP32_FAR_STREAM=0): same launches, same bits
    monkeypatch.setenv("LJMD_FP32_FAR_STREAM", "0")
    with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        sc = np.array(eng.compute_forces())
        a = np.stack(eng.get_state(("a",))["a"])
        ser = np.stack(eng.verlet_steps(3))
    monkeypatch.delenv("LJMD_FP32_FAR_STREAM")
    with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        ser2 = np.stack(eng.verlet_steps(3))
    assert np.array_equal(sc, out["1"][0]) and np.array_equal(a, out["1"][1]) and np.array_equal(ser, ser2)


def test_mixed_precision_energy_series_vs_oracle_n16384(oracle):
    """Mixed-precision trajectory against the ORACLE's own velocity-Verlet series (sequential reference arithmetic),
    n = 16384 (the smallest size the mode accepts), 30 steps from the jittered lattice (~40 s of one host core):
    Etot(t), T(t) within 1e-9 relative at the default r_split, i.e. indistinguishable from fp64 at the level the
    README claims (1e-10 at n = 262144 over 20 steps is asserted by the bench line's own energy check)."""
    from ljmd_amd import _lib
    n, steps = 16384, 30
    p, r, v = synthetic.make_config(n, seed=33)
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    _e, _d, _dd, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    st = {"rx": r[0].copy(), "ry": r[1].copy(), "rz": r[2].copy(), "ux": r[0].copy(), "uy": r[1].copy(),
          "uz": r[2].copy(), "vx": v[0].copy(), "vy": v[1].copy(), "vz": v[2].copy(), "ax": ax, "ay": ay, "az": az}
    sc_o = oracle.run_steps(po, steps, st)                       # [steps, 4]: epot, ekin, d_epot, dd_epot
    out = {}
    for mode in (_lib.PRECISION_FP64, _lib.PRECISION_FP32_FORCE):
        with Engine(p, precision_mode=mode) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            out[mode] = np.stack(eng.verlet_steps(steps), axis=1)
    et_o = sc_o[:, 0] + sc_o[:, 1]
    for mode, tol in ((_lib.PRECISION_FP64, 1e-11), (_lib.PRECISION_FP32_FORCE, 1e-9)):
        sc = out[mode]
        d_et = np.max(np.abs(sc[:, 0] + sc[:, 1] - et_o) / np.abs(et_o))
        d_k = np.max(np.abs(sc[:, 1] - sc_o[:, 1]) / np.abs(sc_o[:, 1]))
        print(f"mode {mode}: max rel dev Etot {d_et:.2e}, Ekin {d_k:.2e} over {steps} steps vs the oracle")
        assert d_et <= tol and d_k <= 10 * tol, (mode, d_et, d_k)


def test_fast_path_equals_generic_path(golden, monkeypatch):
    """The sorted / tile-skipping / rcp+Newton fast path against the exact generic kernel
    (dnint minimum image, IEEE divide, no sorting, no skipping) on the same device."""
    g = golden("traj_n4096_200")
    p = init_params(4096, float(g["L"]), float(g["dt"]), float(g["rc"]))
    res = {}
    for mode in ("fast", "generic", "nosort"):
        if mode == "generic":
            monkeypatch.setenv("LJMD_FORCE_GENERIC", "1")
        elif mode == "nosort":
            monkeypatch.setenv("LJMD_SORT", "0")
        with Engine(p) as eng:
            s0 = _start(eng, g)
            sc = np.stack(eng.verlet_steps(25), axis=1)
            st = eng.get_state()
        monkeypatch.delenv("LJMD_FORCE_GENERIC", raising=False)
        monkeypatch.delenv("LJMD_SORT", raising=False)
        res[mode] = (np.array(s0), sc, np.concatenate([np.stack(st[k]) for k in ("r", "ru", "v", "a")]))
    for mode in ("generic", "nosort"):
        assert np.max(np.abs(res["fast"][0] - res[mode][0]) / np.abs(res[mode][0])) < 1e-13
        assert np.max(np.abs(res["fast"][1] - res[mode][1]) / np.abs(res[mode][1])) < 1e-11
        assert np.max(np.abs(res["fast"][2] - res[mode][2])) < 1e-9


def test_resort_keeps_particle_identity(golden):
    """Re-sorting happens inside verlet_steps; get_state must still return original order, and
    set_accel / set_unwrapped must land on the right particles after a sort."""
    g = golden("traj_n4096_200")
    n = 4096
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    r0, v0 = g["r0"], g["v0"]
    with Engine(p) as eng:
        eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        st = eng.get_state(("r", "v", "ru"))
        assert np.array_equal(np.stack(st["r"]), r0) and np.array_equal(np.stack(st["v"]), v0)
        assert np.array_equal(np.stack(st["ru"]), r0)
        tag = np.arange(n, dtype=np.float64)
        eng.set_accel(tag, tag + 0.25, tag + 0.5)
        eng.set_unwrapped(-tag, tag * 2, tag * 3)
        st = eng.get_state(("a", "ru"))
        assert np.array_equal(st["a"][0], tag) and np.array_equal(st["a"][2], tag + 0.5)
        assert np.array_equal(st["ru"][1], tag * 2)
        eng.set_unwrapped(r0[0], r0[1], r0[2])
        eng.compute_forces()
        eng.verlet_steps(65)                      # crosses three re-sorts (every 20 steps)
        fin = eng.get_state(("r", "ru"))
    # unwrapped - wrapped must be an integer number of box lengths for the SAME particle
    k = (np.stack(fin["ru"]) - np.stack(fin["r"])) / p.box_length
    assert np.max(np.abs(k - np.round(k))) < 1e-9
    assert np.max(np.abs(np.stack(fin["ru"]) - r0)) < 1.5    # nobody moved far in 65 steps (t = 0.33)


@pytest.mark.parametrize("split", ["5", "0"])
def test_mixed_precision_mode_accuracy_and_drift(split, monkeypatch):
    """BASELINE config 5 (fp32 pair arithmetic for far tile pairs, fp64 near pairs + accumulation +
    integrator) against the fp64 engine on the same start: forces to fp32-level accuracy, the
    200-step total-energy series within 1e-7 relative of the fp64 one (the reference's own energy
    wander over such a run is ~1e-4)."""
    from ljmd_amd import _lib
    monkeypatch.setenv("LJMD_FP32_SPLIT", split)
    n = 32768
    p, r, v = synthetic.make_config(n, seed=21)
    out = {}
    for mode in (_lib.PRECISION_FP64, _lib.PRECISION_FP32_FORCE):
        with Engine(p, precision_mode=mode) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            sc0 = eng.compute_forces()
            a0 = np.stack(eng.get_state(("a",))["a"])
            e, k, d, dd = eng.verlet_steps(200)
            out[mode] = (np.array(sc0), a0, e + k, np.stack(eng.get_state(("v",))["v"]))
    f64, mix = out[_lib.PRECISION_FP64], out[_lib.PRECISION_FP32_FORCE]
    # epot, d_epot, dd_epot; d_epot = 24 (S6 - 2 S12) cancels to ~1/10 of its terms, hence the looser all-fp32 bound
    assert np.max(np.abs(mix[0] - f64[0]) / np.abs(f64[0])) < (2e-7 if split == "5" else 5e-6)
    amax = np.abs(f64[1]).max()
    # all-fp32 (split 0): tile-relative fp32 coordinates give ~3e-6 relative error on the close, large forces
    assert np.abs(mix[1] - f64[1]).max() < (2e-7 if split == "5" else 2e-5) * amax
    if split == "5":
        assert np.abs(mix[1] - f64[1]).max() < 1e-8 * amax                    # near pairs (the big forces) stay fp64
    assert np.max(np.abs(mix[2] - f64[2]) / np.abs(f64[2])) < 1e-7            # Etot(t), 200 steps
    # Newton 3 holds per pair, but row and column sides round their fp32 partial sums differently
    assert np.abs(mix[3].sum(axis=1)).max() < (1e-6 if split == "5" else 5e-3)
    with pytest.raises(ljmd_amd.LjmdError):
        Engine(init_params(4096, 17.2, 0.005, 8.0), precision_mode=_lib.PRECISION_FP32_FORCE)   # n too small


@pytest.mark.parametrize("budget_gb", ["64", "24"])
def test_one_million_particles_single_gpu_indexing(budget_gb, monkeypatch):
    """BASELINE config 4 size (N = 1 048 576 = 4 * 64^3, FCC + jitter) on ONE GPU: the column-side slab is
    52 GB here (default budget 64 GB: sized for 288 GB of HBM) and block offsets exceed 2^32 -- an indexing test; with
    a budget of 24 GB the engine chooses 4-wave workgroups that combine in LDS (13 GB).  Properties:
    total force = 0 (Newton 3), sampled rows against a direct numpy evaluation, two steps keep Etot."""
    monkeypatch.setenv("LJMD_SLAB_BUDGET_GB", budget_gb)
    n = 1048576
    p, r, v = synthetic.make_config(n, lattice="fcc")
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0, d0, dd0 = eng.compute_forces()
        k0 = eng.kinetic_energy()
        a = np.stack(eng.get_state(("a",))["a"])
        e, k, d, dd = eng.verlet_steps(2)
    amax = np.abs(a).max()
    assert np.isfinite(e0) and np.abs(a.sum(axis=1)).max() < 1e-9 * amax * np.sqrt(n)
    L = p.box_length
    rows = np.random.Generator(np.random.PCG64(9)).choice(n, size=6, replace=False)
    for i in rows:
        dvec = r[:, i:i + 1] - r
        dvec -= L * np.round(dvec / L)
        r2_ = (dvec * dvec).sum(axis=0)
        m = (r2_ < p.rc_square) & (np.arange(n) != i)
        u = 1.0 / r2_[m]
        u3 = u * u * u
        a_np = 24.0 * (((2.0 * u3 * u3 - u3) * u) * dvec[:, m]).sum(axis=1)
        assert np.abs(a_np - a[:, i]).max() < 1e-10 * max(np.abs(a_np).max(), 1.0)
    etot = e + k
    assert abs(etot[-1] - (e0 + k0)) < 1e-5 * abs(e0 + k0)


def test_argument_guards_and_sequence_errors():
    with pytest.raises(ljmd_amd.LjmdError) as ei:
        Engine(ljmd_amd.SimParams(n=10, box_length=10.0, dt=0.005, rc=5.0))      # rc >= L/2
    assert ei.value.code == -1 and "L/2" in ei.value.message
    p = init_params(64, 5.0, 0.005, 2.0)
    with Engine(p) as eng:
        with pytest.raises(ljmd_amd.LjmdError) as e2:
            eng.compute_forces()                                                    # no state yet
        assert e2.value.code == -4
        z = np.linspace(0.1, 4.9, 64)
        eng.set_state(z, z[::-1].copy(), z, z, z, z)
        with pytest.raises(ljmd_amd.LjmdError):
            eng.verlet_steps(1)                                                     # no accelerations yet
        eng.compute_forces()
        assert eng.verlet_steps(0)[0].shape == (0,)
    with pytest.raises(ValueError):
        ljmd_amd.compute_lj_potential_energy(p, ljmd_amd.SimState())               # not allocated


def test_production_loop_files_vs_reference(tmp_path, golden):
    """Config 1 end to end (input file -> rv_init.dat -> energies + rva.dat) against the files
    the reference's own programs wrote.  Text file: 7 significant digits; rva.dat: raw fp64."""
    import shutil
    from ljmd_amd import io_formats, simulation
    src = GOLDEN / "ref_run_n108_oi100"
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs" / "one_run").mkdir(parents=True)
    shutil.copy(src / "input_simulation_parameters.txt", tmp_path / "inputs")
    shutil.copy(src / "rv_init.dat", tmp_path / "outputs")
    res = simulation.run_md_simulation(tmp_path)
    assert res.n_samples == 9
    mine = io_formats.read_energies(tmp_path / "outputs" / "one_run" / "instantaneous_energies.dat")
    ref = io_formats.read_energies(src / "instantaneous_energies.dat")
    assert mine.shape == ref.shape == (9, 6)
    # steps <= 1000 at N=108: P stays within ~1e-7 of the reference up to step 1000
    assert np.allclose(mine, ref, rtol=5e-6, atol=0)
    h1, s1 = io_formats.read_rva(tmp_path / "outputs" / "one_run" / "rva.dat")
    h2, s2 = io_formats.read_rva(src / "rva.dat")
    assert h1 == h2 and s1.shape == s2.shape == (9, 4, 3, 108)
    assert (tmp_path / "outputs" / "one_run" / "rva.dat").stat().st_size == (src / "rva.dat").stat().st_size
    assert np.abs(s1[0] - s2[0]).max() < 1e-9          # first snapshot (step 200): inside the horizon


def test_async_steps_and_snapshot_equal_the_synchronous_path(golden):
    """ljmd_enqueue_steps/collect_steps + ljmd_snapshot_begin/end (the pipelined production loop) against
    ljmd_verlet_steps + ljmd_get_state: same kernels, same order -> bitwise equal scalars and arrays, even
    though 45 further steps (two re-sorts) are enqueued between snapshot_begin and snapshot_end."""
    g = golden("traj_n4096_200")
    p = init_params(4096, float(g["L"]), float(g["dt"]), float(g["rc"]))
    r0, v0 = g["r0"], g["v0"]
    with Engine(p) as ref:
        ref.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        ref.compute_forces()
        sc_a = ref.verlet_steps(30)
        st_a = ref.get_state()
        sc_b = ref.verlet_steps(45)
        st_b = ref.get_state()
    with Engine(p) as eng:
        eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        eng.compute_forces()
        eng.enqueue_steps(30)
        got_a = eng.collect_steps(30)
        eng.snapshot_begin()
        eng.enqueue_steps(45)                       # runs while the snapshot is in flight
        snap = eng.snapshot_end()
        got_b = eng.collect_steps(45)
        fin = eng.get_state()
    for x, y in zip(sc_a + sc_b, got_a + got_b):
        assert np.array_equal(x, y)
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(snap[key]), np.stack(st_a[key])), key
        assert np.array_equal(np.stack(fin[key]), np.stack(st_b[key])), key


@pytest.mark.parametrize("n,mode,devices,steps", [
    (4096, 0, None, 40), (20000, 0, None, 25), (65536, 0, None, 12), (262144, 0, None, 6), (262144, 1, None, 6),
    (65536, 0, [0, 0, 0, 0], 12)])
def test_sampled_segments_skip_the_energy_sums_and_nothing_else(n, mode, devices, steps):
    """ljmd_enqueue_steps_sampled (potential-energy sums on the last step of the segment only -- where the reference
    samples, md_simulation_program.f90:361) and ljmd_verlet_steps with NULL outputs against the every-step path:
    r, ru, v, a, ekin of every step and the sampled step's epot / d_epot / dd_epot are BITWISE equal; the steps in
    between report NaN (never a stale number) wherever the forces-only kernel exists (Newton-3 kernels, one wave per
    workgroup)."""
    p, r, v = synthetic.make_config(n, seed=11)
    kw = dict(precision_mode=mode) if devices is None else dict(precision_mode=mode, devices=devices)
    with Engine(p, **kw) as ref:
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        ref.compute_forces()
        want = np.stack(ref.verlet_steps(steps))                 # rows epot, ekin, d_epot, dd_epot
        want2 = np.stack(ref.verlet_steps(steps))
        st = ref.get_state()
        kernel = ref.pair_kernel_name()
    with Engine(p, **kw) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.enqueue_steps(steps, sampled=True)
        got = np.stack(eng.collect_steps(steps))
        eng.advance(steps - 1)                                   # nobody reads these steps' sums
        got2 = np.stack(eng.verlet_steps(1))
        fin = eng.get_state()
        sc = eng.compute_forces()                                # always evaluates the sums
    with Engine(p, **kw) as chk:
        chk.set_state(*fin["r"], *fin["v"])
        # (a fresh engine sorts the particles into another slot order: same sums, other rounding)
        # (fp64: the virial is a difference of two large sums; mixed: the far sums are accumulated in fp32 per tile pass)
        assert np.allclose(np.array(chk.compute_forces()), np.array(sc), rtol=1e-11 if mode == 0 else 1e-7, atol=0.0)
    assert np.array_equal(got[1], want[1])                       # ekin, every step
    assert np.array_equal(got[:, -1], want[:, -1])               # the sampled step, all four
    assert np.array_equal(got2[:, 0], want2[:, -1])
    between = got[[0, 2, 3], :-1]
    assert np.all(np.isnan(between) | (between == want[[0, 2, 3], :-1]))
    if "n3" in kernel and os.environ.get("LJMD_N3_WG_WAVES", "1") == "1":
        assert np.all(np.isnan(between)), kernel
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(fin[key]), np.stack(st[key])), key


def test_async_api_sequence_errors():
    p = init_params(256, 8.0, 0.005, 3.0)
    z = np.linspace(0.1, 7.9, 256)
    with Engine(p) as eng:
        with pytest.raises(ljmd_amd.LjmdError) as e:
            eng.snapshot_begin()                                   # no state yet
        assert e.value.code == -4
        eng.set_state(z, z[::-1].copy(), np.roll(z, 5), z * 0, z * 0, z * 0)
        with pytest.raises(ljmd_amd.LjmdError):
            eng.enqueue_steps(1)                                   # no accelerations yet
        eng.compute_forces()
        with pytest.raises(ljmd_amd.LjmdError) as e:
            eng.snapshot_end()                                     # nothing in flight
        assert e.value.code == -4
        eng.snapshot_begin()
        with pytest.raises(ljmd_amd.LjmdError) as e:
            eng.snapshot_begin()                                   # one snapshot at a time
        assert e.value.code == -4
        st = eng.snapshot_end()
        assert np.array_equal(st["r"][0], z)
        eng.enqueue_steps(4096)                                    # LJMD_MAX_PENDING_STEPS
        with pytest.raises(ljmd_amd.LjmdError) as e:
            eng.enqueue_steps(1)                                   # record ring would overflow
        assert e.value.code == -4
        with pytest.raises(ljmd_amd.LjmdError):
            eng.collect_steps(4097)
        assert eng.collect_steps(3)[0].shape == (3,)               # the last 3 of the 4096; older ones dropped
        eng.enqueue_steps(2)
        assert eng.collect_steps(2)[1].shape == (2,)


def test_resume_from_an_rva_snapshot(golden, tmp_path):
    """Checkpoint / resume (the reference cannot resume, but its rva.dat records hold r, ru, v, a):
    a run continued from a snapshot written to and read back from an rva.dat file follows the uninterrupted run.
    Not bitwise -- the resumed engine re-sorts its particles on a different cadence, which changes summation
    orders -- but to rounding level over the 40 steps compared."""
    from ljmd_amd import io_formats
    g = golden("traj_n4096_200")
    n = 4096
    p = init_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    r0, v0 = g["r0"], g["v0"]
    with Engine(p) as eng:
        eng.set_state(r0[0], r0[1], r0[2], v0[0], v0[1], v0[2])
        eng.compute_forces()
        eng.verlet_steps(30)
        snap = eng.get_state()
        with io_formats.RvaWriter(tmp_path / "rva.dat", n, p.box_length, p.dt, 30, 1) as w:
            w.write_snapshot(snap["r"], snap["ru"], snap["v"], snap["a"])
        sc_a = eng.verlet_steps(40)
        fin_a = eng.get_state()
    hdr, snaps = io_formats.read_rva(tmp_path / "rva.dat")
    assert hdr["n"] == n and snaps.shape == (1, 4, 3, n)
    r, ru, v, a = snaps[0]
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.set_unwrapped(ru[0], ru[1], ru[2])
        eng.set_accel(a[0], a[1], a[2])
        sc_b = eng.verlet_steps(40)                    # no force call needed: a(t) came from the snapshot
        fin_b = eng.get_state()
    for x, y in zip(sc_a, sc_b):
        assert np.max(np.abs(x - y) / np.abs(x)) < 1e-11
    for key in ("r", "ru", "v"):
        assert np.abs(np.stack(fin_a[key]) - np.stack(fin_b[key])).max() < 1e-10, key


def test_config2_10000_steps_vs_the_references_own_series():
    """BASELINE config 2 (N = 4096, 10 000 steps) against the raw per-step scalars of the REAL reference run to
    the same length from the same start (tests/golden/traj_n4096_10000.npz, 41 min of one core, produced by
    oracle/ref_harness over the reference's modules).  Inside the chaos horizon: step by step, 1e-10.  Beyond it
    the two trajectories are different members of the same NVE ensemble: equal conserved energy (block means),
    equal <T> and <P> within their statistical error."""
    from ljmd_amd.physics import observables
    g = np.load(GOLDEN / "traj_n4096_10000.npz")
    ref = g["scalars"]                                          # [10001, 4] incl. t = 0
    n = 4096
    p, r, v = synthetic.make_config(n)
    assert float(g["L"]) == p.box_length and float(g["rc"]) == p.rc
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0, d0, dd0 = eng.compute_forces()
        k0 = eng.kinetic_energy()
        e, k, d, dd = eng.verlet_steps(10000)
    mine = np.vstack([[e0, k0, d0, dd0], np.stack([e, k, d, dd], axis=1)])
    et, er = mine[:, 0] + mine[:, 1], ref[:, 0] + ref[:, 1]
    rel = np.abs(et - er) / np.abs(er)
    assert rel[:201].max() <= REL_TRAJ                           # 200 steps: 1e-10
    horizon = int(np.argmax(rel > REL_TRAJ))
    assert horizon > 300, horizon                                # the reference leaves ITSELF (FMA on/off) at ~950 for N = 108
    # conserved quantity: means over the second half agree far inside the instantaneous fluctuation
    fluct = er[5000:].std() / abs(er.mean())
    assert abs(et[5000:].mean() - er[5000:].mean()) / abs(er.mean()) < 0.5 * fluct
    assert abs(et[5000:].std() / er[5000:].std() - 1.0) < 0.25   # same fluctuation level
    Tm, Tr = 2.0 * mine[5000:, 1].mean() / (3.0 * n), 2.0 * ref[5000:, 1].mean() / (3.0 * n)
    assert abs(Tm / Tr - 1.0) < 5e-3
    Pm = np.mean([observables(p, a, b, c)[2] for a, b, c in mine[5000:, :3]])
    Pr = np.mean([observables(p, a, b, c)[2] for a, b, c in ref[5000:, :3]])
    assert abs(Pm / Pr - 1.0) < 2e-2


@pytest.mark.parametrize("n,knobs", [(108, {}), (500, {}), (3000, {}), (4096, {}), (5000, {}), (8192, {}), (16384, {}), (20000, {}),
                                     (32768, {}),
                                     (4096, {"LJMD_N3_ROW_TILES": "2"}), (5000, {"LJMD_N3_ROW_TILES": "2"}),
                                     (8192, {"LJMD_N3_ROW_TILES": "1"}),
                                     (2000, {"LJMD_N3_ROW_TILES": "2", "LJMD_N3_MIN_N": "1"})])
def test_fused_launches_are_bitwise_equal_to_the_separate_kernels(n, knobs, monkeypatch):
    """LJMD_FUSE=1 (default): tile boxes written by the drift kernel, finalize folded into the kick kernel through
    a last-block ticket -- and, for single-rank systems of up to 20 000 particles (LJMD_FUSE_TAIL), a step in TWO launches:
    the pair kernel working its pass descriptors out itself, and tile_tail_kernel (slab reduction + kick + step record +
    the next step's drift / wrap / half-kick / boxes, one block per tile); below 4096 particles the gather kernel works
    its mask words out itself; and inside a batch the step record is folded by the NEXT tail launch
    (LJMD_FUSE_DEFER_RECORD).  Same values, same reductions, same order -> the same bits as the separate launches,
    sampled (forces-only) segments included."""
    for k, val in knobs.items():               # two-tile row groups (one tail block per group) where the library takes one, and back
        monkeypatch.setenv(k, val)
    p, r, v = synthetic.make_config(n, seed=9)
    out = []
    for fuse, step, defer in (("1", "1", "1"), ("1", "1", "0"), ("1", "0", "1"), ("0", "0", "1")):
        monkeypatch.setenv("LJMD_FUSE", fuse)
        monkeypatch.setenv("LJMD_FUSE_TAIL", step)
        monkeypatch.setenv("LJMD_FUSE_DEFER_RECORD", defer)
        with Engine(p) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            e0 = eng.compute_forces()
            sc = np.stack(eng.verlet_steps(45))          # crosses re-sorts for n >= 1024
            e1 = eng.compute_forces()                    # a force call right after steps: boxes must be recomputed
            eng.enqueue_steps(7, sampled=True)
            sc2 = np.stack(eng.collect_steps(7))
            st = eng.get_state()
            out.append((e0, sc, e1, np.stack([np.stack(st[k]) for k in ("r", "ru", "v", "a")]), sc2))
    for other in out[1:]:
        assert out[0][0] == other[0] and out[0][2] == other[2]
        assert np.array_equal(out[0][1], other[1]) and np.array_equal(out[0][3], other[3])
        assert np.array_equal(out[0][4], other[4], equal_nan=True)


def test_failed_batch_poisons_the_handle_until_set_state(monkeypatch):
    """A failure in the middle of a batch of steps (here injected: the force phase of the 4th evaluation fails
    behind an already enqueued drift) leaves no trajectory point on the device: every stepping entry point then
    returns LJMD_ERR_STATE until ljmd_set_state, after which the handle behaves like a fresh one (bitwise)."""
    p, r, v = synthetic.make_config(4096, seed=12)
    with Engine(p) as ref:
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e_ref = ref.compute_forces()
        sc_ref = np.stack(ref.verlet_steps(30), axis=1)
    monkeypatch.setenv("LJMD_INJECT_FAILURE_AT_STEP", "3")     # evaluations: t = 0 force call, steps 1, 2, [3]
    with Engine(p) as eng:
        monkeypatch.delenv("LJMD_INJECT_FAILURE_AT_STEP")
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.compute_forces() == e_ref
        with pytest.raises(ljmd_amd.LjmdError, match="injected failure"):
            eng.verlet_steps(10)
        for call in (lambda: eng.verlet_steps(1), lambda: eng.enqueue_steps(1), lambda: eng.collect_steps(1),
                     eng.compute_forces):
            with pytest.raises(ljmd_amd.LjmdError, match="poisoned"):
                call()
        eng.get_state()                                        # reading the (meaningless) state is still allowed
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])      # recovery
        assert eng.compute_forces() == e_ref
        assert np.array_equal(np.stack(eng.verlet_steps(30), axis=1), sc_ref)


def test_stateless_verlet_step_resident_fast_path(monkeypatch):
    """ljmd_verlet_step called in a loop the way the reference's own program does (md_simulation_program.f90:
    303-353 only READS the arrays between steps): from the second call on the arrays are the ones the library handed
    back, so it steps the resident state -- no upload, no re-sort.  Same trajectory as the strict path
    (LJMD_STATELESS_FASTPATH=0) within rounding (the slot order, hence the summation order, differs), and any
    change of any array by the caller falls back to the strict path."""
    import time
    from ljmd_amd import physics
    n, steps = 32768, 12
    p, r, v = synthetic.make_config(n, seed=4)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LJMD_STATELESS_FASTPATH", mode)
        physics.stateless_reset()
        st = init_state(p)
        st.rx[:], st.ry[:], st.rz[:] = r
        st.vx[:], st.vy[:], st.vz[:] = v
        physics.compute_lj_potential_energy(p, st)
        sc = []
        t0 = time.perf_counter()
        for k in range(steps):
            if k == 7:
                st.vx[5] += 0.0                                # same bytes: still the fast path
            if k == 9:
                st.vx[5] = np.nextafter(st.vx[5], 1.0)         # the caller touched the state: strict path, exact effect
            sc.append(physics.verlet_step(p, st))
        res[mode] = (np.array(sc), np.stack([st.rx, st.vx, st.ax]).copy(), time.perf_counter() - t0)
    physics.stateless_reset()
    fast, strict = res["1"], res["0"]
    assert np.max(np.abs(fast[0] - strict[0]) / np.abs(strict[0])) < 1e-11
    assert np.abs(fast[1] - strict[1]).max() < 1e-9
    print(f"stateless verlet_step loop, n = {n}: resident fast path {fast[2] / steps * 1e3:.2f} ms/call, "
          f"strict {strict[2] / steps * 1e3:.2f} ms/call")
    assert fast[2] < strict[2]


@pytest.mark.parametrize("n", [20000, 16384])
@pytest.mark.parametrize("wg", ["2", "4"])
def test_lds_combining_workgroups_equal_one_wave_per_workgroup(wg, n, monkeypatch, oracle):
    """pair_n3_kernel<., 4, W>: W consecutive row groups per workgroup, column-side partial accelerations combined
    in LDS (one slab block per workgroup and column tile).  Same pairs, different summation tree on the column side:
    against the one-wave form to rounding, against the oracle within the usual bounds, and a 25-step trajectory
    (crosses a re-sort); run-to-run bitwise.  n = 20000 leaves a partially filled last tile and row group, and its 79 row
    groups are no multiple of W: the column-side blocks are numbered by workgroup (N3Args::slab_j); n = 16384 (64 row
    groups) numbers them by offset / W."""
    monkeypatch.setenv("LJMD_N3_ROW_TILES", "4")
    p, r, v = synthetic.make_config(n, seed=17)
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    ao = np.stack([ax, ay, az])
    out = {}
    for w in ("1", wg, wg):
        monkeypatch.setenv("LJMD_N3_WG_WAVES", w)
        with Engine(p) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            sc0 = eng.compute_forces()
            a0 = np.stack(eng.get_state(("a",))["a"])
            sc = np.stack(eng.verlet_steps(25), axis=1)
        out.setdefault(w, []).append((np.array(sc0), a0, sc))
    one, (wa, wb) = out["1"][0], out[wg]
    assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1]) and np.array_equal(wa[2], wb[2])   # deterministic
    assert np.max(np.abs(wa[0] - one[0]) / np.abs(one[0])) < 1e-13
    assert np.abs(wa[1] - one[1]).max() < 1e-12 * np.abs(one[1]).max()
    assert np.max(np.abs(wa[2] - one[2]) / np.abs(one[2])) < 1e-11
    for mine, ref in zip(wa[0], (e_o, d_o, dd_o)):
        assert rel(mine, ref) <= 1e-12            # 2e8 terms per sum on both sides (as in the full-size parity test)
    assert np.abs(wa[1] - ao).max() <= REL_ACCEL * np.abs(ao).max()


@pytest.mark.parametrize("n,row_tiles,target,both", [
    (20000, "4", "1000000", "1"), (20000, "4", "20000", "1"), (20000, "2", "1000000", "1"),
    (16384, "4", "1000000", "1"), (16384, "4", "1000000", "0"), (16384, "2", "30000", "1"),
    (12288, "1", "1000000", "1"), (12288, "1", "1000000", "0"), (5000, "2", "1000000", "1"), (5000, "1", "1000000", "1")])
def test_single_pass_work_items_equal_slices_of_offsets(n, row_tiles, target, both, monkeypatch, oracle):
    """Work items finer than whole offsets (N3Args::uchunk): down to ONE pass per work item, the tie d = NG / 2 worked from
    both sides (both_ties: the steps 0 .. 31 by the lower row group, 1 .. 32 by the upper one) or by its lower row group.
    Same pairs, another summation tree: against one work item per row group to rounding, against the oracle within the
    usual bounds, a 25-step trajectory (crosses a re-sort), run-to-run bitwise.  n = 20000 and 5000 leave a partially filled
    last tile and row group; 16384 / 4 tiles and 12288 / 1 tile have an even number of row groups (a tie)."""
    monkeypatch.setenv("LJMD_N3_ROW_TILES", row_tiles)
    monkeypatch.setenv("LJMD_FUSE_TAIL", "0")
    p, r, v = synthetic.make_config(n, seed=19)
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    ao = np.stack([ax, ay, az])
    out = {}
    for key, knobs in (("whole", {"LJMD_N3_TARGET_WAVES": "1", "LJMD_N3_BOTH_TIES": "0"}),
                       ("cut", {"LJMD_N3_TARGET_WAVES": target, "LJMD_N3_BOTH_TIES": both}),
                       ("cut2", {"LJMD_N3_TARGET_WAVES": target, "LJMD_N3_BOTH_TIES": both})):
        for k, val in knobs.items():
            monkeypatch.setenv(k, val)
        with Engine(p) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            sc0 = eng.compute_forces()
            a0 = np.stack(eng.get_state(("a",))["a"])
            sc = np.stack(eng.verlet_steps(25), axis=1)
        out[key] = (np.array(sc0), a0, sc)
    one, wa, wb = out["whole"], out["cut"], out["cut2"]
    assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1]) and np.array_equal(wa[2], wb[2])   # deterministic
    assert np.max(np.abs(wa[0] - one[0]) / np.abs(one[0])) < 1e-13
    assert np.abs(wa[1] - one[1]).max() < 1e-12 * np.abs(one[1]).max()
    assert np.max(np.abs(wa[2] - one[2]) / np.abs(one[2])) < 1e-11
    for mine, ref in zip(wa[0], (e_o, d_o, dd_o)):
        assert rel(mine, ref) <= 1e-12
    assert np.abs(wa[1] - ao).max() <= REL_ACCEL * np.abs(ao).max()


@pytest.mark.parametrize("wg", ["2", "4"])
def test_mixed_precision_far_pass_beside_lds_combining_workgroups(wg, monkeypatch, oracle):
    """The fp32 far pass is always one wave per workgroup and numbers its column-side blocks by offset on one rank
    (CS2 = Dmax + 1 blocks per column tile), whatever the fp64 near kernel's workgroups do: with W = 2 / 4 and 79 row
    groups (no multiple of W) the near kernel numbers ITS blocks by workgroup, and the far pass must not inherit that
    (blocks of neighbouring column tiles would collide and the last tiles write past the slab).  Against the one-wave
    configuration and the oracle at the mixed mode's bounds; run-to-run bitwise."""
    from ljmd_amd import _lib
    n = 20000
    p, r, v = synthetic.make_config(n, seed=17)
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    e_o, d_o, dd_o, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    ao = np.stack([ax, ay, az])
    out = {}
    for w in ("1", wg, wg):
        monkeypatch.setenv("LJMD_N3_WG_WAVES", w)
        with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            assert eng.pair_kernel_name() == "pair_n3_f32_kernel"
            sc0 = np.array(eng.compute_forces())
            a0 = np.stack(eng.get_state(("a",))["a"])
            sc = np.stack(eng.verlet_steps(12), axis=1)            # crosses a re-sort
        out.setdefault(w, []).append((sc0, a0, sc))
    one, (wa, wb) = out["1"][0], out[wg]
    assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1]) and np.array_equal(wa[2], wb[2])
    assert np.isfinite(wa[1]).all()
    # the far pass is the same code in both: what differs is the near kernel's column-side summation tree
    assert np.max(np.abs(wa[0] - one[0]) / np.abs(one[0])) < 1e-12
    assert np.abs(wa[1] - one[1]).max() < 1e-11 * np.abs(one[1]).max()
    assert np.max(np.abs(wa[2] - one[2]) / np.abs(one[2])) < 1e-9
    for mine, ref in zip(wa[0], (e_o, d_o, dd_o)):
        assert rel(mine, ref) <= 5e-9
    # the reference's potential is truncated, not shifted (lj_potential_energy.f90:132): a pair within fp32 rounding of rc
    # (a relative shell of 1e-7: ~30 of the 1e8 pairs inside the cutoff here) may fall on the other side of the fp32 test
    # and then adds or drops the whole edge force 24 (rc^-7 - 2 rc^-13) = 1.9e-7 at rc = 14.3 -- at the bench size
    # (rc = 33.8) that is 5e-10 and disappears in the bound; here it is the bound
    edge = 24.0 * abs(p.rc ** -7 - 2.0 * p.rc ** -13)
    assert np.abs(wa[1] - ao).max() <= 1e-9 * np.abs(ao).max() + 2.0 * edge


def test_full_size_invariances_permutation_translation_reflection():
    """Size-independent properties of the force routine at the bench size (n = 262144), where no CPU run of the
    reference exists beyond the single oracle comparison above: the result must not depend on the ORDER of the
    particles (other tiles, other work items, other summation trees: agreement to rounding), on a rigid TRANSLATION of
    the periodic system (other images, other tile boxes), or on a REFLECTION x -> L - x (forces change sign in x)."""
    n = 262144
    p, r, v = synthetic.make_config(n, seed=41)
    L = p.box_length
    rng = np.random.Generator(np.random.PCG64(99))
    perm = rng.permutation(n)
    shift = np.array([0.37 * L, -1.21 * L, 0.5 * L])
    r_shift = np.mod(r + shift[:, None], L)
    r_refl = r.copy()
    r_refl[0] = np.mod(L - r[0], L)
    res = []
    with Engine(p) as eng:
        for pos in (r, np.ascontiguousarray(r[:, perm]), r_shift, r_refl):
            eng.set_state(pos[0], pos[1], pos[2], v[0], v[1], v[2])
            sc = np.array(eng.compute_forces())
            res.append((sc, np.stack(eng.get_state(("a",))["a"])))
    (s0, a0), (s1, a1), (s2, a2), (s3, a3) = res
    amax = np.abs(a0).max()
    for name, s in (("permutation", s1), ("translation", s2), ("reflection", s3)):
        assert np.max(np.abs(s - s0) / np.abs(s0)) <= 1e-12, (name, s, s0)
    assert np.abs(a1 - a0[:, perm]).max() <= 1e-12 * amax
    # a translated / reflected coordinate differs from the original in its last bits (mod, L - x): the pair distances
    # change by ~1e-14 relative, the r^-13 forces by ~1e-13 per close pair
    assert np.abs(a2 - a0).max() <= 1e-10 * amax
    a3[0] *= -1.0
    assert np.abs(a3 - a0).max() <= 1e-10 * amax
    assert np.abs(a0.sum(axis=1)).max() <= 1e-9 * amax                      # Newton's third law, all pairs
