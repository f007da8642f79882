"""Pins the C restatement (oracle/ljmd_oracle.c) to the REAL reference: every golden vector in
tests/golden/ was produced by the reference compiled with amdflang (oracle/make_golden.py).
Bit-exact unless stated."""
import json

import numpy as np

from conftest import GOLDEN

KEYS = ("rx", "ry", "rz", "ux", "uy", "uz", "vx", "vy", "vz", "ax", "ay", "az")


def test_kat_minimum_image_wrap_ran3(oracle):
    k = json.loads((GOLDEN / "kat.json").read_text())
    L, invL = 10.0, 0.1
    assert [oracle.minimum_image(x, L, invL) for x in (9.5, 5.0, -5.0)] == k["mic"] == [-0.5, -5.0, 5.0]
    assert [oracle.minimum_image(x, L, invL) for x in (4.999999, -14.9, 25.0)] == k["mic2"]
    x = np.array([-0.4, 10.2, 20.7]); y = np.array([9.8, 0.0, 10.0]); z = np.array([-10.0, -1e-20, 9.999999999999999])
    oracle.wrap_positions(x, y, z, L)
    assert x.tolist() == k["wrapx"] and y.tolist() == k["wrapy"] and z.tolist() == k["wrapz"]
    assert z[1] == L  # the wrap CAN return exactly L (x = -1e-20): positions live in [0, L], not [0, L)
    assert oracle.ran3_sequence(-12345, 8).tolist() == k["ran3_seed_-12345"]


def test_force_fcc108_known_answer(oracle, golden):
    g = golden("force_fcc108")
    L = float(g["L"])
    p = oracle.derive_params(108, L, 0.005, float(g["rc"]))
    rx, ry, rz = oracle.fcc_lattice(3, L)
    e, d, dd, ax, ay, az = oracle.compute_forces(p, rx, ry, rz)
    assert [e, d, dd] == g["scalars"].tolist()
    # the survey's published known answer
    assert e == -7.3290568709265926e+02 and d == 2.7873485322985493e+03 and dd == -1.9041264300238413e+02
    assert np.array_equal(np.stack([ax, ay, az]), g["a"])


def test_force_cases_bit_exact(oracle, golden):
    for name in ("force_n108", "force_n500", "force_n4000", "force_n4096", "force_n500_unwrapped"):
        g = golden(name)
        p = oracle.derive_params(int(g["n"]), float(g["L"]), 0.005, float(g["rc"]))
        r = np.ascontiguousarray(g["r"])
        e, d, dd, ax, ay, az = oracle.compute_forces(p, r[0].copy(), r[1].copy(), r[2].copy())
        assert [e, d, dd] == g["scalars"].tolist(), name
        assert np.array_equal(np.stack([ax, ay, az]), g["a"]), name


def test_tail_constants(oracle, golden):
    # perfect FCC has no pair inside... no: it has; instead isolate the tail by differencing two cutoffs
    g = golden("force_fcc108")
    p = oracle.derive_params(108, float(g["L"]), 0.005, float(g["rc"]))
    te, td, tdd = oracle.tail_corrections(p)
    assert te < 0 and td > 0 and tdd < 0
    assert abs(te) < 50 and abs(td) < 500


def test_tail_switch_off_branch(oracle, golden):
    """lj_potential_energy.f90:36,214-219: with use_tail_corrections off the reference adds 0.0 instead of the three
    constants.  The oracle's switch (ora_set_tail_corrections) leaves the pair sums, forces included, bit for bit alone."""
    g = golden("force_n500")
    p = oracle.derive_params(500, float(g["L"]), 0.005, float(g["rc"]))
    r = g["r"]
    on = oracle.compute_forces(p, r[0].copy(), r[1].copy(), r[2].copy())
    oracle.set_tail_corrections(False)
    try:
        off = oracle.compute_forces(p, r[0].copy(), r[1].copy(), r[2].copy())
    finally:
        oracle.set_tail_corrections(True)
    again = oracle.compute_forces(p, r[0].copy(), r[1].copy(), r[2].copy())
    te = oracle.tail_corrections(p)
    for k in range(3):
        assert on[k] == again[k] and off[k] != on[k]
        assert abs((on[k] - off[k]) - te[k]) <= 4e-16 * max(abs(on[k]), abs(te[k]))     # one rounding of the final addition
    for k in (3, 4, 5):
        assert np.array_equal(on[k], off[k])


def _run_traj(oracle, g, nsteps):
    n = int(g["n"])
    p = oracle.derive_params(n, float(g["L"]), float(g["dt"]), float(g["rc"]))
    r0, v0 = g["r0"], g["v0"]
    st = {"rx": r0[0].copy(), "ry": r0[1].copy(), "rz": r0[2].copy(),
          "ux": r0[0].copy(), "uy": r0[1].copy(), "uz": r0[2].copy(),
          "vx": v0[0].copy(), "vy": v0[1].copy(), "vz": v0[2].copy()}
    e, d, dd, ax, ay, az = oracle.compute_forces(p, st["rx"], st["ry"], st["rz"])
    st.update(ax=ax, ay=ay, az=az)
    k0 = oracle.ekin_fused(st["vx"], st["vy"], st["vz"])
    sc = oracle.run_steps(p, nsteps, st)
    return np.vstack([[e, k0, d, dd], sc]), st


def test_traj_n108_10000_steps_bit_exact(oracle, golden):
    """T3: sequential i<j, no FMA => identical trajectory for 10 000 steps."""
    g = golden("traj_n108")
    sc, st = _run_traj(oracle, g, 10000)
    ref = g["scalars"]
    for col, name in ((0, "epot"), (2, "d_epot"), (3, "dd_epot")):
        assert np.array_equal(sc[:, col], ref[:, col]), name
    # ekin: the reference's `sum` intrinsic order is compiler-defined (vectorised partial sums)
    assert np.max(np.abs(sc[:, 1] - ref[:, 1]) / ref[:, 1]) < 2e-15
    assert np.array_equal(np.stack([st[k] for k in KEYS]), g["final"])


def test_traj_n4096_first_steps_bit_exact(oracle, golden):
    g = golden("traj_n4096_200")
    nsteps = 30
    sc, _ = _run_traj(oracle, g, nsteps)
    ref = g["scalars"][:nsteps + 1]
    for col in (0, 2, 3):
        assert np.array_equal(sc[:, col], ref[:, col])
    assert np.max(np.abs(sc[:, 1] - ref[:, 1]) / ref[:, 1]) < 2e-15


def test_rows_form_matches_pair_form(oracle, golden):
    """The full-matrix row form used for sharding sums the same physics (order differs)."""
    g = golden("force_n500")
    n = int(g["n"])
    p = oracle.derive_params(n, float(g["L"]), 0.005, float(g["rc"]))
    r = np.ascontiguousarray(g["r"])
    ax, ay, az, se, sd, sdd = oracle.rows_raw(p, 0, n, r[0], r[1], r[2])
    te, td, tdd = oracle.tail_corrections(p)
    assert abs((4.0 * 0.5 * se + te) - g["scalars"][0]) < 1e-12 * abs(g["scalars"][0])
    assert abs((24.0 * 0.5 * sd + td) - g["scalars"][1]) < 1e-12 * abs(g["scalars"][1])
    assert abs((24.0 * 0.5 * sdd + tdd) - g["scalars"][2]) < 1e-12 * abs(g["scalars"][2])
    a = 24.0 * np.stack([ax, ay, az])
    assert np.max(np.abs(a - g["a"])) < 1e-12 * np.max(np.abs(g["a"]))


def test_observables_match_reference_energies_file(oracle, golden):
    """T and P formulas (md_means.f90:221,227) against the reference program's own text output
    (7 significant digits)."""
    from ljmd_amd import io_formats
    g = golden("traj_n108")
    p = oracle.derive_params(108, float(g["L"]), float(g["dt"]), float(g["rc"]))
    rows = io_formats.read_energies(GOLDEN / "ref_run_n108_oi10" / "instantaneous_energies.dat")
    assert rows.shape == (90, 6)
    sc = g["scalars"]
    for i, row in enumerate(rows):
        step = 110 + 10 * i
        etot, temp, press = oracle.observables(p, sc[step, 0], sc[step, 1], sc[step, 2])
        mine = np.array([sc[step, 0], sc[step, 1], etot, temp, press])
        assert np.allclose(mine, row[1:], rtol=2e-6, atol=0), (step, mine, row)
