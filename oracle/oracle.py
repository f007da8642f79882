"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE (parity oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  See oracle/ljmd_oracle.c for the contract.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
REF_DIR = HERE / "_ref"
REF_HARNESS = REF_DIR / "ref_harness"

dp = C.POINTER(C.c_double)


class OraParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("num_cells", C.c_int32),
                ("box_length", C.c_double), ("inv_box_length", C.c_double),
                ("volume", C.c_double), ("density", C.c_double),
                ("dt", C.c_double), ("dt_half", C.c_double), ("dt_square_half", C.c_double),
                ("rc", C.c_double), ("rc_square", C.c_double)]


class Ran3State(C.Structure):
    _fields_ = [("ma", C.c_double * 56), ("inext", C.c_int32), ("inextp", C.c_int32), ("iff", C.c_int32)]


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(HERE), "liboracle.so"], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        L = C.CDLL(str(LIB))
        L.ora_derive_params.restype = C.c_int
        L.ora_derive_params.argtypes = [C.POINTER(OraParams), C.c_int32, C.c_double, C.c_double, C.c_double]
        L.ora_minimum_image.restype = C.c_double
        L.ora_minimum_image.argtypes = [C.c_double] * 3
        L.ora_wrap_positions.argtypes = [dp, dp, dp, C.c_int32, C.c_double]
        L.ora_tail_corrections.argtypes = [C.POINTER(OraParams), dp, dp, dp]
        L.ora_set_tail_corrections.argtypes = [C.c_int]
        L.ora_compute_lj_potential_energy.argtypes = [C.POINTER(OraParams)] + [dp] * 9
        L.ora_verlet_step.argtypes = [C.POINTER(OraParams)] + [dp] * 13
        L.ora_ekin_fused.restype = C.c_double
        L.ora_ekin_fused.argtypes = [dp, dp, dp, C.c_int32]
        L.ora_unwrapped_update.argtypes = [C.POINTER(OraParams)] + [dp] * 9
        L.ora_observables.argtypes = [C.POINTER(OraParams), C.c_double, C.c_double, C.c_double, dp, dp, dp]
        L.ora_run_steps.argtypes = [C.POINTER(OraParams), C.c_int32] + [dp] * 13
        L.ora_rows_raw.argtypes = [C.POINTER(OraParams), C.c_int32, C.c_int32] + [dp] * 9
        L.ora_ran3.restype = C.c_double
        L.ora_ran3.argtypes = [C.POINTER(Ran3State), C.POINTER(C.c_int32)]
        L.ora_build_fcc_lattice.argtypes = [C.c_int32, C.c_double, dp, dp, dp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(dp)


def derive_params(n: int, box_length: float, dt: float, rc: float) -> OraParams:
    p = OraParams()
    code = lib().ora_derive_params(C.byref(p), n, box_length, dt, rc)
    if code != 0:
        raise ValueError(f"oracle: parameter guard {code} violated")
    return p


def minimum_image(dx, L, invL) -> float:
    return lib().ora_minimum_image(dx, L, invL)


def wrap_positions(rx, ry, rz, L) -> None:
    lib().ora_wrap_positions(_p(rx), _p(ry), _p(rz), len(rx), L)


def tail_corrections(p: OraParams):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().ora_tail_corrections(C.byref(p), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def set_tail_corrections(on: bool) -> None:
    """the reference's compile-time switch use_tail_corrections (lj_potential_energy.f90:36); True = as shipped"""
    lib().ora_set_tail_corrections(1 if on else 0)


def compute_forces(p: OraParams, rx, ry, rz):
    """-> (epot, d_epot, dd_epot, ax, ay, az)"""
    n = p.n
    ax, ay, az = (np.empty(n) for _ in range(3))
    e, d, dd = C.c_double(), C.c_double(), C.c_double()
    lib().ora_compute_lj_potential_energy(C.byref(p), _p(rx), _p(ry), _p(rz), _p(ax), _p(ay), _p(az),
                                          C.byref(e), C.byref(d), C.byref(dd))
    return e.value, d.value, dd.value, ax, ay, az


def verlet_step(p: OraParams, st: dict):
    """st: dict of the nine arrays (rx..az), updated in place -> (epot, ekin, d_epot, dd_epot)"""
    outs = [C.c_double() for _ in range(4)]
    lib().ora_verlet_step(C.byref(p), *[_p(st[k]) for k in ("rx", "ry", "rz", "vx", "vy", "vz", "ax", "ay", "az")],
                          *[C.byref(o) for o in outs])
    return tuple(o.value for o in outs)


def ekin_fused(vx, vy, vz) -> float:
    return lib().ora_ekin_fused(_p(vx), _p(vy), _p(vz), len(vx))


def run_steps(p: OraParams, nsteps: int, st: dict) -> np.ndarray:
    """st: dict with rx..rz, ux..uz, vx..vz, ax..az (12 arrays), updated in place.
    -> scalars[nsteps, 4] = epot, ekin, d_epot, dd_epot after each step."""
    sc = np.empty((nsteps, 4))
    keys = ("rx", "ry", "rz", "ux", "uy", "uz", "vx", "vy", "vz", "ax", "ay", "az")
    lib().ora_run_steps(C.byref(p), nsteps, *[_p(st[k]) for k in keys], _p(sc))
    return sc


def observables(p: OraParams, epot, ekin, d_epot):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().ora_observables(C.byref(p), epot, ekin, d_epot, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def rows_raw(p: OraParams, i0: int, i1: int, rx, ry, rz):
    """Full-matrix rows [i0,i1): -> (ax, ay, az raw [i1-i0], s_epot, s_d, s_dd raw, double counted)"""
    m = i1 - i0
    ax, ay, az = (np.empty(m) for _ in range(3))
    e, d, dd = C.c_double(), C.c_double(), C.c_double()
    lib().ora_rows_raw(C.byref(p), i0, i1, _p(rx), _p(ry), _p(rz), _p(ax), _p(ay), _p(az),
                       C.byref(e), C.byref(d), C.byref(dd))
    return ax, ay, az, e.value, d.value, dd.value


def ran3_sequence(seed: int, count: int) -> np.ndarray:
    st = Ran3State()
    s = C.c_int32(seed)
    return np.array([lib().ora_ran3(C.byref(st), C.byref(s)) for _ in range(count)])


def fcc_lattice(num_cells: int, box_length: float):
    n = 4 * num_cells ** 3
    rx, ry, rz = (np.empty(n) for _ in range(3))
    lib().ora_build_fcc_lattice(num_cells, box_length, _p(rx), _p(ry), _p(rz))
    return rx, ry, rz


# ---- the real reference (oracle/_ref), when built -------------------------------------

def ref_available() -> bool:
    return REF_HARNESS.exists()


def write_case(path, n, L, dt, rc, rx, ry, rz, vx, vy, vz) -> None:
    with open(path, "wb") as f:
        f.write(np.int32(n).tobytes())
        f.write(np.array([L, dt, rc], dtype=np.float64).tobytes())
        for a in (rx, ry, rz, vx, vy, vz):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())


def ref_force(workdir, n, L, dt, rc, rx, ry, rz):
    workdir = Path(workdir)
    z = np.zeros(n)
    write_case(workdir / "in.bin", n, L, dt, rc, rx, ry, rz, z, z, z)
    subprocess.run([str(REF_HARNESS), "force", str(workdir / "in.bin"), str(workdir / "out.bin")], check=True)
    raw = np.fromfile(workdir / "out.bin", dtype=np.float64)
    return raw[0], raw[1], raw[2], raw[3:3 + n].copy(), raw[3 + n:3 + 2 * n].copy(), raw[3 + 2 * n:].copy()


def ref_traj(workdir, n, L, dt, rc, rx, ry, rz, vx, vy, vz, nsteps):
    """-> (scalars[nsteps+1, 4] incl. t=0, final dict of 12 arrays)"""
    workdir = Path(workdir)
    write_case(workdir / "in.bin", n, L, dt, rc, rx, ry, rz, vx, vy, vz)
    subprocess.run([str(REF_HARNESS), "traj", str(workdir / "in.bin"), str(nsteps), str(workdir / "out.bin")],
                   check=True)
    raw = np.fromfile(workdir / "out.bin", dtype=np.float64)
    sc = raw[:4 * (nsteps + 1)].reshape(nsteps + 1, 4).copy()
    rest = raw[4 * (nsteps + 1):].reshape(12, n)
    keys = ("rx", "ry", "rz", "ux", "uy", "uz", "vx", "vy", "vz", "ax", "ay", "az")
    return sc, {k: rest[i].copy() for i, k in enumerate(keys)}


def ref_bench(n: int, ncalls: int):
    """-> (seconds_total, pairs_per_second) of the real reference force routine, 1 core."""
    out = subprocess.run([str(REF_HARNESS), "bench", str(n), str(ncalls)], check=True, capture_output=True,
                         text=True).stdout.split()
    return float(out[2]), float(out[3])


# ---- trajectory analysis pair pass (scripts/md_one_run_analysis.py:556-584) --------------------

def rdf_histogram_np(x, y, z, L, nbins, rmax, hist) -> None:
    """Adds ONE snapshot's pair-distance counts to hist (uint64[nbins]); every unordered pair counts 2,
    same arithmetic, operation for operation, as the reference's inner loop (numpy, no FMA)."""
    dr = rmax / nbins
    n = len(x)
    for i in range(n - 1):
        dx = x[i + 1:] - x[i]
        dy = y[i + 1:] - y[i]
        dz = z[i + 1:] - z[i]
        dx -= L * np.rint(dx / L)
        dy -= L * np.rint(dy / L)
        dz -= L * np.rint(dz / L)
        r = np.sqrt(dx * dx + dy * dy + dz * dz)
        bins = (r[r < rmax] / dr).astype(int)
        np.add.at(hist, bins, np.uint64(2))
