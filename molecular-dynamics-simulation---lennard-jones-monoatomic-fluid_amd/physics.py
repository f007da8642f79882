"""The reference's operator interface for the hot path, served by the HIP library.

  compute_lj_potential_energy(params, state)   scripts/physics/lj_potential_energy.f90:46
  verlet_step(params, state)                   scripts/physics/verlet.f90:41
  minimum_image / wrap_positions               scripts/physics/geometry_pbc.f90:80,39 (host helpers)

Both operators keep the reference's argument meaning: `state` is updated in place,
the scalar outputs are returned as a tuple in the reference's argument order.  They
go through the *stateless* C entry points (ljmd_compute_lj_potential_energy /
ljmd_verlet_step), i.e. exactly what the Fortran shim modules bind.

`Engine` is the resident-state interface (ljmd_create ... ljmd_verlet_steps): state
stays in HBM between calls and only the per-step scalars come back.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import LjmdError, c_double_p
from .md_types import SimParams, SimState


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def _check_array(a, n: int, name: str) -> np.ndarray:
    if a is None:
        raise ValueError(f"{name} is not allocated")
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.ndim != 1 or a.shape[0] != n \
            or not a.flags.c_contiguous:
        raise ValueError(f"{name} must be a contiguous float64 array of length {n}")
    return a


def _guard_force_params(p: SimParams, routine: str) -> None:
    # lj_potential_energy.f90:77-81
    if p.n <= 0:
        raise ValueError(f"{routine}(): params%n must be > 0.")
    if p.box_length <= 0.0:
        raise ValueError(f"{routine}(): params%box_length must be > 0.")
    if p.volume <= 0.0:
        raise ValueError(f"{routine}(): params%volume must be > 0.")
    if p.rc <= 0.0:
        raise ValueError(f"{routine}(): params%rc must be > 0.")
    if p.rc_square <= 0.0:
        raise ValueError(f"{routine}(): params%rc_square must be > 0.")


def compute_lj_potential_energy(params: SimParams, state: SimState):
    """-> (epot, d_epot, dd_epot); overwrites state.ax/ay/az."""
    _guard_force_params(params, "compute_lj_potential_energy")
    if not state.allocated():
        raise ValueError("compute_lj_potential_energy(): state arrays are not allocated.")
    n = params.n
    arrs = [_check_array(getattr(state, k), n, k) for k in ("rx", "ry", "rz", "ax", "ay", "az")]
    e, d, dd = C.c_double(), C.c_double(), C.c_double()
    lib = _lib.load()
    _lib.check(lib.ljmd_compute_lj_potential_energy(
        n, params.box_length, params.rc, *[_ptr(a) for a in arrs],
        C.byref(e), C.byref(d), C.byref(dd)))
    return e.value, d.value, dd.value


def verlet_step(params: SimParams, state: SimState):
    """-> (epot, ekin, d_epot, dd_epot); updates all nine state arrays in place."""
    if params.n <= 0:
        raise ValueError("verlet_step(): params%n must be > 0.")
    if not state.allocated():
        raise ValueError("verlet_step(): state arrays are not allocated.")
    n = params.n
    arrs = [_check_array(getattr(state, k), n, k) for k in SimState.FIELDS]
    out = [C.c_double() for _ in range(4)]
    lib = _lib.load()
    _lib.check(lib.ljmd_verlet_step(n, params.box_length, params.dt, params.rc,
                                    *[_ptr(a) for a in arrs], *[C.byref(o) for o in out]))
    return tuple(o.value for o in out)


def stateless_reset() -> None:
    """Frees the cached engine behind compute_lj_potential_energy / verlet_step (ljmd_stateless_reset)."""
    _lib.load().ljmd_stateless_reset()


def minimum_image(dx: float, box_length: float, inv_box_length: float) -> float:
    """geometry_pbc.f90:80-88 (dnint = round half away from zero); host helper."""
    t = dx * inv_box_length
    n = np.copysign(np.floor(np.abs(t) + 0.5), t)
    return dx - box_length * n


class Engine:
    """HBM-resident simulation on one GPU (or one shard of a multi-GPU run).
    devices=[d0, d1, ...]: ONE process driving several devices (ljmd_create_multi): rank g of len(devices) runs on
    devices[g]; the handle then takes and returns global arrays like a single-GPU one."""

    def __init__(self, params: SimParams, device: int = 0, rank: int = 0, n_ranks: int = 1,
                 precision_mode: int = _lib.PRECISION_FP64, devices=None):
        self._lib = _lib.load()
        self.params = params
        self.rank, self.n_ranks = rank, n_ranks
        h = C.c_void_p()
        if devices is not None:
            devs = (C.c_int32 * len(devices))(*devices)
            _lib.check(self._lib.ljmd_create_multi(C.byref(h), params.n, params.box_length, params.dt, params.rc,
                                                   precision_mode, len(devices), devs))
            self.rank, self.n_ranks = 0, 1          # global arrays in, global arrays out
        else:
            _lib.check(self._lib.ljmd_create(C.byref(h), params.n, params.box_length, params.dt, params.rc,
                                             precision_mode, device, rank, n_ranks))
        self._h = h
        self.shard = params.n // self.n_ranks

    # -- lifecycle ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ljmd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, status: int) -> None:
        _lib.check(status, self._h)

    # -- state transfer ----------------------------------------------------
    def set_state(self, rx, ry, rz, vx, vy, vz) -> None:
        n = self.params.n
        arrs = [_check_array(np.ascontiguousarray(a, dtype=np.float64), n, "state array")
                for a in (rx, ry, rz, vx, vy, vz)]
        self._ck(self._lib.ljmd_set_state(self._h, *[_ptr(a) for a in arrs]))

    def set_accel(self, ax, ay, az) -> None:
        n = self.params.n
        arrs = [_check_array(np.ascontiguousarray(a, dtype=np.float64), n, "accel array") for a in (ax, ay, az)]
        self._ck(self._lib.ljmd_set_accel(self._h, *[_ptr(a) for a in arrs]))

    def set_unwrapped(self, ux, uy, uz) -> None:
        n = self.params.n
        arrs = [_check_array(np.ascontiguousarray(a, dtype=np.float64), n, "unwrapped array") for a in (ux, uy, uz)]
        self._ck(self._lib.ljmd_set_unwrapped(self._h, *[_ptr(a) for a in arrs]))

    def get_state(self, which=("r", "ru", "v", "a")) -> dict:
        """-> {'r': (x,y,z), 'ru': ..., 'v': ..., 'a': ...} of the owned shard."""
        S = self.shard
        out, ptrs = {}, []
        for key in ("r", "ru", "v", "a"):
            if key in which:
                arrs = tuple(np.empty(S, dtype=np.float64) for _ in range(3))
                out[key] = arrs
                ptrs += [_ptr(a) for a in arrs]
            else:
                ptrs += [None, None, None]
        self._ck(self._lib.ljmd_get_state(self._h, *ptrs))
        return out

    # -- hot path ----------------------------------------------------------
    def compute_forces(self):
        e, d, dd = C.c_double(), C.c_double(), C.c_double()
        self._ck(self._lib.ljmd_compute_forces(self._h, C.byref(e), C.byref(d), C.byref(dd)))
        return e.value, d.value, dd.value

    def verlet_steps(self, nsteps: int):
        """-> (epot[nsteps], ekin[nsteps], d_epot[nsteps], dd_epot[nsteps])"""
        outs = [np.empty(nsteps, dtype=np.float64) for _ in range(4)]
        self._ck(self._lib.ljmd_verlet_steps(self._h, nsteps, *[_ptr(o) for o in outs]))
        return tuple(outs)

    def advance(self, nsteps: int) -> None:
        """nsteps Verlet steps whose observables nobody reads (ljmd_verlet_steps with NULL outputs, as the warm-up of
        the initial-configuration driver): forces-only pair kernel, same trajectory bit for bit"""
        self._ck(self._lib.ljmd_verlet_steps(self._h, nsteps, None, None, None, None))

    # -- asynchronous production loop (snapshot I/O overlapped with the next steps) --------
    def enqueue_steps(self, nsteps: int, sampled: bool = False) -> None:
        """sampled: only the last of the nsteps evaluates epot, d_epot, dd_epot (the step the reference samples,
        md_simulation_program.f90:361); collect_steps returns NaN for the others.  r, v, a, ekin are unchanged."""
        fn = self._lib.ljmd_enqueue_steps_sampled if sampled else self._lib.ljmd_enqueue_steps
        self._ck(fn(self._h, nsteps))

    def migrations(self) -> int:
        """ownership migrations so far (multi-device handle: also the automatic ones, LJMD_MULTI_MIGRATE_EVERY)"""
        return int(self._lib.ljmd_multi_migrations(self._h))

    # -- ownership migration (multi-GPU; include/ljmd.h: ljmd_migrate) ------------------------
    def migrate(self) -> None:
        """deal the particles out to the ranks again by position, on the devices, collectives included (multi-device
        handle, or a rank engine with an RCCL communicator: every rank calls it at the same step)"""
        self._ck(self._lib.ljmd_migrate(self._h))

    def migrate_pack(self) -> None:
        self._ck(self._lib.ljmd_migrate_pack(self._h))

    def migrate_buffer(self):
        """-> (device address, total doubles, own offset, own count) of the migration buffer"""
        tot, off, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        p = self._lib.ljmd_migrate_buffer(self._h, C.byref(tot), C.byref(off), C.byref(cnt))
        return p, tot.value, off.value, cnt.value

    def migrate_deal(self) -> None:
        self._ck(self._lib.ljmd_migrate_deal(self._h))

    def particle_ids(self) -> np.ndarray:
        """ids[j] = index, in the arrays given to set_state, of the particle at position j of this engine's arrays"""
        n = self.shard if self.n_ranks > 1 else self.params.n
        ids = np.empty(n, dtype=np.int32)
        self._ck(self._lib.ljmd_particle_ids(self._h, ids.ctypes.data_as(_lib.c_int32_p)))
        return ids

    def set_observables(self, on: bool) -> None:
        """phase API (sharded engines): forces-only force evaluations while off"""
        self._ck(self._lib.ljmd_set_observables(self._h, 1 if on else 0))

    def collect_steps(self, nsteps: int):
        outs = [np.empty(nsteps, dtype=np.float64) for _ in range(4)]
        self._ck(self._lib.ljmd_collect_steps(self._h, nsteps, *[_ptr(o) for o in outs]))
        return tuple(outs)

    def snapshot_begin(self) -> None:
        self._ck(self._lib.ljmd_snapshot_begin(self._h))

    def snapshot_end(self) -> dict:
        """-> {'r': (x,y,z), 'ru': ..., 'v': ..., 'a': ...} as of the matching snapshot_begin."""
        out = {key: tuple(np.empty(self.shard, dtype=np.float64) for _ in range(3)) for key in ("r", "ru", "v", "a")}
        ptrs = [_ptr(a) for key in ("r", "ru", "v", "a") for a in out[key]]
        self._ck(self._lib.ljmd_snapshot_end(self._h, *ptrs))
        return out

    def kinetic_energy(self) -> float:
        k = C.c_double()
        self._ck(self._lib.ljmd_kinetic_energy(self._h, C.byref(k)))
        return k.value

    # -- split phase (multi-GPU) ---------------------------------------------
    def shard_range(self):
        i0, i1 = C.c_int32(), C.c_int32()
        self._ck(self._lib.ljmd_shard_range(self._h, C.byref(i0), C.byref(i1)))
        return i0.value, i1.value

    def exchange_buffer(self):
        """-> (device address, total doubles, own offset, own count)"""
        tot, off, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        p = self._lib.ljmd_exchange_buffer(self._h, C.byref(tot), C.byref(off), C.byref(cnt))
        return p, tot.value, off.value, cnt.value

    def device_ptr(self, which: int, axis: int) -> int:
        return self._lib.ljmd_device_ptr(self._h, which, axis)

    def stream(self) -> int:
        return self._lib.ljmd_stream(self._h)

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.load().ljmd_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes) -> None:
        assert len(unique_id) == _lib.COMM_ID_BYTES
        self._ck(self._lib.ljmd_comm_init(self._h, unique_id))

    def comm_size(self) -> int:
        """Ranks in the engine's RCCL communicator as RCCL reports them (0 = none)."""
        return int(self._lib.ljmd_comm_size(self._h))

    def allgather_positions(self) -> None:
        self._ck(self._lib.ljmd_allgather_positions(self._h))

    def memcpy(self, dst: int, src: int, nbytes: int, kind: int) -> None:
        """kind 1 = host->device, 2 = device->host, 3 = device->device (raw addresses)."""
        self._ck(self._lib.ljmd_memcpy(self._h, dst, src, nbytes, kind))

    def synchronize(self) -> None:
        self._ck(self._lib.ljmd_synchronize(self._h))

    def step_begin(self) -> None:
        self._ck(self._lib.ljmd_step_begin(self._h))

    def step_finish(self) -> None:
        self._ck(self._lib.ljmd_step_finish(self._h))

    def step_forces(self) -> None:
        self._ck(self._lib.ljmd_step_forces(self._h))

    def force_buffers(self, external: bool):
        """-> (fpart address, doubles, frecv address, doubles); see include/ljmd.h"""
        fp, fr = C.c_void_p(), C.c_void_p()
        nfp, nfr = C.c_int64(), C.c_int64()
        self._ck(self._lib.ljmd_force_buffers(self._h, 1 if external else 0, C.byref(fp), C.byref(nfp),
                                              C.byref(fr), C.byref(nfr)))
        return fp.value, nfp.value, fr.value, nfr.value

    def forces_partial(self) -> None:
        self._ck(self._lib.ljmd_forces_partial(self._h))

    def read_partials(self, nsteps: int) -> np.ndarray:
        out = np.empty((nsteps, _lib.PARTIAL_STRIDE), dtype=np.float64)
        self._ck(self._lib.ljmd_read_partials(self._h, nsteps, _ptr(out)))
        return out

    def combine_scalars(self, partials_by_rank: np.ndarray):
        """partials_by_rank: [n_ranks, PARTIAL_STRIDE] of ONE step -> (epot, ekin, d_epot, dd_epot)"""
        p = np.ascontiguousarray(partials_by_rank, dtype=np.float64)
        outs = [C.c_double() for _ in range(4)]
        self._ck(self._lib.ljmd_combine_scalars(self._h, _ptr(p), p.shape[0], *[C.byref(o) for o in outs]))
        return tuple(o.value for o in outs)

    def set_tail_corrections(self, on: bool) -> None:
        """the reference's compile-time switch use_tail_corrections (lj_potential_energy.f90:36): off = epot, d_epot,
        dd_epot without the three mean-field tail constants"""
        self._ck(self._lib.ljmd_set_tail_corrections(self._h, 1 if on else 0))

    # -- measurement -----------------------------------------------------------
    def profile_enable(self, on: bool = True) -> None:
        self._ck(self._lib.ljmd_profile_enable(self._h, 1 if on else 0))

    def pair_kernel_name(self) -> str:
        return self._lib.ljmd_pair_kernel_name(self._h).decode()

    def profile_read(self) -> dict:
        """-> {'pair_ms', 'geometry_ms', 'drift_ms', 'reduce_ms', 'launches'} averages per launch,
        plus '<name>_min': the shortest launch of each interval"""
        ms, lo = (C.c_double * 4)(), (C.c_double * 4)()
        c = C.c_int32()
        self._ck(self._lib.ljmd_profile_read_ex(self._h, ms, lo, C.byref(c)))
        out = {"pair_ms": ms[0], "geometry_ms": ms[1], "drift_ms": ms[2], "reduce_ms": ms[3], "launches": c.value}
        out.update({"pair_ms_min": lo[0], "geometry_ms_min": lo[1], "drift_ms_min": lo[2], "reduce_ms_min": lo[3]})
        return out

    def profile_read_rank(self, rank: int) -> dict:
        """ljmd_profile_read_rank: the same per rank engine, plus the two exchanges of a multi-GPU step
        ('pos_exchange_ms': position all-gather, 'force_exchange_ms': reduce-scatter / all-to-all)"""
        ms, lo, med = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_double * 6)()
        c = C.c_int32()
        self._ck(self._lib.ljmd_profile_read_stats(self._h, rank, ms, lo, med, C.byref(c)))
        names = ("pair_ms", "geometry_ms", "drift_ms", "reduce_ms", "pos_exchange_ms", "force_exchange_ms")
        out = {k: ms[i] for i, k in enumerate(names)}
        out.update({k + "_min": lo[i] for i, k in enumerate(names)})
        out.update({k + "_median": med[i] for i, k in enumerate(names)})
        out["launches"] = c.value
        return out


def observables(params: SimParams, epot: float, ekin: float, d_epot: float):
    """etot, T, P of one sample: md_simulation_program.f90:355,366 + md_means.f90:215-228.
    Note T uses 3N (not 3N-3) in the time series."""
    npd = float(params.n)
    rho = npd / params.volume
    virial = -d_epot
    etot = epot + ekin
    temp = 2.0 * ekin / (3.0 * npd)
    press = rho * temp + virial / (3.0 * params.volume)
    return etot, temp, press
