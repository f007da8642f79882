"""CPU, world_size 2, gloo: the sharding / exchange / scalar-combination logic of
ljmd_amd.distributed.ShardedSimulation -- the same class bench.py drives with RCCL on GPUs.

The per-rank compute engine is INJECTED: here an oracle-backed stand-in (tests only; the product
never imports the oracle) that implements the engine protocol on numpy arrays:
  set_state / forces_partial / step_begin / step_finish / read_partials / combine_scalars
so that what is exercised is exactly the product's host logic: shard ranges, the in-place
all-gather of the position block (issued through the engine), the RCCL-id bootstrap, the per-step partial records and their rank-ordered reduction.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

STRIDE = 8


class OracleShardEngine:
    """Engine protocol on the CPU for ONE rank (full-matrix rows of its shard via the oracle)."""

    def __init__(self, params, rank, world):
        from oracle import oracle as O
        self.O = O
        self.p = params
        self.po = O.derive_params(params.n, params.box_length, params.dt, params.rc)
        self.rank, self.world = rank, world
        self.S = params.n // world
        self.buf = np.zeros((world, 3, self.S))          # exchange buffer, shard-blocked SoA
        self.records = []
        self.ids = np.arange(rank * self.S, (rank + 1) * self.S)
        self.mig = np.zeros((world, 10, self.S))         # migration buffer: ru, v, a, id per slot
        self.observables = []

    # ---- what ShardedSimulation needs -------------------------------------------------
    def allgather_positions(self):
        """Stand-in for ljmd_allgather_positions: the same in-place all-gather, over gloo."""
        import torch
        import torch.distributed as dist
        full = torch.from_numpy(self.buf.reshape(-1))
        cnt = 3 * self.S
        dist.all_gather_into_tensor(full, full[self.rank * cnt:(self.rank + 1) * cnt])

    # -- host-staged fallback protocol (raw addresses of numpy buffers) -----------------------
    def exchange_buffer(self):
        cnt = 3 * self.S
        return self.buf.ctypes.data, self.buf.size, self.rank * cnt, cnt

    def force_buffers(self, external):
        return 0, 0, 0, 0                                   # gather-style engine: nothing to reduce

    def memcpy(self, dst, src, nbytes, kind):
        import ctypes
        ctypes.memmove(dst, src, nbytes)

    def step_forces(self):
        pass

    def comm_unique_id(self):
        return b"fake-rccl-id".ljust(128, b"\0")

    def comm_init(self, uid):
        assert uid == self.comm_unique_id()
        assert not getattr(self, "comm_ready", False), "communicator initialised twice"
        self.comm_ready = True

    # -- ownership migration protocol (include/ljmd.h: ljmd_migrate*), any balanced deal by position will do here ----
    def migrate_pack(self):
        self.mig[self.rank] = np.concatenate([self.ru, self.v, self.a, self.ids[None].astype(float)])

    def migrate_buffer(self):
        cnt = 10 * self.S
        return self.mig.ctypes.data, self.mig.size, self.rank * cnt, cnt

    def migrate_deal(self):
        S = self.S
        order = np.argsort(self.buf[:, 0, :].reshape(-1), kind="stable")      # slabs along x
        mine = order[self.rank * S:(self.rank + 1) * S]
        g, s = mine // S, mine % S
        newpos = self.buf[g, :, s].T.copy()
        self.ru, self.v, self.a = (self.mig[g, 3 * k:3 * k + 3, s].T.copy() for k in range(3))
        self.ids = self.mig[g, 9, s].astype(np.int64)
        self.buf[self.rank] = newpos
        self.migrations = getattr(self, "migrations", 0) + 1

    def migrate(self):
        """the RCCL form: the library does both all-gathers itself"""
        import torch
        import torch.distributed as dist
        self.migrate_pack()
        full = torch.from_numpy(self.mig.reshape(-1))
        cnt = 10 * self.S
        dist.all_gather_into_tensor(full, full[self.rank * cnt:(self.rank + 1) * cnt].clone())
        self.migrate_deal()
        self.allgather_positions()

    def particle_ids(self):
        return self.ids.copy()

    def set_observables(self, on):
        self.observables.append(bool(on))

    def set_state(self, rx, ry, rz, vx, vy, vz):
        S, g = self.S, self.rank
        self.ids = np.arange(g * S, (g + 1) * S)
        for r in range(self.world):
            self.buf[r] = np.stack([rx[r * S:(r + 1) * S], ry[r * S:(r + 1) * S], rz[r * S:(r + 1) * S]])
        sl = slice(g * S, (g + 1) * S)
        self.v = np.stack([vx[sl], vy[sl], vz[sl]]).copy()
        self.ru = self.buf[g].copy()
        self.a = np.zeros((3, S))

    def _all_positions(self):
        return [np.ascontiguousarray(np.concatenate([self.buf[r][k] for r in range(self.world)])) for k in range(3)]

    def _forces(self, kick):
        S, g = self.S, self.rank
        x, y, z = self._all_positions()
        ax, ay, az, se, sd, sdd = self.O.rows_raw(self.po, g * S, (g + 1) * S, x, y, z)
        self.a = 24.0 * np.stack([ax, ay, az])
        s12, s6 = -(se + sd), -(2.0 * se + sd)            # invert epot = s12 - s6, d = -2 s12 + s6
        rec = np.zeros(STRIDE)
        rec[0], rec[1] = 0.5 * s12, 0.5 * s6              # ordered pairs -> unordered
        if kick:
            self.v = self.v + self.a * self.p.dt_half
            rec[2:5] = (self.v * self.v).sum(axis=1)
        self.records.append(rec)

    def forces_partial(self):
        self._forces(False)

    def step_begin(self):
        p, L = self.p, self.p.box_length
        r0 = self.buf[self.rank].copy()
        r1 = (r0 + self.v * p.dt) + self.a * p.dt_square_half
        r1 = r1 - L * np.floor(r1 * p.inv_box_length)
        d = r1 - r0
        t = d * p.inv_box_length
        d = d - L * np.copysign(np.floor(np.abs(t) + 0.5), t)
        self.buf[self.rank][:] = r1
        self.v = self.v + self.a * p.dt_half
        self.ru = self.ru + d

    def step_finish(self):
        self._forces(True)

    def read_partials(self, nsteps):
        out = np.stack(self.records[-nsteps:]) if nsteps else np.zeros((0, STRIDE))
        self.records.clear()
        return out

    def combine_scalars(self, parts):
        s = parts.sum(axis=0) if parts.shape[0] > 1 else parts[0]
        s12 = sum(parts[g, 0] for g in range(parts.shape[0]))
        s6 = sum(parts[g, 1] for g in range(parts.shape[0]))
        kx, ky, kz = (sum(parts[g, c] for g in range(parts.shape[0])) for c in (2, 3, 4))
        te, td, tdd = self.O.tail_corrections(self.po)
        return (4.0 * (s12 - s6) + te, 0.5 * (kx + ky + kz), 24.0 * (-2.0 * s12 + s6) + td,
                24.0 * (26.0 * s12 - 7.0 * s6) + tdd)


def _worker(rank, world, port, n, nsteps, outdir, exchange="rccl", migrate_every=0):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import ljmd_amd  # noqa: F401
    from ljmd_amd import distributed, synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, r, v = synthetic.make_config(n, seed=99)
        eng = OracleShardEngine(p, rank, world)
        distributed.bootstrap_rccl(eng, rank, world)
        assert eng.comm_ready
        eng.comm_ready = False
        assert distributed.try_bootstrap_rccl(eng, rank, world)
        sim = distributed.ShardedSimulation(eng, rank, world, exchange=exchange, migrate_every=migrate_every)
        e0, d0, dd0 = sim.start(r, v)
        if migrate_every:
            # two segments, the second one sampled: a migration at start(), one before the second segment, and the
            # caller's observables switch survives the sampled segment
            sim.set_observables(False)
            h = nsteps // 2
            parts = [sim.run(h)]
            sim.enqueue_steps(nsteps - h, sampled=True)
            parts.append(sim.collect(nsteps - h))
            e, k, d, dd = (np.concatenate([a[c] for a in parts]) for c in range(4))
            assert eng.migrations == 2 and eng.observables[-1] is False and eng.observables[-2] is True
        else:
            e, k, d, dd = sim.run(nsteps)
        np.savez(Path(outdir) / f"rank{rank}.npz", t0=np.array([e0, d0, dd0]), sc=np.stack([e, k, d, dd], axis=1),
                 buf=eng.buf, v=eng.v, ru=eng.ru, a=eng.a, ids=sim.particle_ids())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,exchange,migrate_every", [(2, "rccl", 0), (2, "host", 0), (2, "rccl", 6), (2, "host", 6)])
def test_sharded_run_matches_single_process_oracle(tmp_path, oracle, world, exchange, migrate_every):
    """migrate_every > 0: the particles are dealt out by position at start() and again before the second segment -- the
    ranks then own SETS of particles, identified by particle_ids(), and the exchange buffer is in the order of the deal"""
    import torch.multiprocessing as mp
    from ljmd_amd import synthetic
    n, nsteps = 432, 12
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nsteps, str(tmp_path), exchange, migrate_every))
             for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(180)
        assert p_.exitcode == 0
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    # the pinned single-process oracle on the same start
    p, r, v = synthetic.make_config(n, seed=99)
    po = oracle.derive_params(n, p.box_length, p.dt, p.rc)
    e0, d0, dd0, ax, ay, az = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    st = {"rx": r[0].copy(), "ry": r[1].copy(), "rz": r[2].copy(), "ux": r[0].copy(), "uy": r[1].copy(),
          "uz": r[2].copy(), "vx": v[0].copy(), "vy": v[1].copy(), "vz": v[2].copy(), "ax": ax, "ay": ay, "az": az}
    sc = oracle.run_steps(po, nsteps, st)

    S = n // world
    for rank in range(world):
        out = res[rank]
        # every rank computed the same global scalars (rank-ordered host reduction)
        assert np.array_equal(out["t0"], res[0]["t0"]) and np.array_equal(out["sc"], res[0]["sc"])
        assert np.allclose(out["t0"], [e0, d0, dd0], rtol=1e-12, atol=0)
        assert np.max(np.abs(out["sc"] - sc) / np.abs(sc)) < 1e-10
        # after the last all-gather every rank holds ALL positions, in shard-blocked order of the current owners
        allpos = np.stack([np.concatenate([out["buf"][g][k] for g in range(world)]) for k in range(3)])
        owners = np.concatenate([res[g]["ids"] for g in range(world)])
        assert np.array_equal(np.sort(owners), np.arange(n))                  # every particle owned exactly once
        assert np.max(np.abs(allpos - np.stack([st["rx"], st["ry"], st["rz"]])[:, owners])) < 1e-11
        assert np.array_equal(out["buf"], res[0]["buf"])
        sl = out["ids"]
        assert migrate_every or np.array_equal(sl, np.arange(rank * S, (rank + 1) * S))
        assert np.max(np.abs(out["v"] - np.stack([st["vx"][sl], st["vy"][sl], st["vz"][sl]]))) < 1e-10
        assert np.max(np.abs(out["ru"] - np.stack([st["ux"][sl], st["uy"][sl], st["uz"][sl]]))) < 1e-11


def test_world_size_one_needs_no_process_group(oracle):
    """N = 1 path of bench.py: ShardedSimulation must not touch torch.distributed collectives."""
    from ljmd_amd import distributed, synthetic
    p, r, v = synthetic.make_config(256, seed=3)
    eng = OracleShardEngine(p, 0, 1)
    distributed.bootstrap_rccl(eng, 0, 1)
    sim = distributed.ShardedSimulation(eng, 0, 1)
    e0, d0, dd0 = sim.start(r, v)
    po = oracle.derive_params(256, p.box_length, p.dt, p.rc)
    ref = oracle.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
    assert abs(e0 - ref[0]) < 1e-12 * abs(ref[0])
    e, k, d, dd = sim.run(3)
    assert e.shape == (3,) and np.all(np.isfinite(k))
