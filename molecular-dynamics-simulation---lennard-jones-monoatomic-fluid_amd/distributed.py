"""Multi-GPU run of the hot path: one process per GPU, launched by torch.distributed.run.

Sharding: rank g integrates the contiguous particle range [g*S, (g+1)*S),
S = N / G, and evaluates those rows of the pair matrix against ALL N positions.

Data path -- exactly one exchange per MD step: an in-place RCCL all-gather (over xGMI) of the
freshly drifted position block, 3*P doubles per rank, issued by the engine itself
(`ljmd_allgather_positions`) on its own HIP stream: behind the drift/half-kick kernel, ahead of
the pair kernel, no host synchronisation.  The library talks to RCCL directly; its communicator
is bootstrapped here by shipping rank 0's 128-byte RCCL unique id through the
`torch.distributed` process group (`bootstrap_rccl`).

Control plane -- `torch.distributed` (backend gloo: CPU tensors only): the unique-id broadcast,
barriers, and ONE gather per `run()` of the per-step partial records (S12, S6, Kx, Ky, Kz per
rank), which are then combined on the host in fixed rank order, so the scalars do not depend on
collective internals and every rank computes bit-identical values.

The engine object is injected: the product uses `ljmd_amd.Engine` (HIP); the CPU tests inject an
oracle-backed stand-in to exercise this file's logic with world_size 2 over gloo.
"""
from __future__ import annotations

import numpy as np

PARTIAL_STRIDE = 8


def bootstrap_rccl(engine, rank: int, world: int, group=None) -> None:
    """Creates the engine's RCCL communicator: rank 0 draws the unique id, everybody joins."""
    if world == 1:
        return
    import torch.distributed as dist
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0])


def try_bootstrap_rccl(engine, rank: int, world: int, group=None) -> bool:
    """bootstrap_rccl, but agreed on by all ranks: returns False everywhere if it failed anywhere,
    so that the caller can switch every rank to the host-staged exchange."""
    if world == 1:
        return True
    import torch
    import torch.distributed as dist
    ok = 1
    try:
        bootstrap_rccl(engine, rank, world, group)
    except Exception as exc:  # noqa: BLE001 - any failure means "no RCCL on this node"
        print(f"[ljmd] rank {rank}: RCCL bootstrap failed: {exc}", flush=True)
        ok = 0
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


class ShardedSimulation:
    """exchange = "rccl": the engine's in-library RCCL collectives (the product path).
    exchange = "host": SAFETY NET only -- if RCCL cannot be initialised on a node, positions and
    partial accelerations are staged through pinned host memory and gloo; compute stays on the GPU.
    Slower (PCIe + TCP per step), reported as such by bench.py."""

    def __init__(self, engine, rank: int, world: int, group=None, exchange: str = "rccl", migrate_every: int = 2000):
        self.engine = engine
        self.rank, self.world, self.group = rank, world, group
        self.exchange = exchange
        # ownership migration (include/ljmd.h: ljmd_migrate): at start() -- the caller's index ranges may have nothing to
        # do with space -- and then every `migrate_every` steps (0 = never; the ranks then own index ranges for good)
        self.migrate_every = migrate_every
        self.steps_since_migration = 0
        self.observables = True
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            if exchange == "host":
                self._init_host_exchange()

    # -- host-staged fallback ---------------------------------------------------------
    def _init_host_exchange(self) -> None:
        import torch
        ptr, total, off, cnt = self.engine.exchange_buffer()
        self._x = dict(ptr=ptr, total=total, off=off, cnt=cnt, full=torch.empty(total, dtype=torch.float64))
        fp, nfp, fr, nfr = self.engine.force_buffers(True)      # the engine must not call RCCL itself
        self._f = dict(fp=fp, nfp=nfp, fr=fr, nfr=nfr,
                       part=torch.empty(max(nfp, 1), dtype=torch.float64))

    def _host_allgather(self) -> None:
        x = self._x
        own = x["full"][x["off"]:x["off"] + x["cnt"]]
        self.engine.memcpy(own.data_ptr(), x["ptr"] + 8 * x["off"], 8 * x["cnt"], 2)
        self.dist.all_gather_into_tensor(x["full"], own, group=self.group)
        self.engine.memcpy(x["ptr"], x["full"].data_ptr(), 8 * x["total"], 1)

    def _host_reduce_scatter(self) -> None:
        f = self._f
        if f["nfr"] == 0:
            return
        self.engine.memcpy(f["part"].data_ptr(), f["fp"], 8 * f["nfp"], 2)
        self.dist.all_reduce(f["part"], group=self.group)
        blk = f["part"][self.rank * f["nfr"]:(self.rank + 1) * f["nfr"]]
        self.engine.memcpy(f["fr"], blk.data_ptr(), 8 * f["nfr"], 1)

    # -- the data-path collectives ------------------------------------------------------
    def exchange_positions(self) -> None:
        if self.world == 1:
            return
        if self.exchange == "host":
            self._host_allgather()
        else:
            self.engine.allgather_positions()

    def _finish(self, kick: bool) -> None:
        """pair forces + cross-rank force reduction + (kick) for one evaluation."""
        if self.world > 1 and self.exchange == "host":
            self.engine.step_forces()
            self._host_reduce_scatter()
        if kick:
            self.engine.step_finish()          # RCCL reduce-scatter happens inside (exchange = "rccl")
        else:
            self.engine.forces_partial()

    def _gather_partials(self, mine: np.ndarray) -> np.ndarray:
        """mine: [k, PARTIAL_STRIDE] -> [world, k, PARTIAL_STRIDE] (host, tiny)."""
        if self.world == 1:
            return mine[None]
        import torch
        t = torch.from_numpy(np.ascontiguousarray(mine)).reshape(-1)
        out = torch.empty(self.world * t.numel(), dtype=t.dtype)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.numpy().reshape((self.world,) + tuple(mine.shape))

    def _combine(self, parts: np.ndarray):
        """parts [world, k, stride] -> four arrays of length k, ranks added in rank order."""
        k = parts.shape[1]
        cols = [np.empty(k) for _ in range(4)]
        for s in range(k):
            vals = self.engine.combine_scalars(np.ascontiguousarray(parts[:, s, :]))
            for c, val in zip(cols, vals):
                c[s] = val
        return tuple(cols)

    # -- ownership migration ------------------------------------------------------------
    def migrate(self) -> None:
        """Every rank, between two steps: the particles are dealt out again by position, on the devices.  With the
        library's RCCL exchange one call does it all; the host-staged form moves the G blocks of the migration buffer
        through gloo between the two device phases and repeats the position exchange."""
        if self.world == 1:
            return
        if self.exchange == "host":
            import torch
            self.engine.migrate_pack()
            ptr, total, off, cnt = self.engine.migrate_buffer()
            full = torch.empty(total, dtype=torch.float64)
            own = full[off:off + cnt]
            self.engine.memcpy(own.data_ptr(), ptr + 8 * off, 8 * cnt, 2)
            self.dist.all_gather_into_tensor(full, own, group=self.group)
            self.engine.memcpy(ptr, full.data_ptr(), 8 * total, 1)
            self.engine.migrate_deal()
            self._host_allgather()
        else:
            self.engine.migrate()
        self.steps_since_migration = 0

    def particle_ids(self) -> np.ndarray:
        """index, in the arrays given to start(), of the particle at each position of this rank's get_state arrays"""
        return self.engine.particle_ids()

    def set_observables(self, on: bool) -> None:
        self.observables = bool(on)
        self.engine.set_observables(self.observables)

    # -- the reference's call sequence ------------------------------------------------
    def start(self, r: np.ndarray, v: np.ndarray):
        """Every rank passes the same global r[3, N], v[3, N] (md_simulation_program.f90:221-236).
        -> (epot, d_epot, dd_epot) of the t = 0 force evaluation."""
        self.engine.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        self.exchange_positions()      # every rank re-orders its own block at set_state: share the new order
        if self.migrate_every > 0:
            self.migrate()             # the first deal is by position too
        self._finish(False)
        parts = self._gather_partials(self.engine.read_partials(1))
        e, _k, d, dd = self._combine(parts)
        return e[0], d[0], dd[0]

    def enqueue_steps(self, nsteps: int, sampled: bool = False) -> None:
        """nsteps x {drift+kick1 | all-gather | forces+kick2}; no host synchronisation.
        sampled: potential-energy sums on the last step only (ljmd_enqueue_steps_sampled / ljmd_set_observables)."""
        if self.migrate_every > 0 and self.steps_since_migration >= self.migrate_every:
            self.migrate()
        self.steps_since_migration += nsteps
        for s in range(nsteps):
            if sampled:
                self.engine.set_observables(s == nsteps - 1)
            self.engine.step_begin()
            self.exchange_positions()
            self._finish(True)
        if sampled:
            self.engine.set_observables(self.observables)     # what the caller had chosen, as the C paths do

    def collect(self, nsteps: int):
        """-> (epot, ekin, d_epot, dd_epot) arrays of the last nsteps enqueued steps."""
        return self._combine(self._gather_partials(self.engine.read_partials(nsteps)))

    def run(self, nsteps: int):
        self.enqueue_steps(nsteps)
        return self.collect(nsteps)
