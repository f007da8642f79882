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


class ShardedSimulation:
    def __init__(self, engine, rank: int, world: int, group=None):
        self.engine = engine
        self.rank, self.world, self.group = rank, world, group
        if world > 1:
            import torch.distributed as dist
            self.dist = dist

    # -- the one data-path collective ---------------------------------------------
    def exchange_positions(self) -> None:
        if self.world > 1:
            self.engine.allgather_positions()

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

    # -- the reference's call sequence ------------------------------------------------
    def start(self, r: np.ndarray, v: np.ndarray):
        """Every rank passes the same global r[3, N], v[3, N] (md_simulation_program.f90:221-236).
        -> (epot, d_epot, dd_epot) of the t = 0 force evaluation."""
        self.engine.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        self.exchange_positions()      # every rank re-orders its own block at set_state: share the new order
        self.engine.forces_partial()
        parts = self._gather_partials(self.engine.read_partials(1))
        e, _k, d, dd = self._combine(parts)
        return e[0], d[0], dd[0]

    def enqueue_steps(self, nsteps: int) -> None:
        """nsteps x {drift+kick1 | all-gather | forces+kick2}; no host synchronisation."""
        for _ in range(nsteps):
            self.engine.step_begin()
            self.exchange_positions()
            self.engine.step_finish()

    def collect(self, nsteps: int):
        """-> (epot, ekin, d_epot, dd_epot) arrays of the last nsteps enqueued steps."""
        return self._combine(self._gather_partials(self.engine.read_partials(nsteps)))

    def run(self, nsteps: int):
        self.enqueue_steps(nsteps)
        return self.collect(nsteps)
