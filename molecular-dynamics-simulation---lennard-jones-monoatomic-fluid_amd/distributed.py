"""Multi-GPU run of the hot path: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Sharding: rank g integrates the contiguous particle range
[g*S, (g+1)*S), S = N / G, and evaluates those rows of the pair matrix against ALL N
positions.  Per MD step there is exactly one data-path exchange: an in-place all-gather
of the freshly drifted position shard (3*S doubles per rank) into the shard-blocked
exchange buffer, issued on the engine's own HIP stream right after the drift/half-kick
kernel, so that it is ordered behind it and ahead of the pair kernel without any host
synchronisation.  The per-step scalar partial sums (S12, S6, Kx, Ky, Kz per rank) stay on
the device; they are gathered once per `run()` call and combined on the host in fixed rank
order, so the result does not depend on collective internals.

The engine object is injected: the product uses `ljmd_amd.Engine` (HIP); the CPU tests
inject an oracle-backed stand-in to exercise this file's sharding/exchange logic with gloo.
"""
from __future__ import annotations

import contextlib

import numpy as np

PARTIAL_STRIDE = 8


class _DeviceArray:
    """Zero-copy view of library-owned HBM for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}


def hip_exchange_tensors(engine, device_index: int):
    """-> (full exchange buffer as a 1-D torch tensor on the GPU, view of this rank's block)."""
    import torch
    ptr, total, off, cnt = engine.exchange_buffer()
    full = torch.as_tensor(_DeviceArray(ptr, total), device=torch.device("cuda", device_index))
    return full, full[off:off + cnt]


def hip_stream_context(engine, device_index: int):
    """Makes the engine's HIP stream torch's current stream so collectives are ordered on it."""
    import torch
    ext = torch.cuda.ExternalStream(engine.stream(), device=torch.device("cuda", device_index))
    return lambda: torch.cuda.stream(ext)


class ShardedSimulation:
    def __init__(self, engine, full_tensor, own_view, rank: int, world: int, group=None,
                 stream_context=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.full, self.own = full_tensor, own_view
        self.rank, self.world, self.group = rank, world, group
        self.stream_context = stream_context or contextlib.nullcontext

    # -- the one data-path collective ---------------------------------------------
    def exchange_positions(self) -> None:
        if self.world == 1:
            return
        with self.stream_context():
            self.dist.all_gather_into_tensor(self.full, self.own, group=self.group)

    def _gather_partials(self, mine: np.ndarray) -> np.ndarray:
        """mine: [k, PARTIAL_STRIDE] -> [world, k, PARTIAL_STRIDE] (host, tiny)."""
        import torch
        if self.world == 1:
            return mine[None]
        t = torch.from_numpy(np.ascontiguousarray(mine))
        if self.dist.get_backend(self.group) == "nccl":
            t = t.to(self.full.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy()

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
