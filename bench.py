#!/usr/bin/env python3
"""bench.py -- MD steps/s of the LJ force + velocity-Verlet hot path at N = 262 144, fp64
(BASELINE.json metric), on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full MD step on synthetic input already resident in HBM: drift + wrap +
half-kick + unwrapped update, all-pairs LJ forces/energy/virial, second half-kick, kinetic
energy.  For N > 1 the SAME 262 144-particle system is sharded by particle rows (strong
scaling) with one all-gather of positions and one reduce-scatter of partial accelerations per step.
Rank 0 prints ONE JSON line.

Multi-GPU launch ladder (N > 1).  The process the driver (or a user) starts never touches torch or the GPU: it is a
WATCHDOG that runs each attempt as a fresh child process with a deadline, kills the child's whole process tree when
the deadline passes or the child fails, and moves to the next rung:
  ranks-rccl   one process per GPU (torch.distributed.run), RCCL all-gather + reduce-scatter issued inside libljmd.so
  multi-rccl   ONE process driving all N devices (ljmd_create_multi), ncclCommInitAll + grouped collectives
  multi-copy   the same with peer-to-peer hipMemcpyAsync pulls + a rank-ordered sum (no RCCL at all)
  multi-host   the same staged through pinned host memory (neither RCCL nor peer access)
The line of the first rung that delivers one is printed, with `config.launch_mode`, `config.exchange` and the outcome
of every rung tried in `config.ladder`.  Under the driver's own `torch.distributed.run` every rank process is such a
watchdog for its own rank's child in the first rung; only rank 0's goes on to the single-process rungs.
Environment: LJMD_BENCH_LADDER (comma list of rungs, default all four), LJMD_BENCH_DEADLINES (seconds per rung, default
190,100,80,80: with the kill grace of every rung inside a 600 s driver limit), LJMD_BENCH_SHARE_DEVICE=1 (rehearsal: every rank on device 0),
LJMD_BENCH_EXCHANGE=host (rehearsal: the one-process-per-GPU form with its host-staged gloo exchange, nothing else).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

N_PARTICLES = 262144
FLOP_PER_UNORDERED_PAIR = 33.8   # reference's Newton-3 loop: 21 outside + 0.493 * 26 inside the cutoff
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64 (256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz / 2)
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E (MI355X_MICROARCH.md)
PKG = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"

LADDER = ("ranks-rccl", "multi-rccl", "multi-copy", "multi-host")
DEADLINES = (190.0, 100.0, 80.0, 80.0)   # + per timed-out rung <= 12 s of kill grace, 5 s reader join, 5 s pause: < 540 s in all
EXCHANGE_LABEL = {
    "ranks-rccl": "RCCL all-gather + reduce-scatter inside libljmd.so, one process per GPU",
    "ranks-host": "HOST-STAGED (requested: LJMD_BENCH_EXCHANGE=host): PCIe + gloo, one process per GPU",
    "multi-rccl": "RCCL all-gather + reduce-scatter (ncclCommInitAll, grouped calls), one process for all GPUs",
    "multi-copy": "peer-to-peer hipMemcpyAsync pulls + rank-ordered sum (no RCCL), one process for all GPUs",
    "multi-host": "pulls staged through pinned host memory (no RCCL, no peer access), one process for all GPUs",
}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (N = 1 only)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s: float = 20.0) -> dict:
    """Times the CPU path on this box's host cores on a bounded sample of the same workload.
    kind = "reference": oracle/_ref/ref_harness (the real reference, 1 core, compiled -O2) when
    the prebuilt binary travelled with the snapshot; otherwise kind = "port": the C oracle."""
    from oracle import oracle as O
    pairs_full = N_PARTICLES * (N_PARTICLES - 1) / 2.0
    n_s = 32768                                   # ~5.4e8 pairs: 10-20 s on one core
    if O.ref_available():
        secs, pps = O.ref_bench(n_s, 1)
        kind, cores = "reference", 1
        what = f"real reference (oracle/_ref, amdflang -O2), 1 force call at N={n_s}"
    else:
        from ljmd_amd import synthetic
        p, r, _ = synthetic.make_config(n_s)
        po = O.derive_params(p.n, p.box_length, p.dt, p.rc)
        t0 = time.perf_counter()
        O.compute_forces(po, r[0].copy(), r[1].copy(), r[2].copy())
        secs = time.perf_counter() - t0
        pps = (n_s * (n_s - 1) / 2.0) / secs
        kind, cores = "port", 1
        what = f"C oracle (gcc -O2, no FMA), 1 force call at N={n_s}"
    return {"value": pps / pairs_full, "unit": "steps/s", "cores": cores, "kind": kind,
            "pairs_per_s": pps, "seconds": secs, "host_cores_available": os.cpu_count(),
            "sample": what + f", same rho/rc/jitter recipe; steps/s = pairs/s / {pairs_full:.4e} pairs per step "
                             f"(the O(N) integrator is <0.1% of a CPU step)"}


def host_cpu_share() -> int:
    """Cores this process may actually use: min(visible CPUs, cgroup CPU quota)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline_all_cores(rows: int = 16384) -> dict:
    """the C oracle's OpenMP full-matrix form (every ordered pair, same per-pair arithmetic as
    the reference loop, no Newton-3 scatter so that rows are independent) on this box's CPU share, on a bounded
    sample of the bench workload: the first `rows` rows of the N = 262144 pair matrix against all columns."""
    cores = host_cpu_share()
    os.environ["OMP_NUM_THREADS"] = str(cores)          # before liboracle.so (libgomp) is loaded
    from oracle import oracle as O
    from ljmd_amd import synthetic
    p, r, _ = synthetic.make_config(N_PARTICLES)
    po = O.derive_params(p.n, p.box_length, p.dt, p.rc)
    x, y, z = (np.ascontiguousarray(a) for a in r)
    O.rows_raw(po, 0, 256, x, y, z)                      # thread-pool warm-up
    t0 = time.perf_counter()
    O.rows_raw(po, 0, rows, x, y, z)
    secs = time.perf_counter() - t0
    step_s = secs * N_PARTICLES / rows
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "ordered_pairs_per_s": rows * (N_PARTICLES - 1) / secs, "seconds": secs,
            "sample": f"C oracle, OpenMP full-matrix rows (gcc -O2, no FMA), rows 0..{rows - 1} of the N={N_PARTICLES} "
                      f"bench configuration against all columns on {cores} threads (cgroup CPU share of "
                      f"{os.cpu_count()} visible CPUs); one force evaluation = {N_PARTICLES // rows}x the sample"}


# ---------------------------------------------------------------------------------------------------------------------
# The watchdog and its ladder (N > 1).  Nothing in this section imports torch or touches a GPU.
# ---------------------------------------------------------------------------------------------------------------------
def _log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


class DescendantTracker:
    """Remembers every process `proc` has started while it is still alive.  torch.distributed.run puts its workers into
    sessions of their own, and once the launcher has been reaped /proc no longer links them to it (they are re-parented
    to init): a listing taken after `proc.wait()` finds nothing.  psutil.Process objects carry the creation time, so a
    recycled PID is never signalled."""

    def __init__(self, proc, period_s: float = 0.5):
        import threading
        self.proc, self.period_s, self.seen = proc, period_s, {}
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._poll, daemon=True)
        self._th.start()

    def _scan(self) -> None:
        import psutil
        try:
            for c in psutil.Process(self.proc.pid).children(recursive=True):
                self.seen.setdefault((c.pid, c.create_time()), c)
        except psutil.Error:
            pass

    def _poll(self) -> None:
        while not self._stop.is_set():
            self._scan()
            self._stop.wait(self.period_s)

    def stop(self) -> list:
        """-> the descendants seen so far (alive or not)"""
        self._stop.set()
        self._th.join(timeout=2.0)
        return list(self.seen.values())


def kill_process_tree(proc, grace_s: float = 6.0, known=()) -> None:
    """Ends `proc`, every process it has started (listed from /proc now) and the `known` descendants a
    DescendantTracker saw earlier -- by exact PID + creation time, never by pattern; SIGTERM first, SIGKILL for
    whatever is left after `grace_s`."""
    import signal
    import psutil
    victims = {}
    try:
        root = psutil.Process(proc.pid)
        for v in root.children(recursive=True) + [root]:
            victims[(v.pid, v.create_time())] = v
    except psutil.Error:
        pass
    for v in known:
        try:
            if v.is_running():                      # (False for a recycled PID: the creation time differs)
                victims.setdefault((v.pid, v.create_time()), v)
        except psutil.Error:
            pass
    victims = list(victims.values())
    for v in victims:
        try:
            v.send_signal(signal.SIGTERM)
        except psutil.Error:
            pass
    _gone, alive = psutil.wait_procs(victims, timeout=grace_s)
    for v in alive:
        try:
            v.kill()
        except psutil.Error:
            pass
    psutil.wait_procs(alive, timeout=grace_s)
    try:
        proc.wait(timeout=grace_s)
    except Exception:  # noqa: BLE001 - a zombie we cannot reap is not worth losing the next rung for
        pass


def run_attempt(cmd, env, deadline_s: float, relay: bool = True) -> dict:
    """One rung: `cmd` as a fresh child (own session) with a deadline.  Its stdout is read line by line: the JSON
    bench line is kept, everything else goes to our stderr.  -> {"rc", "line", "seconds", "outcome"};
    outcome: "ok" | "timeout" | "exit <rc>" | "no line"."""
    import subprocess
    import threading
    t0 = time.monotonic()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1,
                            start_new_session=True)
    box = {"line": None}

    def reader():
        for ln in proc.stdout:
            if ln.startswith("{") and '"metric"' in ln:
                box["line"] = ln.strip()
            elif relay:
                print(ln, end="", file=sys.stderr, flush=True)

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    tracker = DescendantTracker(proc)
    try:
        rc = proc.wait(timeout=deadline_s)
        outcome = "ok" if rc == 0 else f"exit {rc}"
        known = tracker.stop()
    except subprocess.TimeoutExpired:
        _log(f"deadline of {deadline_s:.0f} s passed: ending the attempt's process tree (pid {proc.pid})")
        known = tracker.stop()
        kill_process_tree(proc, known=known)
        rc, outcome = -9, "timeout"
    # a rank that died may leave siblings behind (torch.distributed.run ends them, but make sure): the child has been
    # reaped by now, so only the descendants the tracker saw while it lived can still be found
    if rc != 0 and outcome != "timeout":
        kill_process_tree(proc, grace_s=3.0, known=known)
    th.join(timeout=5.0)
    if rc == 0 and box["line"] is None:
        outcome = "no line"
    return {"rc": rc, "line": box["line"], "seconds": time.monotonic() - t0, "outcome": outcome}


def ladder_plan(n_ranks: int, launched: bool, argv) -> list:
    """-> [(mode, cmd, env)] of the rungs this process will try.  Test hook: LJMD_BENCH_ATTEMPT_CMD_<MODE> (JSON list)
    replaces the command of a rung (tests/test_bench_ladder.py uses it to stand in for a hung or a healthy child)."""
    import socket
    me = str(Path(__file__).resolve())
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "1")
    if os.environ.get("LJMD_BENCH_EXCHANGE", "") == "host":
        modes = ["ranks-host"]
    else:
        modes = [m.strip() for m in os.environ.get("LJMD_BENCH_LADDER", ",".join(LADDER)).split(",") if m.strip()]
    bad = [m for m in modes if m not in LADDER and m != "ranks-host"]
    if bad:
        raise SystemExit(f"LJMD_BENCH_LADDER: unknown rung(s) {bad}; known: {LADDER}")
    launcher_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE",
                     "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ROLE_NAME")
    plan = []
    for mode in modes:
        env = dict(base, LJMD_BENCH_MODE=mode)
        if mode.startswith("ranks"):
            env["LJMD_BENCH_ROLE"] = "ranks"
            if launched:
                cmd = [sys.executable, me] + list(argv)        # this rank's child, in the launcher's environment
            else:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    port = s.getsockname()[1]
                env = {k: v for k, v in env.items() if k not in launcher_keys and not k.startswith("TORCHELASTIC_")}
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
                       "--master-addr", "127.0.0.1", "--master-port", str(port), me] + list(argv)
        else:
            env = {k: v for k, v in env.items() if k not in launcher_keys and not k.startswith("TORCHELASTIC_")}
            env["LJMD_BENCH_ROLE"] = "multi"
            env["LJMD_MULTI_EXCHANGE"] = mode.split("-", 1)[1]
            cmd = [sys.executable, me] + list(argv)
        hook = os.environ.get("LJMD_BENCH_ATTEMPT_CMD_" + mode.upper().replace("-", "_"))
        if hook:
            cmd = json.loads(hook)
        plan.append((mode, cmd, env))
    return plan


def run_ladder(n_ranks: int, argv) -> int:
    """The watchdog.  Returns the exit code; prints at most one JSON line (rank 0 / the plain parent only)."""
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ and int(os.environ["WORLD_SIZE"]) > 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    plan = ladder_plan(n_ranks, launched, argv)
    try:
        deadlines = [float(x) for x in os.environ.get("LJMD_BENCH_DEADLINES", "").split(",") if x.strip()]
    except ValueError:
        deadlines = []
    deadlines = deadlines or list(DEADLINES)
    history = []
    for i, (mode, cmd, env) in enumerate(plan):
        if launched and rank != 0 and not mode.startswith("ranks"):
            return 0                                       # the single-process rungs are rank 0's watchdog's business
        deadline = deadlines[min(i, len(deadlines) - 1)]
        if rank == 0:
            _log(f"rung {i + 1}/{len(plan)} '{mode}' (deadline {deadline:.0f} s): {' '.join(cmd)}")
        res = run_attempt(cmd, env, deadline, relay=True)
        history.append({"mode": mode, "outcome": res["outcome"], "seconds": round(res["seconds"], 2)})
        if res["rc"] == 0 and res["line"] is not None:
            if rank == 0:
                try:
                    line = json.loads(res["line"])
                    line.setdefault("config", {})["ladder"] = history
                    line["config"].setdefault("launch_mode", mode)
                    print(json.dumps(line), flush=True)
                except ValueError:
                    print(res["line"], flush=True)
            return 0
        if launched and rank != 0:
            if res["rc"] == 0:
                continue                                   # (a rank > 0 prints no line: success is the exit code)
            # this rank's child failed or hung: rank 0's watchdog sees the same and carries on alone
            return 0 if any(not m.startswith("ranks") for m, _c, _e in plan[i + 1:]) else 1
        if rank == 0:
            _log(f"rung '{mode}' gave no line ({res['outcome']} after {res['seconds']:.1f} s)")
        if res["outcome"] == "timeout":
            time.sleep(float(os.environ.get("LJMD_BENCH_GRACE_S", "5")))   # the other ranks' watchdogs end theirs
    if rank == 0:
        _log("no rung delivered a bench line: " + json.dumps(history))
    return 1 if rank == 0 or not launched else 0


# ---------------------------------------------------------------------------------------------------------------------
# The measurement (runs in the child processes of the ladder, or directly for N = 1)
# ---------------------------------------------------------------------------------------------------------------------
class MultiHandleSim:
    """ShardedSimulation's interface over ONE multi-device handle (ljmd_create_multi): the library runs the ranks and
    both exchanges itself, global arrays in and out."""

    def __init__(self, engine):
        self.engine = engine

    def start(self, r, v):
        self.engine.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        return self.engine.compute_forces()

    def enqueue_steps(self, nsteps: int, sampled: bool = False) -> None:
        self.engine.enqueue_steps(nsteps, sampled=sampled)

    def collect(self, nsteps: int):
        return self.engine.collect_steps(nsteps)

    def run(self, nsteps: int):
        self.enqueue_steps(nsteps)
        return self.collect(nsteps)


def committed_parity_summary():
    """newest profiles/rNN_mixed_precision_parity_vs_oracle.json (written by tests/test_gpu_parity.py on the GPU box)"""
    hits = sorted((ROOT / "profiles").glob("r[0-9][0-9]_mixed_precision_parity_vs_oracle.json"))
    if not hits:
        return None
    doc = dict(json.loads(hits[-1].read_text()), source=f"profiles/{hits[-1].name}")
    # measured with these kernel sources and the default r_split?  Otherwise the figures belong to another build:
    # quoted as found under a key that says so (as load_pmc does for the counter summaries)
    split_now = float(os.environ.get("LJMD_FP32_SPLIT", "5.0"))
    if doc.get("kernel_source_sha16") != kernel_source_sha16() or float(doc.get("r_split_sigma", 5.0)) != split_now:
        return {"stale_committed_parity": doc,
                "note": "measured with other kernel sources or another r_split than this run's: quoted as found"}
    return doc


def mixed_precision_leg(args, p, r, v, etot_fp64, barrier_extra=None) -> dict:
    """K timed steps of the mixed-precision engine from the start the fp64 headline used (warm-up included), on device 0."""
    from ljmd_amd import Engine
    from ljmd_amd import _lib as _abi
    with Engine(p, device=0, precision_mode=_abi.PRECISION_FP32_FORCE) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        if args.warmup > 0:
            eng.verlet_steps(args.warmup)
        eng.profile_enable(True)
        eng.synchronize()
        if barrier_extra:
            barrier_extra.cuda.synchronize()
        t0 = time.perf_counter()
        eng.enqueue_steps(args.steps)
        eng.synchronize()
        if barrier_extra:
            barrier_extra.cuda.synchronize()
        el = time.perf_counter() - t0
        prof = eng.profile_read_rank(0)
        e, k, _d, _dd = eng.collect_steps(args.steps)
    et = e + k
    out = {"metric": f"md_steps_per_sec_n{p.n}_mixed_fp32_far_pairs", "value": args.steps / el, "unit": "steps/s",
           "ms_per_step": 1e3 * el / args.steps, "pair_kernels_ms_avg": prof["pair_ms"],
           "pair_kernels_ms_min": prof["pair_ms_min"], "pair_kernels_ms_median": prof["pair_ms_median"],
           "dtype": "f32 far pairs (box distance > 5 sigma) / f64 near pairs, accumulation and integrator",
           "etot_max_rel_dev_from_fp64_series": float(np.max(np.abs(et - etot_fp64) / np.abs(etot_fp64))),
           "note": "same K steps from the same start as the fp64 headline; not the headline"}
    parity = committed_parity_summary()
    if parity:
        out["parity_vs_oracle"] = parity
    return out


def kernel_source_sha16() -> str:
    """identifies the kernel sources a committed PMC summary was collected with (tools/pmc_summary.py stores the same)"""
    hsh = hashlib.sha256()
    for name in ("ljmd_kernels.hip", "ljmd_internal.h"):
        hsh.update((PKG / "csrc" / name).read_bytes())
    return hsh.hexdigest()[:16]


def measure(args) -> None:
    role = os.environ.get("LJMD_BENCH_ROLE", "ranks")
    mode = os.environ.get("LJMD_BENCH_MODE", "single" if args.gpus == 1 else "ranks-rccl")
    rank = int(os.environ.get("RANK", "0")) if role == "ranks" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if role == "ranks" else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if role == "ranks" else 1
    n_ranks = args.gpus                                  # ranks of the decomposition, whichever way they are driven
    if role == "ranks" and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launcher and --gpus disagree")

    # dmabuf IPC: the only mode the host driver of this pool supports for cross-process device memory (RCCL)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    # HIP runtime bookkeeping: torch ships its own libamdhip64 / librccl / libhsa-runtime64 (same SONAMEs as
    # the system ROCm ones).  torch is imported FIRST here, so when libljmd.so is loaded next the dynamic
    # loader resolves its NEEDED libamdhip64.so.7 / librccl.so.1 to the copies torch already mapped: ONE HIP
    # runtime and ONE RCCL per process, shared by torch and the engine (checked via /proc/self/maps), and
    # `torch.cuda.synchronize()` below really covers the engine's kernels.  The engine's own
    # hipStreamSynchronize + hipDeviceSynchronize (Engine.synchronize) is called beside it anyway.
    torch_gpu = torch.cuda.is_available()
    share = os.environ.get("LJMD_BENCH_SHARE_DEVICE", "0") == "1"
    if torch_gpu:
        torch.cuda.set_device(0 if share else local_rank % max(1, torch.cuda.device_count()))
    import ljmd_amd  # noqa: F401
    from ljmd_amd import Engine, synthetic, distributed
    from ljmd_amd import _lib as _abi

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (RCCL-id broadcast, barrier, partial-record gather): CPU tensors over gloo;
        # the position all-gather is RCCL over xGMI, issued inside libljmd.so on the engine's stream
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n = args.n
    p, r, v = synthetic.make_config(n)
    precision = _abi.PRECISION_FP32_FORCE if args.mode == "mixed" else _abi.PRECISION_FP64
    ndev = max(1, _abi.load().ljmd_device_count())
    if role == "multi":
        # ONE process, all devices.  Rehearsal on a box with fewer GPUs (LJMD_BENCH_SHARE_DEVICE=1): every rank on
        # device 0 -- RCCL refuses that, so the multi-rccl rung fails there and the ladder moves on
        devices = [0] * n_ranks if share else list(range(n_ranks))
        if not share and n_ranks > ndev:
            raise SystemExit(f"--gpus {n_ranks} but only {ndev} device(s) visible")
        eng = Engine(p, precision_mode=precision, devices=devices)
        want = os.environ.get("LJMD_MULTI_EXCHANGE", "rccl")
        if want == "rccl" and eng.comm_size() != n_ranks:
            raise SystemExit(f"multi-rccl rung: the handle has no RCCL communicator over {n_ranks} ranks")
        sim = MultiHandleSim(eng)
        exchange_key = "multi-" + want
    else:
        # a launcher may hand every rank its own single visible device (HIP_VISIBLE_DEVICES): then it is device 0
        device = 0 if share else (local_rank if local_rank < ndev else local_rank % ndev)
        eng = Engine(p, device=device, rank=rank, n_ranks=world, precision_mode=precision)
        if world == 1:
            exchange_key = "none"
        elif os.environ.get("LJMD_BENCH_EXCHANGE", "") == "host":
            exchange_key = "ranks-host"
        elif distributed.try_bootstrap_rccl(eng, rank, world):
            exchange_key = "ranks-rccl"
        else:
            # no silent fallback inside a rung: the watchdog goes down the ladder and the line says which rung ran
            raise SystemExit("ranks-rccl rung: the RCCL communicator could not be initialised on every rank")
        sim = distributed.ShardedSimulation(eng, rank, world, exchange="host" if exchange_key == "ranks-host" else "rccl")

    def barrier():
        eng.synchronize()                    # hipStreamSynchronize + hipDeviceSynchronize on the engine's device(s)
        if torch_gpu:
            if role == "multi" and not share:
                for d in range(n_ranks):
                    torch.cuda.synchronize(d)
            else:
                torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            eng.synchronize()

    def read_profiles():
        """-> per-rank interval averages, rank order (every rank of the decomposition, however it is driven)"""
        if role == "multi":
            return [eng.profile_read_rank(g) for g in range(n_ranks)]
        mine = eng.profile_read_rank(rank)
        if dist is None:
            return [mine]
        box = [None] * world
        dist.all_gather_object(box, mine)
        return box

    def timed(nsteps, sampled=False):
        eng.profile_enable(True)
        barrier()
        t0 = time.perf_counter()
        sim.enqueue_steps(nsteps, sampled=sampled)
        barrier()
        el = time.perf_counter() - t0
        profs = read_profiles()
        eng.profile_enable(False)
        return el, profs, sim.collect(nsteps)

    sim.start(r, v)
    if args.warmup > 0:
        sim.run(args.warmup)
    kernel_name = eng.pair_kernel_name()
    elapsed, profs, (epot, ekin, d_epot, dd_epot) = timed(args.steps)
    slow = max(range(len(profs)), key=lambda g: profs[g]["pair_ms"])     # the slowest rank bounds the step
    prof = profs[slow]
    force_ms, launches = prof["pair_ms"], prof["launches"]

    # production rate: the same K steps timed again after the lattice has melted (>= 300 steps from the start).
    # The tiles of a liquid are looser than those of the jittered lattice, so the tile mask keeps more pairs;
    # reported beside the headline, never as `value`.
    liquid = None
    if not args.no_liquid and n >= 4096:
        done = args.warmup + args.steps
        while done < 300:
            k = min(100, 300 - done)
            sim.run(k)
            done += k
        el_liq, profs_liq, _sc = timed(args.steps)
        slow_l = max(profs_liq, key=lambda q: q["pair_ms"])
        liquid = (el_liq, slow_l["pair_ms"], done, slow_l.get("pair_ms_min"), slow_l.get("pair_ms_median"))

    # the production loop's rate: the reference reads epot / d_epot / dd_epot only every output_interval steps
    # (md_simulation_program.f90:361; 100 in its input file), so the driver runs the steps in between with the
    # forces-only pair kernel (ljmd_enqueue_steps_sampled).  The same K steps again as ONE sampled segment, in the
    # state the previous leg left; the trajectory is bit-identical.  Reported beside the headline, never as `value`.
    sampled = None
    if not args.no_liquid and n >= 4096:
        el_s, profs_s, sc_s = timed(args.steps, sampled=True)
        sampled = (el_s, max(q["pair_ms"] for q in profs_s), bool(np.isfinite(sc_s[0][-1])),
                   int(np.count_nonzero(np.isnan(sc_s[0]))))

    if dist is not None:
        t = torch.tensor([elapsed, liquid[0] if liquid else 0.0, sampled[0] if sampled else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if liquid:
            liquid = (float(t[1].item()),) + liquid[1:]
        if sampled:
            sampled = (float(t[2].item()),) + sampled[1:]

    if rank == 0:
        steps_per_s = args.steps / elapsed
        pairs = n * (n - 1) / 2.0
        etot = epot + ekin
        # roofline of the dominant kernel (pair forces): fp64 vector-ALU bound:
        # algorithmic flops per launch = reference's per-unordered-pair flop count x the pairs
        # this rank's launch covers (its rows x all columns / 2)
        flops_per_launch = FLOP_PER_UNORDERED_PAIR * pairs / n_ranks
        achieved = flops_per_launch / (force_ms * 1e-3) / 1e12 if force_ms > 0 else 0.0
        # per-rank figures of the timed steps, whichever way the ranks were driven (one rank: the same keys)
        multi_cfg = {
            "per_rank_ms": [round(q["pair_ms"] + q["geometry_ms"] + q["drift_ms"] + q["reduce_ms"], 4) for q in profs],
            "position_exchange_ms": [round(q.get("pos_exchange_ms", 0.0), 4) for q in profs],
            "force_exchange_ms": [round(q.get("force_exchange_ms", 0.0), 4) for q in profs],
            "migrations": int(eng.migrations()),
        }
        if n_ranks > 1:
            multi_cfg.update({
                "launch_mode": mode,
                "rccl_ranks_seen": eng.comm_size(),
                "ownership": f"x-slabs of exactly n/G particles dealt by position on the devices (ljmd_migrate); migrations so far: {eng.migrations()}",
                "pair_kernel_ms_per_rank": [round(q["pair_ms"], 4) for q in profs],
                "position_exchange_ms_per_rank": [round(q["pos_exchange_ms"], 4) for q in profs],
                "force_exchange_ms_per_rank": [round(q["force_exchange_ms"], 4) for q in profs],
                "collective_ms_per_step": round(max(q["pos_exchange_ms"] + q["force_exchange_ms"] for q in profs), 4),
                "collective_ms_note": "HIP events on the stream that carries each exchange, from the moment the rank could "
                                      "start it: waiting for the slowest rank is included",
                "force_exchange": os.environ.get("LJMD_FORCE_EXCHANGE", "reducescatter"),
                "overlap_exchange": os.environ.get("LJMD_OVERLAP_EXCHANGE", "1"),
            })
        line = {
            "metric": (f"md_steps_per_sec_n{n}_fp64" if args.mode == "fp64" else f"md_steps_per_sec_n{n}_mixed_fp32_far_pairs"),
            "value": steps_per_s, "unit": "steps/s", "n_gpus": n_ranks, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.mode == "fp64" else "f32 far pairs / f64 near pairs, accumulation and integrator", "data": "synthetic",
            "config": {"workload": f"N={n} LJ fluid, rho=0.8, rc=0.49L, dt=0.005, simple-cubic+5% jitter, T=1.0; "
                                   f"all-pairs force + velocity-Verlet step (BASELINE configs[2])",
                       "particles": n, "sharding": (f"rows/{n_ranks}" + (" (REHEARSAL: all ranks on one device)" if share else "")) if n_ranks > 1 else "single GPU",
                       "exchange": EXCHANGE_LABEL.get(exchange_key, "none"),
                       "unordered_pairs_per_step": pairs, **multi_cfg},
            "pair_interactions_per_sec": pairs * steps_per_s,
            "roofline": {"bound": "fp64-valu", "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                         "kernel": kernel_name, "kernel_ms_avg": force_ms,
                         # what tells a 1-2 % kernel change from box-to-box scatter: the shortest and the median launch
                         # of the K timed ones (same HIP events), and the fraction the shortest one reaches
                         "kernel_ms_min": prof.get("pair_ms_min"), "kernel_ms_median": prof.get("pair_ms_median"),
                         "frac_of_shortest_launch": (flops_per_launch / (prof["pair_ms_min"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
                                                     if prof.get("pair_ms_min", 0.0) > 0 else None),
                         "geometry_prepass_ms_avg": prof["geometry_ms"], "launches_timed": launches,
                         "flop_per_unordered_pair": FLOP_PER_UNORDERED_PAIR,
                         "hbm_algorithmic_GBps": (48.0 * n / n_ranks) / (force_ms * 1e-3) / 1e9 if force_ms > 0 else 0.0,
                         # the other kernels of a step (HIP-event intervals, averages per step): K1 drift/wrap/kick/
                         # unwrapped update incl. the amortised re-sort; slab reduction + second kick + finalize
                         "drift_kick_resort_ms_avg": prof["drift_ms"], "reduce_kick_finalize_ms_avg": prof["reduce_ms"],
                         "drift_kick_algorithmic_bytes": 168.0 * n / n_ranks},
            **({"steps_per_s_liquid": args.steps / liquid[0],
                "liquid": {"equilibration_steps": liquid[2], "ms_per_step": 1e3 * liquid[0] / args.steps,
                           "pair_kernel_ms_avg": liquid[1], "pair_kernel_ms_min": liquid[3], "pair_kernel_ms_median": liquid[4],
                           "note": "same K steps timed again in the equilibrated liquid; not the headline"}}
               if liquid else {}),
            **({"steps_per_s_sampled_segment": args.steps / sampled[0],
                "sampled_segment": {"ms_per_step": 1e3 * sampled[0] / args.steps, "pair_kernel_ms_avg": sampled[1],
                                    "observables_on_last_step_finite": sampled[2], "steps_without_energy_sums": sampled[3],
                                    "note": "same K steps as one ljmd_enqueue_steps_sampled segment (liquid state): energy sums "
                                            "only on the step the reference samples (output_interval); not the headline"}}
               if sampled else {}),
            "energy_check": {"etot_first": float(etot[0]), "etot_last": float(etot[-1]),
                             "rel_drift": float(abs(etot[-1] - etot[0]) / abs(etot[0]))},
        }
        if args.mode == "mixed":
            # two pair kernels (fp64 near, fp32 far) share the timed interval: no single-peak roofline applies
            line["roofline"].update({"bound": "fp64-valu (near pairs) + fp32-valu (far pairs)", "frac": None,
                                     "note": "achieved = reference-algorithm fp64 flop / time of both pair kernels"})
            parity = committed_parity_summary()
            if parity:
                # config 5's accuracy beside its rate: measured deviations of ONE force call of this workload from
                # the CPU oracle over all ordered pairs (tests/test_gpu_parity.py writes the summary)
                line["parity_vs_oracle"] = parity
        # the HBM-bound kernel of the step, K1 (drift + wrap + half-kick + unwrapped update): 168 N algorithmic
        # bytes per launch / its shortest HIP-event interval (= K1 alone; steps that re-sort are longer)
        if prof.get("drift_ms_min", 0.0) > 0.0 and n_ranks == 1:
            k1_gbps = (168.0 * n) / (prof["drift_ms_min"] * 1e-3) / 1e9
            line["roofline_hbm_kernel"] = {"kernel": "drift_kick_kernel", "bound": "hbm", "achieved": k1_gbps,
                                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k1_gbps / HBM_PEAK_GBPS,
                                           "kernel_ms_min": prof["drift_ms_min"],
                                           "algorithmic_bytes": 168.0 * n}

        # Counter evidence from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE need
        # separate passes and cannot be read from inside the run).  A summary is quoted as evidence for THIS run only
        # when it was collected with the kernel sources this run was built from (kernel_source_sha16); otherwise its raw
        # numbers appear under `stale_committed_profile` and nothing is derived from them.
        def committed(stem):
            """newest committed profile of that name (profiles/rNN_<stem>)"""
            hits = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{stem}"))
            return hits[-1] if hits else None

        def headline_instances(kernels, name):
            """template instances of the pair kernel in a committed PMC summary, without the forces-only ones
            (<..., false>: they run only in the sampled-segment leg)"""
            hits = {key: val for key, val in kernels.items() if key.startswith("ljmdk::" + name + "<")}
            full = [val for key, val in hits.items() if not key.rstrip().endswith(("false>", "0>", "(bool)0>"))]
            return full or list(hits.values())

        def load_pmc(stem):
            f = committed(stem)
            if not (n_ranks == 1 and n == N_PARTICLES and args.mode == "fp64" and f):
                return None, None, False
            doc = json.loads(f.read_text())
            return doc, f, doc.get("kernel_source_sha16") == kernel_source_sha16()

        stale = {}
        doc, f, fresh = load_pmc("final_pmc_hbm_traffic.json")
        if doc:
            hits = headline_instances(doc["kernels"], kernel_name)
            k = max(hits, key=lambda val: val["hbm_bytes_per_launch"]) if hits else {}   # the template instance that ran
            if k and fresh:
                line["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = f"profiles/{f.name}"
            elif k:
                stale["hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
                stale["traffic_source"] = f"profiles/{f.name}"
        # VALU issue evidence (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles))
        doc, f, fresh = load_pmc("final_pmc_valu.json")
        if doc:
            hits = headline_instances(doc["kernels"], kernel_name)
            k = max(hits, key=lambda val: val.get("SQ_INSTS_VALU", 0.0)) if hits else {}
            if k.get("valu_issue_frac") and fresh:
                line["roofline"].update({"valu_issue_frac": k["valu_issue_frac"],
                                         "valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"],
                                         "valu_source": f"profiles/{f.name}"})
                if k.get("clock_ghz_observed"):
                    # the shader clock the kernel really ran at in the committed counter pass (kernel cycles / the
                    # dispatch's own duration there); `peak` assumes 2.4 GHz
                    clk = k["clock_ghz_observed"]
                    line["roofline"].update({"clock_ghz_observed": clk,
                                             "frac_at_observed_clock": achieved / (FP64_VALU_PEAK_TFLOPS * clk / 2.4)})
            elif k.get("valu_issue_frac"):
                stale.update({"valu_issue_frac": k["valu_issue_frac"], "valu_source": f"profiles/{f.name}"})
        # executed fp64 instruction mix of the pair kernel: the flop the kernel really executes per launch (exec-masked
        # lanes included) over the LIVE kernel time, beside the algorithmic figure above; LDS bank conflicts
        doc, f, fresh = load_pmc("final_pmc_instruction_mix.json")
        if doc and force_ms > 0:
            hits = headline_instances(doc["kernels"], kernel_name)
            k = max(hits, key=lambda val: val.get("executed_fp64_flop", 0.0)) if hits else {}
            if k.get("executed_fp64_flop") and fresh:
                ex = k["executed_fp64_flop"] / (force_ms * 1e-3) / 1e12
                line["roofline"].update({"executed_fp64_tflops": ex, "executed_fp64_frac_of_peak": ex / FP64_VALU_PEAK_TFLOPS,
                                         "fp64_wave_instructions": {x: k.get("SQ_INSTS_VALU_" + x) for x in
                                                                    ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")},
                                         "lds_wave_instructions": k.get("SQ_INSTS_LDS"),
                                         "lds_bank_conflict_cycles": k.get("SQ_LDS_BANK_CONFLICT"),
                                         "mix_source": f"profiles/{f.name}"})
            elif k.get("executed_fp64_flop"):
                stale.update({"executed_fp64_flop_per_launch": k["executed_fp64_flop"], "mix_source": f"profiles/{f.name}"})
        if stale:
            stale["note"] = ("collected with other kernel sources than this run's (kernel_source_sha16 differs or is "
                             "absent): quoted as found, nothing derived against this run's timing")
            line["roofline"]["stale_committed_profile"] = stale
        # K1 by the profiler's clock (a HIP-event interval around a 9 us kernel is mostly event overhead)
        stats = committed("final_kernel_stats.csv")
        if n_ranks == 1 and n == N_PARTICLES and stats and "roofline_hbm_kernel" in line:
            import csv
            for row in csv.DictReader(open(stats)):
                if row["Name"].startswith("ljmdk::drift_kick_kernel"):
                    ms = float(row["AverageNs"]) * 1e-6
                    gbps = 168.0 * n / (ms * 1e-3) / 1e9
                    line["roofline_hbm_kernel"].update({"rocprof_kernel_ms_avg": ms, "rocprof_achieved": gbps,
                                                        "rocprof_frac": gbps / HBM_PEAK_GBPS,
                                                        "rocprof_source": f"profiles/{stats.name}"})
                    break
        # BASELINE config 5 beside the fp64 headline, in the same run: the same workload with the fp32 far-pair kernel
        # (LJMD_PRECISION_FP32_FORCE), K steps timed the same way from the same start, and what it costs in accuracy --
        # the energy series against this run's fp64 one, and the committed one-force-call parity against the CPU oracle
        if n_ranks == 1 and args.mode == "fp64" and not args.no_liquid and n >= 16384:
            try:
                line["config5_mixed_precision"] = mixed_precision_leg(args, p, r, v, etot, barrier_extra=torch_gpu and torch)
            except Exception as exc:  # noqa: BLE001 - a secondary figure, never a reason to lose the bench line
                line["config5_mixed_precision"] = {"value": None, "error": str(exc)}
        if n_ranks == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # the baseline is a reported number, never a reason to lose the bench line
                line["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {exc}"}
            try:
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            except Exception as exc:
                line["cpu_baseline_all_cores"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port",
                                                  "sample": f"failed: {exc}"}
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=N_PARTICLES, help="override the particle count (parity/debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-liquid", action="store_true",
                    help="skip the second, untimed-for-`value` measurement in the equilibrated liquid (300 extra steps)")
    ap.add_argument("--mode", choices=("fp64", "mixed"), default="fp64",
                    help="mixed = BASELINE config 5 (fp32 far pairs, fp64 near pairs + integrator); the headline metric is fp64")
    return ap.parse_args(argv)


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "LJMD_BENCH_ROLE" not in os.environ:
        raise SystemExit(run_ladder(args.gpus, sys.argv[1:]))      # the watchdog: nothing below runs in this process
    measure(args)


if __name__ == "__main__":
    main()
