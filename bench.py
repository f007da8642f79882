#!/usr/bin/env python3
"""bench.py -- MD steps/s of the LJ force + velocity-Verlet hot path at N = 262 144, fp64
(BASELINE.json metric), on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A plain `python bench.py --gpus N` (N > 1, no launcher environment) starts the N ranks itself: the parent
process -- before it touches torch or the GPU -- runs the torch.distributed.run command above as a CHILD
process, relays rank 0's JSON line and exits with the child's return code.

One "step" = one full MD step on synthetic input already resident in HBM: drift + wrap +
half-kick + unwrapped update, all-pairs LJ forces/energy/virial, second half-kick, kinetic
energy.  For N > 1 the SAME 262 144-particle system is sharded by particle rows (strong
scaling) with one RCCL all-gather of positions per step (issued inside libljmd.so; torch.distributed/gloo is the
control plane: RCCL-id bootstrap, barrier, scalar gather).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
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


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run as a child
    process (this parent has made no torch / HIP call), relay its output -- the JSON line to stdout, everything
    else to stderr -- and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] no launcher environment: starting", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    got_line = False
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            print(ln, end="", flush=True)
            got_line = True
        else:
            print(ln, end="", file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("[bench] the ranks exited cleanly but printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def main() -> None:
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and (not launched or os.environ["WORLD_SIZE"] == "1"):
        raise SystemExit(self_launch(args.gpus))         # parent: nothing below runs here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
    if torch_gpu:
        torch.cuda.set_device(0 if os.environ.get("LJMD_BENCH_SHARE_DEVICE", "0") == "1"
                              else local_rank % max(1, torch.cuda.device_count()))
    import ljmd_amd  # noqa: F401
    from ljmd_amd import Engine, synthetic, distributed

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (RCCL-id broadcast, barrier, partial-record gather): CPU tensors over gloo;
        # the position all-gather is RCCL over xGMI, issued inside libljmd.so on the engine's stream
        dist.init_process_group("gloo", rank=rank, world_size=world)

    n = args.n
    p, r, v = synthetic.make_config(n)
    # rehearsal knobs for a box with fewer GPUs than ranks (RCCL refuses two ranks on one device):
    # LJMD_BENCH_SHARE_DEVICE=1 puts every rank on device 0, LJMD_BENCH_EXCHANGE=host forces the
    # host-staged exchange.  Never set by the driver; a line produced with them says so in `config`.
    share = os.environ.get("LJMD_BENCH_SHARE_DEVICE", "0") == "1"
    from ljmd_amd import _lib as _abi
    # a launcher may hand every rank its own single visible device (HIP_VISIBLE_DEVICES): then it is device 0
    ndev = max(1, _abi.load().ljmd_device_count())
    device = 0 if share else (local_rank if local_rank < ndev else local_rank % ndev)
    eng = Engine(p, device=device, rank=rank, n_ranks=world,
                 precision_mode=_abi.PRECISION_FP32_FORCE if args.mode == "mixed" else _abi.PRECISION_FP64)
    if os.environ.get("LJMD_BENCH_EXCHANGE", "") == "host" or share:
        exchange = "host"
    else:
        exchange = "rccl" if distributed.try_bootstrap_rccl(eng, rank, world) else "host"
    sim = distributed.ShardedSimulation(eng, rank, world, exchange=exchange)

    def barrier():
        eng.synchronize()                    # hipStreamSynchronize + hipDeviceSynchronize on the engine's device
        if torch_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            eng.synchronize()

    e0, d0, dd0 = sim.start(r, v)
    if args.warmup > 0:
        sim.run(args.warmup)
    eng.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    sim.enqueue_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    kernel_name = eng.pair_kernel_name()
    force_ms, launches = prof["pair_ms"], prof["launches"]
    integ_ms = prof["drift_ms"] + prof["reduce_ms"]
    eng.profile_enable(False)
    epot, ekin, d_epot, dd_epot = sim.collect(args.steps)

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
        eng.profile_enable(True)
        barrier()
        t1 = time.perf_counter()
        sim.enqueue_steps(args.steps)
        barrier()
        el_liq = time.perf_counter() - t1
        prof_liq = eng.profile_read()
        eng.profile_enable(False)
        sim.collect(args.steps)
        liquid = (el_liq, prof_liq["pair_ms"], done)

    # the production loop's rate: the reference reads epot / d_epot / dd_epot only every output_interval steps
    # (md_simulation_program.f90:361; 100 in its input file), so the driver runs the steps in between with the
    # forces-only pair kernel (ljmd_enqueue_steps_sampled).  The same K steps again as ONE sampled segment, in the
    # state the previous leg left; the trajectory is bit-identical.  Reported beside the headline, never as `value`.
    sampled = None
    if not args.no_liquid and n >= 4096:
        eng.profile_enable(True)
        barrier()
        t2 = time.perf_counter()
        sim.enqueue_steps(args.steps, sampled=True)
        barrier()
        el_s = time.perf_counter() - t2
        prof_s = eng.profile_read()
        eng.profile_enable(False)
        sc_s = sim.collect(args.steps)
        sampled = (el_s, prof_s["pair_ms"], bool(np.isfinite(sc_s[0][-1])), int(np.count_nonzero(np.isnan(sc_s[0]))))

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
        flops_per_launch = FLOP_PER_UNORDERED_PAIR * pairs / world
        achieved = flops_per_launch / (force_ms * 1e-3) / 1e12 if force_ms > 0 else 0.0
        line = {
            "metric": (f"md_steps_per_sec_n{n}_fp64" if args.mode == "fp64" else f"md_steps_per_sec_n{n}_mixed_fp32_far_pairs"),
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.mode == "fp64" else "f32 far pairs / f64 near pairs, accumulation and integrator", "data": "synthetic",
            "config": {"workload": f"N={n} LJ fluid, rho=0.8, rc=0.49L, dt=0.005, simple-cubic+5% jitter, T=1.0; "
                                   f"all-pairs force + velocity-Verlet step (BASELINE configs[2])",
                       "particles": n, "sharding": (f"rows/{world}" + (" (REHEARSAL: all ranks on one device)" if share else "")) if world > 1 else "single GPU",
                       "exchange": ("RCCL all-gather + reduce-scatter inside libljmd.so" if exchange == "rccl"
                                    else "HOST-STAGED FALLBACK (RCCL init failed): PCIe + gloo") if world > 1 else "none",
                       "unordered_pairs_per_step": pairs,
                       **({"rccl_ranks_seen": eng.comm_size()} if world > 1 else {}),
                       **({"force_exchange": os.environ.get("LJMD_FORCE_EXCHANGE", "reducescatter"),
                           "overlap_exchange": os.environ.get("LJMD_OVERLAP_EXCHANGE", "1")} if world > 1 else {})},
            "pair_interactions_per_sec": pairs * steps_per_s,
            "roofline": {"bound": "fp64-valu", "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                         "kernel": kernel_name, "kernel_ms_avg": force_ms, "geometry_prepass_ms_avg": prof["geometry_ms"], "launches_timed": launches,
                         "flop_per_unordered_pair": FLOP_PER_UNORDERED_PAIR,
                         "hbm_algorithmic_GBps": (48.0 * n / world) / (force_ms * 1e-3) / 1e9 if force_ms > 0 else 0.0,
                         # the other kernels of a step (HIP-event intervals, averages per step): K1 drift/wrap/kick/
                         # unwrapped update incl. the amortised re-sort; slab reduction + second kick + finalize
                         "drift_kick_resort_ms_avg": prof["drift_ms"], "reduce_kick_finalize_ms_avg": prof["reduce_ms"],
                         "drift_kick_algorithmic_bytes": 168.0 * n / world},
            **({"steps_per_s_liquid": args.steps / liquid[0],
                "liquid": {"equilibration_steps": liquid[2], "ms_per_step": 1e3 * liquid[0] / args.steps,
                           "pair_kernel_ms_avg": liquid[1],
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
        # the HBM-bound kernel of the step, K1 (drift + wrap + half-kick + unwrapped update): 168 N algorithmic
        # bytes per launch / its shortest HIP-event interval (= K1 alone; steps that re-sort are longer)
        if prof.get("drift_ms_min", 0.0) > 0.0:
            k1_gbps = (168.0 * n / world) / (prof["drift_ms_min"] * 1e-3) / 1e9
            line["roofline_hbm_kernel"] = {"kernel": "drift_kick_kernel", "bound": "hbm", "achieved": k1_gbps,
                                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k1_gbps / HBM_PEAK_GBPS,
                                           "kernel_ms_min": prof["drift_ms_min"],
                                           "algorithmic_bytes": 168.0 * n / world}
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
        # command (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be read from inside the run)
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

        pmc = committed("final_pmc_hbm_traffic.json")
        if world == 1 and n == N_PARTICLES and args.mode == "fp64" and pmc:
            kernels = json.loads(pmc.read_text())["kernels"]
            hits = headline_instances(kernels, kernel_name)
            k = max(hits, key=lambda val: val["hbm_bytes_per_launch"]) if hits else {}   # the template instance that ran
            if k:
                line["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = f"profiles/{pmc.name}"
        # VALU issue evidence of the same command (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles)), PMC pass
        valu = committed("final_pmc_valu.json")
        if world == 1 and n == N_PARTICLES and args.mode == "fp64" and valu:
            kernels = json.loads(valu.read_text())["kernels"]
            hits = headline_instances(kernels, kernel_name)
            k = max(hits, key=lambda val: val.get("SQ_INSTS_VALU", 0.0)) if hits else {}
            if k.get("valu_issue_frac"):
                line["roofline"].update({"valu_issue_frac": k["valu_issue_frac"],
                                         "valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"],
                                         "valu_source": f"profiles/{valu.name}"})
        # executed fp64 instruction mix of the pair kernel (PMC passes 5 / 6 of tools/collect_profiles.sh): the flop the
        # kernel really executes per launch (exec-masked lanes included) over the LIVE kernel time, beside the
        # algorithmic figure above; LDS bank conflicts of the parked column tiles
        mixf = committed("final_pmc_instruction_mix.json")
        if world == 1 and n == N_PARTICLES and args.mode == "fp64" and mixf and force_ms > 0:
            kernels = json.loads(mixf.read_text())["kernels"]
            hits = headline_instances(kernels, kernel_name)
            k = max(hits, key=lambda val: val.get("executed_fp64_flop", 0.0)) if hits else {}
            if k.get("executed_fp64_flop"):
                ex = k["executed_fp64_flop"] / (force_ms * 1e-3) / 1e12
                line["roofline"].update({"executed_fp64_tflops": ex, "executed_fp64_frac_of_peak": ex / FP64_VALU_PEAK_TFLOPS,
                                         "fp64_wave_instructions": {x: k.get("SQ_INSTS_VALU_" + x) for x in
                                                                    ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")},
                                         "lds_wave_instructions": k.get("SQ_INSTS_LDS"),
                                         "lds_bank_conflict_cycles": k.get("SQ_LDS_BANK_CONFLICT"),
                                         "mix_source": f"profiles/{mixf.name}"})
        # K1 by the profiler's clock (a HIP-event interval around a 9 us kernel is mostly event overhead)
        stats = committed("final_kernel_stats.csv")
        if world == 1 and n == N_PARTICLES and stats and "roofline_hbm_kernel" in line:
            import csv
            for row in csv.DictReader(open(stats)):
                if row["Name"].startswith("ljmdk::drift_kick_kernel"):
                    ms = float(row["AverageNs"]) * 1e-6
                    gbps = 168.0 * n / (ms * 1e-3) / 1e9
                    line["roofline_hbm_kernel"].update({"rocprof_kernel_ms_avg": ms, "rocprof_achieved": gbps,
                                                        "rocprof_frac": gbps / HBM_PEAK_GBPS,
                                                        "rocprof_source": f"profiles/{stats.name}"})
                    break
        if world == 1 and not args.no_cpu_baseline:
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


if __name__ == "__main__":
    main()
