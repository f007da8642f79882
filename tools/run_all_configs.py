#!/usr/bin/env python3
"""One pass over BASELINE.json's configs[1..4] on one MI355X (configs[0], the reference's own N = 108 run, is the
drop-in test tests/test_gpu_dropin.py).  Prints one JSON line per config; the summary is committed under profiles/.
  2: N = 4096, 10 000 steps, energy conservation vs the reference's raw scalars (tests/golden/traj_n4096_*.npz)
  3: N = 262 144 fp64 (the bench workload)
  4: N = 1 048 576 -- on ONE GPU here (the 8-GPU run is the driver's): 1-rank step time, and rank 0 of an 8-rank
     decomposition with the collectives left out (what one of eight GPUs would compute per step)
  5: N = 262 144 mixed precision
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ljmd_amd  # noqa: E402,F401
from ljmd_amd import Engine, synthetic, _lib, init_params  # noqa: E402
from ljmd_amd.physics import observables  # noqa: E402


def timed_steps(eng, steps):
    eng.synchronize()
    t0 = time.perf_counter()
    out = eng.verlet_steps(steps)
    return out, time.perf_counter() - t0


def config2():
    gold = ROOT / "tests" / "golden" / "traj_n4096_10000.npz"
    g = np.load(gold if gold.exists() else ROOT / "tests" / "golden" / "traj_n4096_200.npz")
    p, r, v = synthetic.make_config(4096)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = eng.compute_forces()
        k0 = eng.kinetic_energy()
        (e, k, d, _dd), secs = timed_steps(eng, 10000)
        vel = np.stack(eng.get_state(("v",))["v"])
    etot = np.concatenate([[e0[0] + k0], e + k])
    ref = g["scalars"]
    ref_etot = ref[:, 0] + ref[:, 1]
    m = min(len(ref_etot), len(etot))
    rel = np.abs(etot[:m] - ref_etot[:m]) / np.abs(ref_etot[:m])
    out = {"config": 2, "n": 4096, "steps": 10000, "steps_per_s": 10000 / secs,
           "etot_t0": float(etot[0]), "etot_t0_reference": float(ref_etot[0]),
           "max_rel_dev_etot_first_200_steps_vs_reference": float(rel[:201].max()),
           "etot_peak_to_peak_rel": float((etot.max() - etot.min()) / abs(etot.mean())),
           "momentum_per_particle_final": float(np.abs(vel.sum(axis=1)).max() / 4096)}
    if m > 5000:
        T = 2.0 * k / (3.0 * 4096)
        Tr = 2.0 * ref[1:, 1] / (3.0 * 4096)
        P = np.array([observables(p, a, b, c)[2] for a, b, c in zip(e[5000:], k[5000:], d[5000:])])
        Pr = np.array([observables(p, a, b, c)[2] for a, b, c in ref[5001:, :3]])
        out.update({"reference_steps_available": int(m - 1),
                    "first_step_with_etot_rel_dev_above_1e-10": int(np.argmax(rel > 1e-10)) if (rel > 1e-10).any() else None,
                    "mean_etot_rel_dev_steps_5000_10000": float(abs(etot[5000:].mean() - ref_etot[5000:].mean()) / abs(ref_etot.mean())),
                    "etot_peak_to_peak_rel_reference": float((ref_etot.max() - ref_etot.min()) / abs(ref_etot.mean())),
                    "T_mean_last_5000": float(T[5000:].mean()), "T_mean_last_5000_reference": float(Tr[5000:].mean()),
                    "P_mean_last_5000": float(P.mean()), "P_mean_last_5000_reference": float(Pr.mean())})
    return out


def config35(mode):
    p, r, v = synthetic.make_config(262144)
    with Engine(p, precision_mode=_lib.PRECISION_FP32_FORCE if mode == "mixed" else _lib.PRECISION_FP64) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.verlet_steps(3)
        (e, k, _d, _dd), secs = timed_steps(eng, 20)
    return {"config": 3 if mode == "fp64" else 5, "n": 262144, "mode": mode, "steps": 20, "steps_per_s": 20 / secs,
            "pair_interactions_per_s": 262144 * 262143 / 2 * 20 / secs, "etot_last": float(e[-1] + k[-1])}


def config4():
    n = 1048576
    p, r, v = synthetic.make_config(n)
    out = {"config": 4, "n": n}
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = eng.compute_forces()[0]
        eng.verlet_steps(1)
        (e, k, _d, _dd), secs = timed_steps(eng, 5)
        out.update({"one_gpu_steps_per_s": 5 / secs, "one_gpu_ms_per_step": 1e3 * secs / 5,
                    "one_gpu_pair_interactions_per_s": n * (n - 1) / 2 * 5 / secs, "epot_t0": e0})
    # rank 0 of 8: exchange buffer filled by hand, force exchange marked external (tools/probe_rank.py)
    G = 8
    engines = [Engine(p, rank=g, n_ranks=G) for g in range(G)]
    for en in engines:
        en.force_buffers(True)
        en.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    for en in engines:
        en.synchronize()
    for src in engines:
        sp, _tot, off, cnt = src.exchange_buffer()
        for dst in engines:
            if dst is not src:
                dst.memcpy(dst.exchange_buffer()[0] + 8 * off, sp + 8 * off, 8 * cnt, 3)
    eng = engines[0]
    eng.forces_partial()
    eng.step_begin(); eng.step_forces(); eng.step_finish()
    eng.read_partials(2)
    eng.profile_enable(True)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.step_begin(); eng.step_forces(); eng.step_finish()
    eng.synchronize()
    wall = (time.perf_counter() - t0) / 5
    prof = eng.profile_read()
    out.update({"rank0_of_8_ms_per_step_without_collectives": 1e3 * wall, "rank0_of_8_pair_kernel_ms": prof["pair_ms"],
                "exchange_bytes_per_rank_per_step": 2 * 3 * 8 * (n // G)})
    for en in engines:
        en.close()
    # the whole sharded system through the single-process multi-device handle (what md_simulation_gpu does with
    # LJMD_GPUS=8), all eight ranks on this one card: peer-copy exchange, rank-ordered force sum
    with Engine(p, devices=[0] * G) as multi:
        multi.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e8 = multi.compute_forces()[0]
        (e, k, _d, _dd), secs = timed_steps(multi, 2)
        vel = np.stack(multi.get_state(("v",))["v"])
    out.update({"multi_handle_8_ranks_one_card_epot_t0": e8, "multi_handle_epot_rel_diff_vs_one_rank": abs(e8 - out["epot_t0"]) / abs(out["epot_t0"]),
                "multi_handle_8_ranks_one_card_ms_per_step": 1e3 * secs / 2,
                "multi_handle_momentum_per_particle": float(np.abs(vel.sum(axis=1)).max() / n)})
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "3", "4", "5"]
    for c in which:
        rec = {"2": config2, "3": lambda: config35("fp64"), "4": config4, "5": lambda: config35("mixed")}[c]()
        print(json.dumps(rec), flush=True)
