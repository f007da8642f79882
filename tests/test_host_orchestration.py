"""CPU-only: the HOST orchestration of libljmd.so -- staging buffers, slot permutation bookkeeping, scalar ring,
snapshots, the stateless fast path, and above all the single-process multi-GPU exchange (grouped RCCL collectives
over ncclCommInitAll communicators; peer-copy exchange) -- executed under AddressSanitizer + UBSan against a fake
HIP/RCCL runtime (tests/fakehip: host memory, inert streams, no-op kernels, REAL collectives between the
communicators of one process).  Every extent the host code hands to hipMemcpyAsync / ncclAllGather /
ncclReduceScatter / ncclSend+Recv is bounds-checked; kernel results are meaningless here and not looked at.
The multi-rank RCCL path cannot run on the one-GPU box, so this is where its buffer arithmetic is exercised."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

PKG = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
ASAN_LIB = PKG / "csrc" / "obj" / "libljmd_asan.so"
FAKE = ROOT / "tests" / "fakehip" / "libfakehip.so"

SCRIPT = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import ljmd_amd
from ljmd_amd import Engine, _lib, synthetic, physics, init_state, distributed
fake = C.CDLL(%(fake)r)
for f in ("fakehip_kernel_launches", "fakehip_copies", "fakehip_collectives"):
    getattr(fake, f).restype = C.c_long
lib = _lib.load()
assert lib.ljmd_device_count() == 8

def drive(eng, r, v, steps, ranks=1):
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    eng.compute_forces()
    eng.kinetic_energy()
    eng.verlet_steps(steps)                       # crosses re-sorts
    eng.enqueue_steps(7); eng.snapshot_begin(); eng.enqueue_steps(5)
    snap = eng.snapshot_end()
    eng.collect_steps(5)
    st = eng.get_state()
    eng.set_accel(*st["a"]); eng.set_unwrapped(*st["ru"])
    eng.verlet_steps(3)
    eng.enqueue_steps(6, sampled=True); eng.collect_steps(6)      # forces-only steps, sums on the last one
    eng.advance(4)
    # the production driver's order around a sample: begin, next segment, end -- with a migration due in between
    eng.snapshot_begin(); eng.migrate(); eng.enqueue_steps(4); eng.snapshot_end(); eng.collect_steps(4)
    eng.profile_enable(True); eng.enqueue_steps(3); eng.collect_steps(3)
    prof = [eng.profile_read_rank(g) for g in range(ranks)]
    assert all(q["launches"] == 3 for q in prof), prof
    eng.profile_enable(False)
    return st

# single engines: gather kernel sizes, Newton-3 with 1 / 2 / 4 tiles per row group, padded shards, mixed precision
for n, env in ((108, {}), (500, {}), (3000, {"LJMD_N3_MIN_N": "1"}), (4096, {}), (16384, {}),
               (20000, {"LJMD_N3_ROW_TILES": "4", "LJMD_N3_WG_WAVES": "4"}), (16384, {"mode": "1"}),
               (20000, {"LJMD_N3_WG_WAVES": "2", "mode": "1"})):     # mixed precision beside LDS-combining workgroups
    mode = int(env.pop("mode", "0"))
    os.environ.update(env)
    p, r, v = synthetic.make_config(n, seed=3)
    with Engine(p, precision_mode=mode) as eng:
        st = drive(eng, r, v, 45)
        assert all(a.shape == (n,) for k in ("r", "ru", "v", "a") for a in st[k])
    for k in env:
        del os.environ[k]

# single-process multi-device: RCCL (distinct devices) and peer-copy exchange (forced, and duplicate devices)
c0 = fake.fakehip_collectives()
for n, devices, env in ((16384, [0, 1], {}), (16384, [0, 1, 2, 3], {}), (32768, list(range(8)), {}),
                        (3000, [0, 1, 2], {"LJMD_N3_MIN_N": "1"}), (4096, [2, 5], {"LJMD_N3": "0"}),
                        (16384, [0, 1, 2, 3], {"LJMD_MULTI_EXCHANGE": "copy"}), (16384, [0, 0, 0, 0], {}),
                        (16384, [0, 1, 2, 3], {"LJMD_MULTI_MIGRATE_EVERY": "5"}),     # ownership migration between the segments
                        (16384, [0, 1, 2, 3], {"LJMD_MULTI_EXCHANGE": "host", "LJMD_MULTI_MIGRATE_EVERY": "7"}),   # pinned-host staging
                        (4096, [1, 3], {"LJMD_N3": "0", "LJMD_MULTI_EXCHANGE": "host"}),
                        (16384, [0, 1, 2, 3], {"LJMD_OVERLAP_EXCHANGE": "0"}),         # exchanges on the engine streams
                        (16384, [0, 1, 2, 3], {"LJMD_MULTI_THREADS": "0"}),            # one host thread for all ranks, grouped RCCL
                        (16384, [0, 0, 0, 0], {"LJMD_MULTI_THREADS": "0", "LJMD_MULTI_EXCHANGE": "host"}),
                        (3072, [0, 1, 2], {"LJMD_N3": "0"}),                           # gather kernels, RCCL, one thread per rank
                        (12288, [0, 1, 2], {"LJMD_MULTI_EXCHANGE": "copy", "LJMD_N3_MIN_N": "1", "LJMD_MULTI_MIGRATE_EVERY": "3"}),
                        (24576, [0, 1, 2], {"LJMD_N3_ROW_TILES": "4", "LJMD_N3_WG_WAVES": "2", "LJMD_N3_MIN_N": "1"})):
    os.environ.update(env)
    p, r, v = synthetic.make_config(n, seed=5)
    with Engine(p, devices=devices) as eng:
        rccl = len(set(devices)) == len(devices) and env.get("LJMD_MULTI_EXCHANGE") not in ("copy", "host")
        assert eng.comm_size() == (len(devices) if rccl else 0)
        st = drive(eng, r, v, 25, ranks=len(devices))
        assert eng.migrations() >= 2                    # at set_state and the explicit one, plus the periodic ones
        assert all(a.shape == (n,) for k in ("r", "ru", "v", "a") for a in st[k])
    for k in env:
        del os.environ[k]
assert fake.fakehip_collectives() > c0 + 100       # the grouped collectives really moved data

# one-process-per-GPU form on a 1-rank communicator, both force-exchange forms, overlapped and serial gather
for fx in ("reducescatter", "alltoall"):
    for ov in ("1", "0"):
        os.environ.update(LJMD_FORCE_COLLECTIVES="1", LJMD_FORCE_EXCHANGE=fx, LJMD_OVERLAP_EXCHANGE=ov, LJMD_N3_MIN_N="1")
        p, r, v = synthetic.make_config(8192, seed=7)
        with Engine(p) as eng:
            eng.comm_init(Engine.comm_unique_id())
            sim = distributed.ShardedSimulation(eng, 0, 1)
            sim.start(r, v)
            sim.run(25)
        for k in ("LJMD_FORCE_COLLECTIVES", "LJMD_FORCE_EXCHANGE", "LJMD_OVERLAP_EXCHANGE", "LJMD_N3_MIN_N"):
            del os.environ[k]

# sharded engines (split-phase API) with the caller doing the exchange through ljmd_memcpy
p, r, v = synthetic.make_config(8192, seed=9)
engines = [Engine(p, rank=g, n_ranks=4, device=g) for g in range(4)]
for e in engines:
    e.force_buffers(True)
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
for e in engines:
    e.step_forces()
for e in engines:
    e.forces_partial()
for _ in range(3):
    for e in engines:
        e.step_begin()
    for e in engines:
        e.step_forces()
    for e in engines:
        e.step_finish()
# ownership migration through the split-phase entry points, the caller moving the blocks
for e in engines:
    e.migrate_pack()
bufs = [e.migrate_buffer() for e in engines]
for d, e in enumerate(engines):
    for g in range(4):
        if g != d:
            ptr_g, total, off_g, cnt = bufs[g]
            e.memcpy(bufs[d][0] + 8 * off_g, ptr_g + 8 * off_g, 8 * cnt, 3)
for e in engines:
    e.migrate_deal()
    assert e.particle_ids().shape == (2048,) and e.migrations() == 1
for e in engines:
    e.read_partials(4)
    e.close()

# stateless entry points: strict path, resident fast path, fallback after the caller touched an array
p, r, v = synthetic.make_config(4096, seed=11)
st = init_state(p)
st.rx[:], st.ry[:], st.rz[:] = r
st.vx[:], st.vy[:], st.vz[:] = v
physics.compute_lj_potential_energy(p, st)
for k in range(6):
    if k == 4:
        st.vx[3] += 1e-3
    physics.verlet_step(p, st)
physics.stateless_reset()

# trajectory-analysis entry point
hist = (C.c_uint64 * 64)()
x = np.random.default_rng(1).uniform(0, 10, 3 * 500).reshape(3, 500)
dp = C.POINTER(C.c_double)
assert lib.ljmd_rdf_histogram(500, x[0].ctypes.data_as(dp), x[1].ctypes.data_as(dp), x[2].ctypes.data_as(dp), 10.0, 64,
                              4.0, hist) == 0

# time-origin averages (MSD / VACF)
from ljmd_amd import analysis
tr = np.random.default_rng(2).normal(size=(3, 7, 300))
assert analysis.time_origin_average_gpu(0, *tr, max_lag=4, origin_stride=2).shape == (5,)
assert analysis.time_origin_average_gpu(1, *tr).shape == (7,)

# a failure half-way through a batch poisons the handle; set_state revives it
os.environ["LJMD_INJECT_FAILURE_AT_STEP"] = "2"
p, r, v = synthetic.make_config(4096, seed=13)
with Engine(p) as eng:
    del os.environ["LJMD_INJECT_FAILURE_AT_STEP"]
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2]); eng.compute_forces()
    try:
        eng.verlet_steps(5); raise SystemExit("expected the injected failure")
    except ljmd_amd.LjmdError as e:
        assert "injected" in str(e)
    try:
        eng.verlet_steps(1); raise SystemExit("expected LJMD_ERR_STATE")
    except ljmd_amd.LjmdError as e:
        assert "poisoned" in str(e)
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2]); eng.compute_forces(); eng.verlet_steps(4)

print("launches", fake.fakehip_kernel_launches(), "copies", fake.fakehip_copies(), "collectives", fake.fakehip_collectives())
print("host orchestration under sanitizers: ok")
"""


def test_host_orchestration_under_sanitizers_with_fake_runtime():
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not ASAN_LIB.exists() or not rt:
        pytest.skip("sanitizer build absent: make -C .../csrc asan")
    if not FAKE.exists():
        subprocess.run(["make", "-C", str(FAKE.parent)], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=f"{rt[-1]} {FAKE}", LJMD_LIBRARY=str(ASAN_LIB), FAKEHIP_DEVICES="8",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=24")
    for k in [k for k in env if k.startswith("LJMD_") and k != "LJMD_LIBRARY"]:
        del env[k]
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": str(ROOT), "fake": str(FAKE)}], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "host orchestration under sanitizers: ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr
