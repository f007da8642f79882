"""-m gpu: the multi-rank device path on ONE GPU.

The driver's 8-GPU node is the only place where >1 rank can own a card, so the sharded
kernels (row shard vs all-N columns, shard-blocked exchange buffer, per-rank partial records)
are exercised here by running G engines with rank = 0..G-1 on device 0 and performing the
all-gather with device-to-device copies; the collective itself is covered by the gloo test
(tests/test_distributed_gloo.py) and by a 1-rank RCCL communicator below.
"""
import numpy as np
import pytest

import ljmd_amd
from ljmd_amd import Engine, distributed, init_params, synthetic

pytestmark = pytest.mark.gpu


def _hip():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")       # already loaded by libljmd.so: same runtime instance
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    return hip


def _emulated_allgather(engines):
    """hipMemcpy device-to-device of every rank's own block into every other rank's exchange buffer."""
    hip = _hip()
    for e in engines:
        e.synchronize()
    for src in engines:
        sp, _tot, off, cnt = src.exchange_buffer()
        for dst in engines:
            if dst is not src:
                dp = dst.exchange_buffer()[0]
                assert hip.hipMemcpy(dp + 8 * off, sp + 8 * off, 8 * cnt, 3) == 0      # device to device
    assert hip.hipDeviceSynchronize() == 0


def _emulated_reduce_scatter(engines):
    """What ncclReduceScatter(sum) does on a real node: frecv[g] = sum over ranks of fpart_rank[g]."""
    hip = _hip()
    G = len(engines)
    for e in engines:
        e.synchronize()
    parts = []
    for e in engines:
        fp, nfp, _fr, nfr = e.force_buffers(True)
        if nfr == 0:
            return                                       # gather kernels: nothing to exchange
        host = np.empty(nfp)
        assert hip.hipMemcpy(host.ctypes.data, fp, 8 * nfp, 2) == 0                    # device to host
        parts.append(host.reshape(G, -1))
    for g, e in enumerate(engines):
        tot = parts[0][g].copy()
        for src in range(1, G):
            tot += parts[src][g]
        fr = e.force_buffers(True)[2]
        assert hip.hipMemcpy(fr, tot.ctypes.data, 8 * tot.size, 1) == 0                # host to device
    assert hip.hipDeviceSynchronize() == 0


def _emulated_migration(engines):
    """ljmd_migrate with the caller as the collective: pack, every block of the migration buffer copied into every other
    rank's buffer device to device, deal, and the position exchange again."""
    hip = _hip()
    for e in engines:
        e.migrate_pack()
    for e in engines:
        e.synchronize()
    bufs = [e.migrate_buffer() for e in engines]
    for g, (sp, _tot, off, cnt) in enumerate(bufs):
        for d, dst in enumerate(engines):
            if d != g:
                assert hip.hipMemcpy(bufs[d][0] + 8 * off, sp + 8 * off, 8 * cnt, 3) == 0
    assert hip.hipDeviceSynchronize() == 0
    for e in engines:
        e.migrate_deal()
    _emulated_allgather(engines)


@pytest.mark.parametrize("n,G,n3,migrate", [(4096, 2, False, False), (4096, 4, False, True), (3000, 3, False, False),
                                            (4096, 2, True, True), (4096, 4, True, False), (3000, 3, True, True),
                                            (8192, 8, True, True), (8192, 8, True, False)])
def test_sharded_engines_match_single_engine(n, G, n3, migrate, monkeypatch):
    """migrate: the ownership migration of the one-process-per-GPU form (ljmd_migrate_pack / _deal around the caller's
    exchange) right after set_state and again after 8 steps -- the ranks then own slabs dealt by position instead of
    index ranges, identified by ljmd_particle_ids; same bounds against the single engine."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1" if n3 else "100000000")
    p, r, v = synthetic.make_config(n, seed=5)
    nsteps = 15
    with Engine(p) as one:
        one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = one.compute_forces()
        ref = np.stack(one.verlet_steps(nsteps), axis=1)
        ref_state = one.get_state()

    engines = [Engine(p, rank=g, n_ranks=G) for g in range(G)]
    try:
        for e in engines:
            e.force_buffers(True)                      # the test performs the force exchange
            e.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        _emulated_allgather(engines)                   # every rank re-ordered its own block
        if migrate:
            _emulated_migration(engines)
        for e in engines:
            e.step_forces()
        _emulated_reduce_scatter(engines)
        for e in engines:
            e.forces_partial()
        parts0 = np.stack([e.read_partials(1)[0] for e in engines])
        t0 = engines[0].combine_scalars(parts0)
        assert np.allclose([t0[0], t0[2], t0[3]], e0, rtol=1e-13, atol=0)
        for step in range(nsteps):
            if migrate and step == 8:
                _emulated_migration(engines)           # step records of the first 8 steps stay pending across it
            for e in engines:
                e.step_begin()
            _emulated_allgather(engines)
            for e in engines:
                e.step_forces()
            _emulated_reduce_scatter(engines)
            for e in engines:
                e.step_finish()
        parts = np.stack([e.read_partials(nsteps) for e in engines])          # [G, nsteps, 8]
        sc = np.array([engines[0].combine_scalars(np.ascontiguousarray(parts[:, s])) for s in range(nsteps)])
        assert np.max(np.abs(sc - ref) / np.abs(ref)) < 1e-11
        S = n // G
        owned = []
        for g, e in enumerate(engines):
            if migrate:
                # a migrated rank owns the SET particle_ids names: asking for its index range fails loudly
                with pytest.raises(ljmd_amd.LjmdError, match="ljmd_particle_ids"):
                    e.shard_range()
            else:
                assert e.shard_range() == (g * S, (g + 1) * S)
            ids = e.particle_ids()
            owned.append(ids)
            assert e.migrations() == (2 if migrate else 0)
            if not migrate:
                assert np.array_equal(ids, np.arange(g * S, (g + 1) * S))
            st = e.get_state()
            for key in ("r", "ru", "v", "a"):
                mine = np.stack(st[key])
                want = np.stack(ref_state[key])[:, ids]
                scale = max(np.abs(want).max(), 1.0)
                assert np.abs(mine - want).max() < 1e-9 * scale, (g, key)
        assert np.array_equal(np.sort(np.concatenate(owned)), np.arange(n))     # every particle owned exactly once
        if migrate:
            # a rank's block is compact: its extent along the first split axis is about half the box (G >= 2)
            x0 = np.stack(engines[0].get_state(("r",))["r"])[0]
            assert x0.max() - x0.min() < 0.62 * p.box_length
            # the GLOBAL arrays of set_state order keep working as input: a round trip of a and ru through ids
            glob = {key: np.empty((3, n)) for key in ("ru", "a")}
            for g, e in enumerate(engines):
                st = e.get_state(("ru", "a"))
                for key in glob:
                    glob[key][:, owned[g]] = np.stack(st[key])
            for g, e in enumerate(engines):
                before = e.get_state(("ru", "a"))
                e.set_unwrapped(*glob["ru"])
                e.set_accel(*glob["a"])
                after = e.get_state(("ru", "a"))
                for key in glob:
                    assert np.array_equal(np.stack(after[key]), np.stack(before[key])), (g, key)
        with pytest.raises(ljmd_amd.LjmdError):
            engines[0].allgather_positions()           # n_ranks > 1 without ljmd_comm_init
    finally:
        for e in engines:
            e.close()


def test_rccl_communicator_single_rank():
    """The in-library RCCL path with a 1-rank communicator (all this box can host): unique id,
    ncclCommInitRank, the in-place ncclAllGather on the engine's stream, teardown."""
    uid = Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    p, r, v = synthetic.make_config(4096, seed=8)
    with Engine(p) as eng, Engine(p) as ref:
        eng.comm_init(uid)
        with pytest.raises(ljmd_amd.LjmdError):
            eng.comm_init(uid)                          # already initialised
        sim = distributed.ShardedSimulation(eng, 0, 1)
        e0 = sim.start(r, v)
        # call the collective by hand on the 1-rank communicator: must be a no-op copy in place
        before = np.stack(eng.get_state(("r",))["r"])
        eng._ck(eng._lib.ljmd_allgather_positions(eng._h))
        eng.synchronize()
        assert np.array_equal(before, np.stack(eng.get_state(("r",))["r"]))
        sc = sim.run(5)
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert ref.compute_forces() == e0
        assert np.array_equal(np.stack(sc), np.stack(ref.verlet_steps(5)))


@pytest.mark.parametrize("overlap,exchange", [("1", "reducescatter"), ("0", "reducescatter"), ("1", "alltoall"), ("0", "alltoall")])
def test_rccl_collectives_really_issued_by_the_engine(monkeypatch, overlap, exchange):
    """LJMD_FORCE_COLLECTIVES=1: a 1-rank engine goes through the multi-rank code path for real --
    ncclAllGather (in place) between the position update and the pair kernel, ncclReduceScatter of the
    partial accelerations into the receive buffer the kick kernel reads -- both enqueued by the library
    with no host synchronisation.  overlap=1 (default): the all-gather runs on the communication stream
    concurrently with the velocity half-kick, fenced by two events (re-sort steps take the serial form);
    overlap=0: everything on the engine's stream.  RCCL refuses two ranks on one device,
    so the communicator has one rank; everything else (buffers, stream order, kernels) is the G > 1 path.
    The trajectory must equal the plain single-GPU one bit for bit."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    p, r, v = synthetic.make_config(16384, seed=3)
    with Engine(p) as ref:
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e_ref = ref.compute_forces()
        sc_ref = ref.verlet_steps(25)                    # crosses a re-sort
        st_ref = ref.get_state()
    monkeypatch.setenv("LJMD_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("LJMD_OVERLAP_EXCHANGE", overlap)
    # exchange=alltoall: ncclSend/ncclRecv of the partial-acceleration blocks inside one group + a local
    # rank-order sum, instead of ncclReduceScatter
    monkeypatch.setenv("LJMD_FORCE_EXCHANGE", exchange)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        with pytest.raises(ljmd_amd.LjmdError, match="ljmd_comm_init"):
            eng.forces_partial()                         # collectives demanded, no communicator yet
        eng.comm_init(Engine.comm_unique_id())
        eng.allgather_positions()
        eng.forces_partial()                             # pair kernel -> reduce -> ncclReduceScatter -> x24
        rec0 = eng.read_partials(1)
        e0 = eng.combine_scalars(rec0)
        assert (e0[0], e0[2], e0[3]) == e_ref
        for _ in range(25):
            eng.step_begin()
            eng.allgather_positions()
            eng.step_finish()
        recs = eng.read_partials(25)
        sc = [eng.combine_scalars(recs[k:k + 1]) for k in range(25)]
        st = eng.get_state()
    assert np.array_equal(np.array(sc).T, np.stack(sc_ref))
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(st[key]), np.stack(st_ref[key])), key


def test_bench_two_processes_host_staged_exchange(tmp_path):
    """bench.py end to end with TWO processes sharing this box's single GPU, started the way the driver may start
    it: a PLAIN `python bench.py --gpus 2` with no launcher environment -- bench.py then runs its ranks under
    torch.distributed.run as a child process and relays rank 0's JSON line.  The exchange is the host-staged
    safety net because RCCL refuses two ranks on one device.  Checks the multi-process control flow of the very
    script the driver launches, and that the 2-rank trajectory equals the 1-rank one."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LJMD_BENCH_SHARE_DEVICE="1", LJMD_BENCH_EXCHANGE="host")
    common = ["--steps", "4", "--warmup", "1", "--particles", "32768", "--no-cpu-baseline"]
    out2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"] + common, env=env,
                          capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    line2 = json.loads([ln for ln in out2.stdout.splitlines() if ln.startswith("{")][-1])
    out1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1"] + common, capture_output=True,
                          text=True, timeout=600)
    assert out1.returncode == 0, out1.stderr[-2000:]
    line1 = json.loads([ln for ln in out1.stdout.splitlines() if ln.startswith("{")][-1])
    assert len([ln for ln in out2.stdout.splitlines() if ln.strip()]) == 1          # stdout = the ONE JSON line
    assert line2["n_gpus"] == 2 and "HOST-STAGED" in line2["config"]["exchange"]
    assert line2["config"]["rccl_ranks_seen"] == 0                                   # no communicator in this rehearsal
    for key in ("per_rank_ms", "position_exchange_ms", "force_exchange_ms"):         # one entry per rank, in every rung's line
        assert len(line2["config"][key]) == 2, key
    assert all(ms > 0 for ms in line2["config"]["per_rank_ms"]) and line2["config"]["migrations"] >= 0
    assert len(line1["config"]["per_rank_ms"]) == 1 and line1["config"]["migrations"] == 0
    for key in ("etot_first", "etot_last"):
        a, b = line2["energy_check"][key], line1["energy_check"][key]
        assert abs(a - b) <= 1e-11 * abs(b), (key, a, b)
    assert line2["value"] > 0 and line2["roofline"]["kernel"] == "pair_n3_kernel"


# ---- ONE process, several ranks: ljmd_create_multi -----------------------------------------------------------

@pytest.mark.parametrize("n,G", [(16384, 2), (16384, 4), (8192, 8), (3000, 3)])
def test_multi_device_handle_emulated_ranks(n, G, monkeypatch):
    """ljmd_create_multi with the SAME device listed G times: G rank engines on this box's one card, the library's
    own copy exchange (peer pulls behind events + rank-ordered sum; RCCL refuses two ranks on one device) -- the
    whole single-process multi-device control flow of the thin Fortran driver (LJMD_GPUS), with global arrays in
    and out.  Against the single engine: same bounds as the emulated-ranks test above; against a re-run: bitwise."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    p, r, v = synthetic.make_config(n, seed=5)
    nsteps = 25                                            # crosses a re-sort (every 20 steps)
    with Engine(p) as one:
        one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e0 = one.compute_forces()
        k0 = one.kinetic_energy()
        ref = np.stack(one.verlet_steps(nsteps), axis=1)
        ref_state = one.get_state()
    runs = []
    for _rep in range(2):
        with Engine(p, devices=[0] * G) as multi:
            assert multi.comm_size() == 0                  # copy exchange: no communicator
            assert multi.pair_kernel_name() == "pair_rows_generic_kernel"    # nothing set yet
            multi.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            assert multi.pair_kernel_name() == "pair_n3_kernel"
            m0 = multi.compute_forces()
            assert np.allclose(m0, e0, rtol=1e-13, atol=0)
            assert abs(multi.kinetic_energy() - k0) <= 1e-14 * k0
            sc = np.stack(multi.verlet_steps(nsteps), axis=1)
            st = multi.get_state()
            with pytest.raises(ljmd_amd.LjmdError, match="multi-device"):
                multi.step_begin()                         # split-phase API belongs to the one-process-per-GPU form
            runs.append((m0, sc, st))
    m0, sc, st = runs[0]
    assert np.max(np.abs(sc - ref) / np.abs(ref)) < 1e-11
    for key in ("r", "ru", "v", "a"):
        mine, want = np.stack(st[key]), np.stack(ref_state[key])
        assert np.abs(mine - want).max() < 1e-9 * max(np.abs(want).max(), 1.0), key
    assert runs[1][0] == m0 and np.array_equal(runs[1][1], sc)                 # run-to-run: bitwise
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(runs[1][2][key]), np.stack(st[key])), key


@pytest.mark.parametrize("n,G", [(16384, 4), (8192, 8)])
def test_multi_device_handle_ownership_migration(n, G, monkeypatch):
    """A rank owns an index range, i.e. a fixed set of particles that diffuses out of its slab in a liquid; every
    LJMD_MULTI_MIGRATE_EVERY steps the multi-device handle deals the particles out again by position.
    With a migration at set_state and before every 20-step segment: the energy series and the final state -- in the CALLER's
    particle order, unwrapped positions included -- against the same run without migration and against the single engine;
    set_unwrapped / set_accel round trips through the owner table; run-to-run bitwise.  The calls come in the production
    driver's order (md_simulation_gpu.f90): collect, snapshot_begin, enqueue the next segment -- a migration is due right
    there, with the snapshot in flight and, one segment later, with step records still pending -- snapshot_end."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    p, r, v = synthetic.make_config(n, seed=9)
    segs, seg = 6, 20

    def run(migrate_every, devices):
        monkeypatch.setenv("LJMD_MULTI_MIGRATE_EVERY", str(migrate_every))
        kw = dict(devices=devices) if devices else {}
        with Engine(p, **kw) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            eng.compute_forces()
            rows, snap = [], None
            eng.enqueue_steps(seg)
            for k in range(segs):
                if k == 4:
                    eng.enqueue_steps(seg)             # ... with the records of segment 4 still waiting on the devices
                    rows.append(np.stack(eng.collect_steps(2 * seg), axis=1))
                    continue
                if k == 5:
                    continue
                rows.append(np.stack(eng.collect_steps(seg), axis=1))
                if k == 2:
                    eng.snapshot_begin()
                eng.enqueue_steps(seg)                 # migration due, snapshot in flight
                if k == 2:
                    snap = eng.snapshot_end()
            st = eng.get_state()
            n_mig = eng.migrations()
            # round trip through the owner table: what get_state returned goes back in and comes out unchanged
            eng.set_unwrapped(*st["ru"])
            eng.set_accel(*st["a"])
            st2 = eng.get_state()
            for key in ("r", "ru", "v", "a"):
                assert np.array_equal(np.stack(st2[key]), np.stack(st[key])), key
            more = np.stack(eng.verlet_steps(5), axis=1)
        return np.concatenate(rows), st, snap, n_mig, more

    sc_m, st_m, snap_m, n_mig, more_m = run(seg, [0] * G)
    assert n_mig == segs                                       # at set_state and before every segment but the first
    sc_0, st_0, snap_0, n0, more_0 = run(0, [0] * G)
    assert n0 == 0
    sc_1, st_1, snap_1, _n1, more_1 = run(0, None)             # the single engine
    for sc in (sc_0, sc_1):
        assert np.max(np.abs(sc_m - sc) / np.abs(sc)) < 1e-9
    assert np.max(np.abs(more_m - more_1) / np.abs(more_1)) < 1e-9
    for ref in (st_0, st_1):
        for key in ("r", "ru", "v", "a"):
            mine, want = np.stack(st_m[key]), np.stack(ref[key])
            assert np.abs(mine - want).max() < 1e-7 * max(np.abs(want).max(), 1.0), key     # identities intact
    for key in ("r", "ru", "v", "a"):
        assert np.abs(np.stack(snap_m[key]) - np.stack(snap_1[key])).max() < 1e-7 * max(np.abs(np.stack(snap_1[key])).max(), 1.0), key
    sc_again, st_again, _s, _n, _m = run(seg, [0] * G)
    assert np.array_equal(sc_again, sc_m)
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(st_again[key]), np.stack(st_m[key])), key


def test_multi_device_handle_one_rank_equals_plain_engine_bitwise(monkeypatch):
    """n_gpus = 1: RCCL communicator over one device (ncclCommInitAll), no exchange needed -- bit for bit the plain
    engine, including the asynchronous production-loop entry points the Fortran driver uses."""
    p, r, v = synthetic.make_config(16384, seed=9)
    out = []
    for devices in (None, [0]):
        with Engine(p, devices=devices) as eng:
            eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
            e0 = eng.compute_forces()
            eng.enqueue_steps(12)
            eng.snapshot_begin()
            eng.enqueue_steps(10)                          # runs while the snapshot leaves the device
            snap = eng.snapshot_end()
            sc = np.stack(eng.collect_steps(10), axis=1)
            out.append((e0, sc, snap, eng.get_state(), eng.comm_size()))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    for key in ("r", "ru", "v", "a"):
        assert np.array_equal(np.stack(out[0][2][key]), np.stack(out[1][2][key])), key
        assert np.array_equal(np.stack(out[0][3][key]), np.stack(out[1][3][key])), key
    assert out[0][4] == 0 and out[1][4] == 1               # RCCL itself reports the 1-rank communicator


def test_multi_device_handle_rccl_collectives_one_rank(monkeypatch):
    """LJMD_FORCE_COLLECTIVES=1 on a 1-device multi handle: the grouped ncclAllGather / ncclReduceScatter calls of
    the single-process path really run (1-rank communicator, all this box can host), trajectory bitwise unchanged."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    p, r, v = synthetic.make_config(16384, seed=3)
    with Engine(p) as ref:
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e_ref = ref.compute_forces()
        sc_ref = np.stack(ref.verlet_steps(25), axis=1)
    monkeypatch.setenv("LJMD_FORCE_COLLECTIVES", "1")
    with Engine(p, devices=[0]) as eng:
        assert eng.comm_size() == 1
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.compute_forces() == e_ref
        assert np.array_equal(np.stack(eng.verlet_steps(25), axis=1), sc_ref)


def test_sharded_form_at_bench_size_eight_ranks_one_force_call():
    """The sharded decomposition at the size where it matters (n = 262144, G = 8: 128 row groups per rank, the
    NG/2 tie rule, several offset slices per rank, fpart[G] blocks, rank-ordered force sum), one force call through
    the multi-device handle on this one card against the single engine."""
    n, G = 262144, 8
    p, r, v = synthetic.make_config(n)
    with Engine(p) as one:
        one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e1 = one.compute_forces()
        a1 = np.stack(one.get_state(("a",))["a"])
    with Engine(p, devices=[0] * G) as multi:
        multi.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e8 = multi.compute_forces()
        a8 = np.stack(multi.get_state(("a",))["a"])
    assert np.allclose(e8, e1, rtol=1e-12, atol=0), (e8, e1)
    assert np.abs(a8 - a1).max() <= 1e-12 * np.abs(a1).max()
    assert np.abs(a8.sum(axis=1)).max() <= 1e-9 * np.abs(a1).max()             # Newton 3 across ranks


def test_config4_sharded_eight_ranks_n1048576(oracle):
    """BASELINE config 4 in its defining form: N = 1 048 576 = 4 * 64^3, FCC sites in the reference's order
    (md_initial_config_program.f90:144-178) + jitter, sharded over EIGHT ranks -- here eight rank engines on this box's one
    card through the multi-device handle (peer-copy exchange; fpart[8] blocks of 131 072 particles, NG/2 ties at
    NG = 4096, 3 MB per rank and step of positions).  One force call + 2 steps:
      * accelerations against the CPU oracle's full-matrix rows (the reference's per-pair arithmetic) on 2048-row blocks
        spread over the caller's index range, i.e. over the k-d blocks of all eight ranks: 1e-12 max|a|;
      * the three scalars and the two steps' series against the ONE-rank engine: 1e-12 relative;
      * Newton 3 across ranks: total force zero; the ranks own parts dealt by position (one deal at set_state)."""
    import ctypes
    import os
    n, G = 1048576, 8
    p, r, v = synthetic.make_config(n, lattice="fcc")
    with Engine(p) as one:
        one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e1 = one.compute_forces()
        sc1 = np.stack(one.verlet_steps(2), axis=1)
    with Engine(p, devices=[0] * G) as multi:
        multi.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert multi.migrations() == 1                      # dealt by position at set_state
        e8 = multi.compute_forces()
        a8 = np.stack(multi.get_state(("a",))["a"])
        sc8 = np.stack(multi.verlet_steps(2), axis=1)
    assert np.allclose(e8, e1, rtol=1e-12, atol=0), (e8, e1)
    assert np.max(np.abs(sc8 - sc1) / np.abs(sc1)) < 1e-12
    amax = np.abs(a8).max()
    assert np.abs(a8.sum(axis=1)).max() <= 1e-9 * amax * np.sqrt(n)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    po = oracle.derive_params(p.n, p.box_length, p.dt, p.rc)
    x, y, z = (np.ascontiguousarray(q) for q in r)
    worst = 0.0
    for i0 in range(3 * 2048, n, n // 6):                   # six blocks of 2048 rows across the FCC cell order
        ax, ay, az, _e, _d, _dd = oracle.rows_raw(po, i0, i0 + 2048, x, y, z)
        ao = 24.0 * np.stack([ax, ay, az])
        worst = max(worst, np.abs(a8[:, i0:i0 + 2048] - ao).max() / amax)
    print(f"config 4, 8 ranks: epot rel.diff vs one rank {abs(e8[0] - e1[0]) / abs(e1[0]):.2e}, accelerations vs oracle rows "
          f"{worst:.2e} max|a| ({cores} oracle threads)")
    assert worst <= 1e-12


@pytest.mark.parametrize("ladder", ["", "multi-host"])
def test_bench_ladder_rehearsal_on_one_card(tmp_path, ladder):
    """bench.py --gpus 2 on this one-GPU box with every rank on device 0 (LJMD_BENCH_SHARE_DEVICE=1): RCCL refuses two
    ranks on one device, so the first two rungs of the launch ladder fail (for real, not by stand-ins) and the
    peer-copy rung delivers the line -- watchdog, process-tree handling and the single-process worker end to end.
    ladder = "multi-host": the last rung alone (pinned-host staging).  Either line carries the per-rank figures the
    first multi-GPU session is read by (profiles/r04_multi_gpu_first_contact.md)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LJMD_BENCH_SHARE_DEVICE="1")
    if ladder:
        env.update(LJMD_BENCH_LADDER=ladder)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--particles", "32768", "--no-liquid"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len([ln for ln in out.stdout.splitlines() if ln.strip()]) == 1
    cfg = lines[0]["config"]
    if ladder:
        assert cfg["launch_mode"] == "multi-host" and "pinned host" in cfg["exchange"]
        assert [(a["mode"], a["outcome"]) for a in cfg["ladder"]] == [("multi-host", "ok")]
    else:
        assert cfg["launch_mode"] == "multi-copy" and "peer-to-peer" in cfg["exchange"]
        assert [a["mode"] for a in cfg["ladder"]] == ["ranks-rccl", "multi-rccl", "multi-copy"]
        assert [a["outcome"] for a in cfg["ladder"]][2] == "ok" and all(a["outcome"].startswith("exit") for a in cfg["ladder"][:2])
    assert cfg["rccl_ranks_seen"] == 0 and len(cfg["pair_kernel_ms_per_rank"]) == 2
    # the keys every rung's line carries (also at one GPU): whole step per rank, both exchanges, migrations so far
    assert len(cfg["per_rank_ms"]) == len(cfg["position_exchange_ms"]) == len(cfg["force_exchange_ms"]) == 2
    assert all(ms > 0 for ms in cfg["per_rank_ms"] + cfg["position_exchange_ms"] + cfg["force_exchange_ms"])
    assert cfg["migrations"] >= 1                              # the deal by position at the start
    assert all(ms > 0 for ms in cfg["pair_kernel_ms_per_rank"] + cfg["position_exchange_ms_per_rank"] +
               cfg["force_exchange_ms_per_rank"])
    assert lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0 and lines[0]["energy_check"]["rel_drift"] < 1e-3


@pytest.mark.parametrize("exchange", ["copy", "host"])
def test_multi_device_handle_exchange_forms_are_bitwise_equal(exchange, monkeypatch):
    """The three exchanges of the multi-device handle move the same bytes and add the force blocks in the same rank
    order: peer copies (default on one card), pinned-host staging, and both with the exchanges on the engine streams
    (LJMD_OVERLAP_EXCHANGE=0) give bit-identical trajectories; gather-kernel sizes (no force exchange) included."""
    out = []
    for n, env in ((16384, {"LJMD_N3_MIN_N": "1"}), (2048, {})):
        p, r, v = synthetic.make_config(n, seed=17)
        runs = []
        for x, ov in (("copy", "1"), (exchange, "1"), (exchange, "0")):
            for k, val in dict(env, LJMD_MULTI_EXCHANGE=x, LJMD_OVERLAP_EXCHANGE=ov).items():
                monkeypatch.setenv(k, val)
            with Engine(p, devices=[0, 0, 0, 0]) as eng:
                eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
                e0 = eng.compute_forces()
                sc = np.stack(eng.verlet_steps(45), axis=1)         # crosses re-sorts
                runs.append((e0, sc, eng.get_state()))
        for e0, sc, st in runs[1:]:
            assert e0 == runs[0][0] and np.array_equal(sc, runs[0][1])
            for key in ("r", "ru", "v", "a"):
                assert np.array_equal(np.stack(st[key]), np.stack(runs[0][2][key])), key
        out.append(runs[0][1])
    assert all(np.all(np.isfinite(sc)) for sc in out)


def test_multi_device_handle_rc_at_half_box_falls_back_to_the_generic_kernel():
    """The reference accepts rc_over_L up to 0.5 and rejects only rc >= L/2 (md_types.f90:152).  Within 1e-9 of L/2 the
    fast kernels' minimum image is not safe: a single engine takes the exact generic kernel, and a multi-rank run must do
    the same on every rank (no Newton-3 force exchange allocated) instead of refusing to run."""
    n = 4096
    p0, r, v = synthetic.make_config(n, seed=4)
    p = init_params(n, p0.box_length, p0.dt, 0.5 * p0.box_length * (1.0 - 1e-12))
    with Engine(p) as one:
        one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert one.pair_kernel_name() == "pair_rows_generic_kernel"
        e1 = one.compute_forces()
        sc1 = np.stack(one.verlet_steps(5), axis=1)
    with Engine(p, devices=[0, 0]) as two:
        two.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert two.pair_kernel_name() == "pair_rows_generic_kernel"
        e2 = two.compute_forces()
        sc2 = np.stack(two.verlet_steps(5), axis=1)
    assert np.allclose(e2, e1, rtol=1e-13, atol=0)
    assert np.max(np.abs(sc2 - sc1) / np.abs(sc1)) < 1e-11


def test_multi_device_handle_poisoned_after_failed_batch_and_recovers(monkeypatch):
    """A failure half-way through a step of the multi-device handle (injected in rank 0's force phase, after every
    rank's drift and the position exchange) leaves the ranks a phase apart: the parent refuses to step until
    ljmd_set_state, which re-synchronises every rank; afterwards the trajectory is the fresh one, bitwise."""
    monkeypatch.setenv("LJMD_N3_MIN_N", "1")
    p, r, v = synthetic.make_config(8192, seed=21)
    with Engine(p, devices=[0, 0, 0, 0]) as ref:
        ref.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        e_ref = ref.compute_forces()
        sc_ref = np.stack(ref.verlet_steps(12), axis=1)
    monkeypatch.setenv("LJMD_INJECT_FAILURE_AT_STEP", "4")
    with Engine(p, devices=[0, 0, 0, 0]) as eng:
        monkeypatch.delenv("LJMD_INJECT_FAILURE_AT_STEP")
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.compute_forces() == e_ref
        with pytest.raises(ljmd_amd.LjmdError, match="injected failure"):
            eng.verlet_steps(10)
        with pytest.raises(ljmd_amd.LjmdError, match="poisoned"):
            eng.verlet_steps(1)
        with pytest.raises(ljmd_amd.LjmdError, match="poisoned"):
            eng.compute_forces()
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        assert eng.compute_forces() == e_ref
        assert np.array_equal(np.stack(eng.verlet_steps(12), axis=1), sc_ref)
