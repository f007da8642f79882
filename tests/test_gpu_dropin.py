"""-m gpu: the Fortran side of the boundary.

1. bin/md_simulation_gpu -- our thin Fortran driver (ISO_C_BINDING -> libljmd.so), BASELINE
   config 1 end to end from the reference's own input file and rv_init.dat.
2. oracle/_ref/md_simulation_program_gpu and md_initial_config_program_gpu -- the REFERENCE's
   unmodified main programs and base/stats modules, compiled (in the build container, where
   /root/reference exists) against OUR drop-in modules lj_potential_energy / verlet: the
   literal "swap two files, link -lljmd" integration of INTEGRATION.md.  Compared with the
   files the pure reference wrote (tests/golden/ref_run_n108_*).
"""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from ljmd_amd import io_formats

pytestmark = pytest.mark.gpu

PKG = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd"
REF = ROOT / "oracle" / "_ref"


def _workdir(tmp_path, tag):
    src = GOLDEN / (tag if tag.startswith("ref_run_") else f"ref_run_n108_{tag}")
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs" / "one_run").mkdir(parents=True)
    shutil.copy(src / "input_simulation_parameters.txt", tmp_path / "inputs")
    shutil.copy(src / "rv_init.dat", tmp_path / "outputs")
    return src


HORIZON_STEPS = 200      # inside it the GPU series equals the reference's to <= 1e-10 (tests/test_gpu_parity.py)


def _compare_run(tmp_path, src, n_rows, dt=0.005):
    mine = io_formats.read_energies(tmp_path / "outputs" / "one_run" / "instantaneous_energies.dat")
    ref = io_formats.read_energies(src / "instantaneous_energies.dat")
    assert mine.shape == ref.shape == (n_rows, 6)
    # text file: 7 significant digits; up to step 1000 the GPU stays within ~1e-7 of the reference
    assert np.allclose(mine, ref, rtol=5e-6, atol=0), np.max(np.abs(mine - ref) / np.abs(ref))
    l1 = (tmp_path / "outputs" / "one_run" / "instantaneous_energies.dat").read_text().splitlines()
    l2 = (src / "instantaneous_energies.dat").read_text().splitlines()
    assert l1[0] == l2[0] == "# time   epot   ekin   etot   T   P"
    # every row sampled inside the parity horizon: the same BYTES as the reference's file
    inside = [k for k in range(n_rows) if round(ref[k, 0] / dt) <= HORIZON_STEPS]
    assert inside or ref[0, 0] / dt > HORIZON_STEPS
    for k in inside:
        assert l1[1 + k] == l2[1 + k], (k, l1[1 + k], l2[1 + k])
    return len(inside)


def _compare_statistics_files(out_dir, src):
    """corr_*.dat, corrmean_*.dat and md_final_results.txt of a GPU run against the reference's files.
    The GPU trajectory stays within ~1e-7 of the reference up to step 1000, so sample means
    agree to 1e-6; fluctuation quantities (std, coefficients, autocovariances) amplify that by the
    ratio value/fluctuation, hence the looser bounds.  (With IDENTICAL samples the files are
    byte-identical: tests/test_stats.py.)"""
    mine = (out_dir / "md_final_results.txt").read_text().split()
    ref = (src / "md_final_results.txt").read_text().split()
    assert len(mine) == len(ref)
    pairs = [(a, b) for a, b in zip(mine, ref)]
    assert all(a == b for a, b in pairs if not _isnum(b))                  # labels and layout
    nums = [(float(a), float(b)) for a, b in pairs if _isnum(b)]
    assert len(nums) == 32
    for a, b in nums[:9]:                                                    # run parameters: exact
        assert a == b
    for k, (a, b) in enumerate(nums[9:19]):                                  # <.> (even k) 1e-6, std (odd k) 1e-3
        assert abs(a - b) <= (1e-6 if k % 2 == 0 else 1e-3) * abs(b), (k, a, b)
    for a, b in nums[19:]:                                                   # coefficients
        assert abs(a - b) <= 1e-3 * abs(b), (a, b)
    for kind in ("corr", "corrmean"):
        for obs in ("epot", "ekin", "etot", "temp", "press"):
            m = np.loadtxt(out_dir / f"{kind}_{obs}.dat")
            r = np.loadtxt(src / f"{kind}_{obs}.dat")
            assert m.shape == r.shape
            assert (out_dir / f"{kind}_{obs}.dat").read_text().splitlines()[0] == \
                   (src / f"{kind}_{obs}.dat").read_text().splitlines()[0]
            assert np.array_equal(m[:, 0], r[:, 0])
            assert np.abs(m[:, 1] - r[:, 1]).max() <= 1e-3 * abs(r[0, 1]), (kind, obs)
            assert np.abs(m[:, 2] - r[:, 2]).max() <= 1e-3, (kind, obs)


@pytest.mark.parametrize("tag,n_rows", [("oi10", 90), ("oi100", 9)])
def test_thin_fortran_driver_statistics_files(tmp_path, tag, n_rows):
    """the thin driver also writes the end-of-run statistics files."""
    src = _workdir(tmp_path, tag)
    subprocess.run([str(PKG / "bin" / "md_simulation_gpu")], cwd=tmp_path, check=True, timeout=120)
    _compare_run(tmp_path, src, n_rows)
    _compare_statistics_files(tmp_path / "outputs" / "one_run", src)


def test_python_production_loop_statistics_files(tmp_path):
    from ljmd_amd import simulation
    src = _workdir(tmp_path, "oi10")
    res = simulation.run_md_simulation(tmp_path)
    assert res.n_samples == 90 and "gamma" in res.summary["coefficients"]
    _compare_statistics_files(tmp_path / "outputs" / "one_run", src)


def test_thin_fortran_driver_config1(tmp_path):
    exe = PKG / "bin" / "md_simulation_gpu"
    assert exe.exists(), "run __graft_entry__.build() first"
    src = _workdir(tmp_path, "oi100")
    out = subprocess.run([str(exe)], cwd=tmp_path, check=True, capture_output=True, text=True, timeout=120)
    assert "steps/s" in out.stdout
    _compare_run(tmp_path, src, 9)
    h1, s1 = io_formats.read_rva(tmp_path / "outputs" / "one_run" / "rva.dat")
    h2, s2 = io_formats.read_rva(src / "rva.dat")
    assert h1 == h2 and s1.shape == s2.shape
    assert (tmp_path / "outputs" / "one_run" / "rva.dat").stat().st_size == (src / "rva.dat").stat().st_size == 93636
    # the snapshot at step 200 (inside the horizon): all four records r, ru, v, a.  The trajectories differ by
    # the summation order of the forces, amplified over 200 steps + the 100 warm-up steps of rv_init.dat's own
    # history: fp64 records cannot be byte-equal; r, ru, v to 1e-11 absolute, a to 1e-11 of max|a|
    dev = [np.abs(s1[0, w] - s2[0, w]).max() for w in range(4)]
    print("rva.dat snapshot at step 200: max |mine - reference| of r, ru, v, a =", dev)
    assert max(dev[:3]) < 1e-11 and dev[3] < 1e-11 * np.abs(s2[0, 3]).max()
    # later snapshots (steps 300..1000) leave the horizon: chaos, bounded only loosely
    assert np.abs(s1[:, 0] - s2[:, 0]).max() < 1e-3


@pytest.mark.parametrize("tag,n_rows,ranks", [("oi100", 9, "2"), ("oi10", 90, "4")])
def test_thin_fortran_driver_several_ranks_from_one_process(tmp_path, tag, n_rows, ranks):
    """north-star: "thin Fortran driver ... sharded across the GPUs".  LJMD_GPUS > 1 makes md_simulation_gpu create its
    engine with ljmd_create_multi; everything else in the driver is the single-GPU code.  Here the ranks share this
    box's one card (LJMD_DEVICES=0,0,...: peer-copy exchange); BASELINE config 1 against the reference's files with
    the same bounds as the single-engine run, and the binary trajectory against the single-engine run's."""
    src = _workdir(tmp_path, tag)
    env = dict(os.environ, LJMD_GPUS=ranks, LJMD_DEVICES=",".join(["0"] * int(ranks)))
    out = subprocess.run([str(PKG / "bin" / "md_simulation_gpu")], cwd=tmp_path, check=True, capture_output=True,
                         text=True, timeout=300, env=env)
    assert "steps/s" in out.stdout
    _compare_run(tmp_path, src, n_rows)
    _compare_statistics_files(tmp_path / "outputs" / "one_run", src)
    h1, s1 = io_formats.read_rva(tmp_path / "outputs" / "one_run" / "rva.dat")
    assert h1["n"] == 108 and s1.shape == (n_rows, 4, 3, 108)
    if (src / "rva.dat").exists():
        h2, s2 = io_formats.read_rva(src / "rva.dat")
        assert h1 == h2
        dev = [np.abs(s1[0, w] - s2[0, w]).max() for w in range(4)]
        assert max(dev[:3]) < 1e-11 and dev[3] < 1e-11 * np.abs(s2[0, 3]).max()


def test_thin_fortran_driver_error_convention(tmp_path):
    """Missing outputs/one_run -> the reference's `stop` message (md_simulation_program.f90:250)."""
    exe = PKG / "bin" / "md_simulation_gpu"
    src = GOLDEN / "ref_run_n108_oi100"
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs").mkdir()
    shutil.copy(src / "input_simulation_parameters.txt", tmp_path / "inputs")
    shutil.copy(src / "rv_init.dat", tmp_path / "outputs")
    out = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    # `stop 'text'` exits with status 0 under flang, exactly like the reference's own programs
    assert "cannot open outputs/one_run/rva.dat" in (out.stdout + out.stderr)
    assert not (tmp_path / "outputs" / "one_run").exists()


def test_thin_fortran_pipeline_init_then_production(tmp_path):
    """Our two thin drivers back to back (the reference's run_all.sh sequence: init -> production):
    md_initial_config_gpu must reproduce the reference's rv_init.dat (FCC + ran3 velocities + rescale +
    100 warm-up steps on the GPU), and the production driver fed with OUR rv_init.dat must reproduce
    the reference's energies file."""
    src = _workdir(tmp_path, "oi100")
    (tmp_path / "outputs" / "rv_init.dat").unlink()
    subprocess.run([str(PKG / "bin" / "md_initial_config_gpu")], cwd=tmp_path, check=True, timeout=120)
    r1, v1 = io_formats.read_rv_init(tmp_path / "outputs" / "rv_init.dat", 108)
    r2, v2 = io_formats.read_rv_init(src / "rv_init.dat", 108)
    assert (tmp_path / "outputs" / "rv_init.dat").stat().st_size == 5200
    print("rv_init after 100 GPU warm-up steps: max |dr|, |dv| vs the reference =", np.abs(r1 - r2).max(), np.abs(v1 - v2).max())
    assert np.abs(r1 - r2).max() < 1e-12 and np.abs(v1 - v2).max() < 1e-11
    subprocess.run([str(PKG / "bin" / "md_simulation_gpu")], cwd=tmp_path, check=True, timeout=120)
    _compare_run(tmp_path, src, 9)


def test_thin_fortran_pipeline_second_configuration(tmp_path):
    """Everything different from config 1: k = 4 (N = 256), L = 6.5 (rho = 0.93), rc = 0.35 L, dt = 0.002,
    target energy -900, 600 steps sampled every 20 after 40.  Our init driver + our production driver against
    the files the reference's two programs wrote for this input."""
    src = _workdir(tmp_path, "ref_run_n256_k4")
    (tmp_path / "outputs" / "rv_init.dat").unlink()
    subprocess.run([str(PKG / "bin" / "md_initial_config_gpu")], cwd=tmp_path, check=True, timeout=120)
    r1, v1 = io_formats.read_rv_init(tmp_path / "outputs" / "rv_init.dat", 256)
    r2, v2 = io_formats.read_rv_init(src / "rv_init.dat", 256)
    assert (tmp_path / "outputs" / "rv_init.dat").stat().st_size == 2 * (3 * 256 * 8 + 8)
    assert np.abs(r1 - r2).max() < 1e-10 and np.abs(v1 - v2).max() < 1e-9
    subprocess.run([str(PKG / "bin" / "md_simulation_gpu")], cwd=tmp_path, check=True, timeout=120)
    assert _compare_run(tmp_path, src, 28, dt=0.002) == 8        # steps 60..200: byte-equal rows
    _compare_statistics_files(tmp_path / "outputs" / "one_run", src)
    h, snaps = io_formats.read_rva(tmp_path / "outputs" / "one_run" / "rva.dat")
    assert h == dict(n=256, box_length=6.5, dt=0.002, output_interval=20, n_snapshots_expected=28)
    assert snaps.shape == (28, 4, 3, 256)


@pytest.mark.parametrize("gpus", ["1", "2"])
def test_thin_fortran_driver_sampled_segments_write_the_same_bytes(tmp_path, gpus):
    """The production driver runs the steps between two samples with the forces-only pair kernel
    (ljmd_enqueue_steps_sampled; the reference reads the energy sums only at md_simulation_program.f90:361).
    N = 5324 (k = 11: Newton-3 kernel), 120 steps sampled every 30 after 30, init driver first (its warm-up uses
    ljmd_verlet_steps with NULL outputs): LJMD_SAMPLED_STEPS=1 (default) and =0 (sums on every step) must write
    byte-identical instantaneous_energies.dat, rva.dat and md_final_results.txt -- one process and LJMD_GPUS=2."""
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs" / "one_run").mkdir(parents=True)
    (tmp_path / "inputs" / "input_simulation_parameters.txt").write_text(
        "k total_steps output_interval warmup_steps\n11 120 30 30\ndt L rc_over_L\n5.d-3 18.8d0 0.49d0\n"
        "target_total_energy\n-2.2d4\n")
    subprocess.run([str(PKG / "bin" / "md_initial_config_gpu")], cwd=tmp_path, check=True, timeout=300)
    files = ("instantaneous_energies.dat", "rva.dat", "md_final_results.txt")
    got = {}
    # (third run: LJMD_ASYNC_IO=0 -- every sample written before the next steps are enqueued instead of beside them)
    for sampled, async_io in (("1", "1"), ("0", "1"), ("1", "0")):
        for f in files:
            (tmp_path / "outputs" / "one_run" / f).unlink(missing_ok=True)
        env = dict(os.environ, LJMD_SAMPLED_STEPS=sampled, LJMD_ASYNC_IO=async_io, LJMD_GPUS=gpus,
                   LJMD_DEVICES=",".join(["0"] * int(gpus)))
        out = subprocess.run([str(PKG / "bin" / "md_simulation_gpu")], cwd=tmp_path, check=True, capture_output=True,
                             text=True, timeout=300, env=env)
        assert "N=5324" in out.stdout
        got[sampled + async_io] = [(tmp_path / "outputs" / "one_run" / f).read_bytes() for f in files]
    for other in ("01", "10"):
        for f, a, b in zip(files, got["11"], got[other]):
            assert a == b, (f, other)
    rows = io_formats.read_energies(tmp_path / "outputs" / "one_run" / "instantaneous_energies.dat")
    assert rows.shape == (3, 6) and np.all(np.isfinite(rows))


@pytest.mark.skipif(not (REF / "md_simulation_program_gpu").exists(),
                    reason="drop-in binary is built only where /root/reference exists")
def test_reference_main_program_with_gpu_shim(tmp_path):
    """The reference's own md_simulation_program.f90 + md_means/md_correlations/..., with only
    lj_potential_energy.f90 and verlet.f90 swapped for our shim."""
    src = _workdir(tmp_path, "oi10")
    subprocess.run([str(REF / "md_simulation_program_gpu")], cwd=tmp_path, check=True, timeout=300)
    _compare_run(tmp_path, src, 90)
    # the reference's own summary file is produced by its own statistics code on our numbers
    mine = (tmp_path / "outputs" / "one_run" / "md_final_results.txt").read_text().split()
    ref = (src / "md_final_results.txt").read_text().split()
    assert len(mine) == len(ref)
    nums = [(float(a), float(b)) for a, b in zip(mine, ref) if _isnum(a) and _isnum(b)]
    assert len(nums) >= 10
    # means over 90 samples agree far inside their own statistical error
    for a, b in nums[:6]:
        assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (a, b)


@pytest.mark.skipif(not (REF / "md_initial_config_program_gpu").exists(),
                    reason="drop-in binary is built only where /root/reference exists")
def test_reference_init_program_with_gpu_shim(tmp_path):
    """Second caller of the hot path (md_initial_config_program.f90:91,104,113-116): FCC lattice,
    ran3 velocities, rescale, 100 warm-up verlet_step calls -- through the GPU shim."""
    src = _workdir(tmp_path, "oi10")
    (tmp_path / "outputs" / "rv_init.dat").unlink()
    subprocess.run([str(REF / "md_initial_config_program_gpu")], cwd=tmp_path, check=True, timeout=300)
    r1, v1 = io_formats.read_rv_init(tmp_path / "outputs" / "rv_init.dat", 108)
    r2, v2 = io_formats.read_rv_init(src / "rv_init.dat", 108)
    assert np.abs(r1 - r2).max() < 1e-10 and np.abs(v1 - v2).max() < 1e-9


@pytest.mark.parametrize("tag,n", [("k3", 108), ("k4", 256)])
def test_thin_fortran_init_without_warmup_against_reference_bits(tmp_path, tag, n):
    """warmup_steps = 0: rv_init.dat is the lattice + the ran3 velocities after centre-of-mass removal and the
    rescale -- no dynamics.  Positions: the reference's bytes.  Velocities: the host arithmetic is bit-exact
    given the same lattice energy (tests/test_init_host.py); here epot comes from the GPU pair kernel, whose
    summation order differs from the sequential loop (<= 1e-13 relative), and enters through
    scale = sqrt((E_target - epot) / ekin): every velocity within 1e-13 relative of the reference's, and the
    ratio v_mine / v_ref is ONE constant for all 3N components (to the 2 ulp of the two roundings)."""
    src = GOLDEN / f"init_{tag}_warm0"
    (tmp_path / "inputs").mkdir()
    (tmp_path / "outputs").mkdir()
    shutil.copy(src / "input_simulation_parameters.txt", tmp_path / "inputs")
    subprocess.run([str(PKG / "bin" / "md_initial_config_gpu")], cwd=tmp_path, check=True, timeout=120)
    mine = (tmp_path / "outputs" / "rv_init.dat").read_bytes()
    ref = (src / "rv_init.dat").read_bytes()
    rec = 3 * n * 8 + 8
    assert len(mine) == len(ref) == 2 * rec
    assert mine[:rec] == ref[:rec]                                       # record 1 (positions): byte-identical
    _r1, v1 = io_formats.read_rv_init(tmp_path / "outputs" / "rv_init.dat", n)
    _r2, v2 = io_formats.read_rv_init(src / "rv_init.dat", n)
    ratio = v1 / v2
    print(f"init {tag}: velocity ratio mine/reference - 1 = {ratio.mean() - 1:.3e}, spread {np.ptp(ratio):.1e}, "
          f"bytes equal: {mine == ref}")
    assert np.abs(ratio - 1.0).max() <= 1e-13
    assert np.ptp(ratio) <= 1e-15


def _isnum(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def test_shim_compiled_with_the_tail_switch_off(tmp_path, oracle):
    """The reference's `use_tail_corrections` is a compile-time parameter of its module lj_potential_energy (:36); the
    drop-in module declares the same parameter and the library honours it.  Flip it in a copy of OUR shim source, build
    the thin driver against that copy (amdflang is on the box), run BASELINE config 1: every epot of the energies file
    is the reference's minus the tail constant, the pressure loses the virial tail, ekin is untouched."""
    if shutil.which("amdflang") is None:
        pytest.skip("no amdflang on this box")
    fsrc = PKG / "fortran"
    build = tmp_path / "build"
    (build / "obj").mkdir(parents=True)
    mods = ["define_precision", "md_types", "read_input_files", "random_numbers", "md_init_host", "ljmd_c_api",
            "lj_potential_energy", "verlet", "md_stats", "md_run_outputs"]
    for m in mods:
        text = (fsrc / f"{m}.f90").read_text()
        if m == "lj_potential_energy":
            assert "use_tail_corrections = .true." in text
            text = text.replace("use_tail_corrections = .true.", "use_tail_corrections = .false.")
        (build / f"{m}.f90").write_text(text)
    (build / "md_simulation_gpu.f90").write_text((fsrc / "md_simulation_gpu.f90").read_text())
    flags = ["-O2", "-ffp-contract=off", "-module-dir", str(build / "obj"), "-I", str(build / "obj")]
    for m in mods:
        subprocess.run(["amdflang", *flags, "-c", str(build / f"{m}.f90"), "-o", str(build / "obj" / f"{m}.o")], check=True)
    exe = build / "md_simulation_notail"
    subprocess.run(["amdflang", *flags, *[str(build / "obj" / f"{m}.o") for m in mods], str(build / "md_simulation_gpu.f90"),
                    f"-L{PKG}", "-lljmd", f"-Wl,-rpath,{PKG}", "-o", str(exe)], check=True)
    run = tmp_path / "run"
    run.mkdir()
    src = _workdir(run, "oi100")
    out = subprocess.run([str(exe)], cwd=run, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    mine = io_formats.read_energies(run / "outputs" / "one_run" / "instantaneous_energies.dat")
    ref = io_formats.read_energies(src / "instantaneous_energies.dat")
    from ljmd_amd import read_input_files
    prm = read_input_files.read_simulation_parameters(run / "inputs" / "input_simulation_parameters.txt").params
    po = oracle.derive_params(prm.n, prm.box_length, prm.dt, prm.rc)
    te, td, _tdd = oracle.tail_corrections(po)
    assert mine.shape == ref.shape
    # columns: time epot ekin etot T P (7 significant digits in the file)
    assert np.allclose(mine[:, 2], ref[:, 2], rtol=5e-6, atol=0) and np.allclose(mine[:, 4], ref[:, 4], rtol=5e-6, atol=0)
    assert np.allclose(ref[:, 1] - mine[:, 1], te, rtol=2e-4, atol=0), (ref[:3, 1] - mine[:3, 1], te)
    # P = rho T - d_epot / (3 V)  (md_means.f90:227): the virial tail's share
    dp = -td / (3.0 * po.volume)
    assert np.allclose(ref[:, 5] - mine[:, 5], dp, rtol=2e-3, atol=0), (ref[:3, 5] - mine[:3, 5], dp)
