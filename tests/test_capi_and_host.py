"""CPU-only: the C-ABI library loads and exports every symbol include/ljmd.h declares, fails
loudly without a GPU, and the host-side mirror (parameters, input parser, file formats,
synthetic configs) matches the reference's fixtures."""
import ctypes as C
import re
import struct

import numpy as np
import pytest

import ljmd_amd
from ljmd_amd import _lib, io_formats, md_types, read_input_files, synthetic
from conftest import GOLDEN, ROOT


def _declared_symbols():
    text = (ROOT / "include" / "ljmd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ljmd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 25
    lib = C.CDLL(str(_lib.LIB_PATH))
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ljmd.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with the header"
    assert _lib.load().ljmd_version().decode().startswith("ljmd ")


def test_no_gpu_fails_loudly_no_cpu_fallback():
    lib = _lib.load()
    if lib.ljmd_device_count() > 0:
        pytest.skip("a HIP device is present")
    p = md_types.init_params(108, 5.129927840030091, 0.005, 0.49 * 5.129927840030091)
    with pytest.raises(ljmd_amd.LjmdError) as ei:
        ljmd_amd.Engine(p)
    assert ei.value.code == _lib.LJMD_ERR_NO_DEVICE and "no CPU path" in ei.value.message
    st = md_types.init_state(p)
    with pytest.raises(ljmd_amd.LjmdError) as e2:
        ljmd_amd.compute_lj_potential_energy(p, st)
    assert e2.value.code == _lib.LJMD_ERR_NO_DEVICE
    with pytest.raises(ljmd_amd.LjmdError):
        ljmd_amd.verlet_step(p, st)


def test_create_argument_guards_before_device_probe():
    lib = _lib.load()
    h = C.c_void_p()
    for args in ((0, 10.0, 0.005, 2.0), (10, -1.0, 0.005, 2.0), (10, 10.0, 0.005, 5.0),
                 (10, 10.0, 0.0, 2.0), (10, 10.0, 0.005, 0.0)):
        rc = lib.ljmd_create(C.byref(h), args[0], args[1], args[2], args[3], 0, 0, 0, 1)
        assert rc == _lib.LJMD_ERR_INVALID_ARG and _lib.last_error()
    assert lib.ljmd_create(C.byref(h), 10, 10.0, 0.005, 2.0, 0, 0, 0, 3) == _lib.LJMD_ERR_INVALID_ARG  # 10 % 3
    assert lib.ljmd_create(C.byref(h), 10, 10.0, 0.005, 2.0, 7, 0, 0, 1) == _lib.LJMD_ERR_INVALID_ARG  # precision


def test_derived_params_match_oracle_bitwise(oracle):
    for n, L, dt, rc in ((108, 5.129927840030091, 0.005, 0.49 * 5.129927840030091), (4096, 17.235477520255067, 1e-4, 7.0)):
        p = md_types.init_params(n, L, dt, rc)
        o = oracle.derive_params(n, L, dt, rc)
        for f in ("inv_box_length", "volume", "density", "dt_half", "dt_square_half", "rc_square"):
            assert getattr(p, f) == getattr(o, f), f
    with pytest.raises(ValueError, match="must be < L/2"):
        md_types.init_params(10, 10.0, 0.005, 5.0)
    with pytest.raises(ValueError, match="dt must be > 0"):
        md_types.init_params(10, 10.0, 0.0, 2.0)


def test_input_parser_reference_file_and_grammar():
    ctl = read_input_files.read_simulation_parameters(GOLDEN / "ref_run_n108_oi10" / "input_simulation_parameters.txt")
    assert (ctl.params.n, ctl.total_steps, ctl.output_interval, ctl.warmup_steps) == (108, 1000, 10, 100)
    assert ctl.params.box_length == 5.129927840030091 and ctl.params.dt == 5e-3
    assert ctl.params.rc == 0.49 * 5.129927840030091 and ctl.target_total_energy == -500.0
    # the shipped reference input: header-word lines are skipped because they fail the numeric read
    text = """# comment
k   total_steps   output_interval   warmup_steps
5   500000        100               5000

dt        L     rc_over_L
1.d-4    10.0  0.49d0
 # an indented hash is NOT a comment for the reference (column 1 only) but fails the numeric read
target_total_energy
-555.d00
"""
    c = read_input_files.parse_simulation_parameters(text)
    assert (c.params.n, c.total_steps, c.output_interval, c.warmup_steps) == (500, 500000, 100, 5000)
    assert c.params.dt == 1e-4 and c.params.box_length == 10.0 and c.rc_over_L == 0.49
    assert c.target_total_energy == -555.0
    with pytest.raises(ValueError, match="rc_over_L must be <= 0.5"):
        read_input_files.parse_simulation_parameters("1 1 1 0\n0.1 10 0.6\n1\n")
    with pytest.raises(ValueError, match="must be < L/2"):        # 0.5 passes the reader, fails md_types (:152)
        read_input_files.parse_simulation_parameters("1 1 1 0\n0.1 10 0.5\n1\n")
    with pytest.raises(ValueError, match="missing Block 3"):
        read_input_files.parse_simulation_parameters("1 1 1 0\n0.1 10 0.4\n")
    with pytest.raises(ValueError, match="k must be > 0"):
        read_input_files.parse_simulation_parameters("0 1 1 0\n")


def test_rv_init_and_rva_formats_roundtrip_against_reference_files(tmp_path):
    src = GOLDEN / "ref_run_n108_oi100"
    r, v = io_formats.read_rv_init(src / "rv_init.dat", 108)
    io_formats.write_rv_init(tmp_path / "rv.dat", *r, *v)
    assert (tmp_path / "rv.dat").read_bytes() == (src / "rv_init.dat").read_bytes()      # byte-identical
    header, snaps = io_formats.read_rva(src / "rva.dat")
    assert header == dict(n=108, box_length=5.129927840030091, dt=0.005, output_interval=100,
                          n_snapshots_expected=9)
    assert snaps.shape == (9, 4, 3, 108)
    with io_formats.RvaWriter(tmp_path / "rva.dat", 108, header["box_length"], 0.005, 100, 9) as w:
        for s in snaps:
            w.write_snapshot(s[0], s[1], s[2], s[3])
    assert (tmp_path / "rva.dat").read_bytes() == (src / "rva.dat").read_bytes()
    # record markers: 4-byte little-endian, 28-byte header payload
    raw = (src / "rva.dat").read_bytes()
    assert struct.unpack("<i", raw[:4])[0] == 28 and len(raw) == 36 + 9 * 4 * (8 + 3 * 108 * 8)
    with pytest.raises(ValueError):
        io_formats.read_rv_init(src / "rv_init.dat", 107)


def test_energies_row_format_matches_reference_text():
    src = GOLDEN / "ref_run_n108_oi10" / "instantaneous_energies.dat"
    lines = src.read_text().splitlines()
    assert lines[0] == io_formats.ENERGIES_HEADER
    rows = io_formats.read_energies(src)
    for line, row in zip(lines[1:6], rows[:5]):
        # formatting a value that is already rounded to 7 digits must reproduce the text exactly
        assert io_formats.energies_row(*row) == line
    assert io_formats.fortran_1pe13_6(-1.5e-120) == "-1.500000-120"
    assert io_formats.fortran_1pe13_6(0.0) == " 0.000000E+00"


def test_synthetic_configs_are_deterministic_and_sane():
    p, r, v = synthetic.make_config(4096)
    p2, r2, v2 = synthetic.make_config(4096)
    assert np.array_equal(r, r2) and np.array_equal(v, v2)
    g = np.load(GOLDEN / "force_n4096.npz")
    assert np.array_equal(r, g["r"]), "fixture inputs were generated by this very recipe"
    assert abs(p.n / p.volume - 0.8) < 1e-12 and abs(p.rc / p.box_length - 0.49) < 1e-15
    assert np.all(r >= 0) and np.all(r < p.box_length)
    assert np.abs(v.sum(axis=1)).max() < 1e-9
    assert abs(0.5 * np.sum(v * v) - 1.5 * 4096) < 1e-8
    p3, r3, _ = synthetic.make_config(1048576 // 64, lattice="fcc")   # 4 * 16^3
    assert p3.n == 16384 and r3.shape == (3, 16384)
    assert synthetic.make_config(262144)[0].box_length == pytest.approx(68.941910081020, rel=1e-12)


def test_host_code_under_sanitizers(tmp_path):
    """libljmd.so's HOST code built with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C csrc asan`; the
    device code cannot be instrumented on this pool) driven through every entry point that is reachable without a
    GPU: argument guards, the no-device failures of the single- and multi-device constructors and of the stateless
    entry points, error-text plumbing, NULL handles.  Any sanitizer report fails the child process."""
    import glob
    import os
    import subprocess
    import sys
    lib = ROOT / "molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd" / "csrc" / "obj" / "libljmd_asan.so"
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not lib.exists() or not rt:
        pytest.skip("sanitizer build absent: make -C .../csrc asan")
    script = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
import ljmd_amd
from ljmd_amd import _lib, md_types
lib = _lib.load()
assert "asan" in str(_lib.os.environ["LJMD_LIBRARY"])
h = C.c_void_p()
for args in ((0, 10.0, 0.005, 2.0), (10, -1.0, 0.005, 2.0), (10, 10.0, 0.005, 5.0), (10, 10.0, 0.0, 2.0)):
    assert lib.ljmd_create(C.byref(h), *args, 0, 0, 0, 1) == _lib.LJMD_ERR_INVALID_ARG and _lib.last_error()
assert lib.ljmd_create(C.byref(h), 10, 10.0, 0.005, 2.0, 0, 0, 0, 3) == _lib.LJMD_ERR_INVALID_ARG
assert lib.ljmd_create(None, 10, 10.0, 0.005, 2.0, 0, 0, 0, 1) == _lib.LJMD_ERR_INVALID_ARG
devs = (C.c_int32 * 4)(0, 0, 0, 0)
nodev = lib.ljmd_device_count() == 0
rc = lib.ljmd_create(C.byref(h), 4096, 20.0, 0.005, 8.0, 0, 0, 0, 1)
assert rc == (_lib.LJMD_ERR_NO_DEVICE if nodev else 0)
if rc == 0:
    lib.ljmd_destroy(h)
rc = lib.ljmd_create_multi(C.byref(h), 4096, 20.0, 0.005, 8.0, 0, 4, devs)
assert rc == (_lib.LJMD_ERR_NO_DEVICE if nodev else 0), rc
if rc == 0:
    lib.ljmd_destroy(h)
assert lib.ljmd_create_multi(C.byref(h), 4096, 20.0, 0.005, 8.0, 0, 0, None) == _lib.LJMD_ERR_INVALID_ARG
assert lib.ljmd_create_multi(None, 4096, 20.0, 0.005, 8.0, 0, 2, None) == _lib.LJMD_ERR_INVALID_ARG
p = md_types.init_params(108, 5.129927840030091, 0.005, 0.49 * 5.129927840030091)
st = md_types.init_state(p)
if nodev:
    for fn in (ljmd_amd.compute_lj_potential_energy, ljmd_amd.verlet_step):
        try:
            fn(p, st)
            raise SystemExit("expected LJMD_ERR_NO_DEVICE")
        except ljmd_amd.LjmdError as e:
            assert e.code == _lib.LJMD_ERR_NO_DEVICE
    hist = (C.c_uint64 * 8)()
    x = np.zeros(16)
    dp = C.POINTER(C.c_double)
    assert lib.ljmd_rdf_histogram(16, x.ctypes.data_as(dp), x.ctypes.data_as(dp), x.ctypes.data_as(dp), 10.0, 8, 4.0, hist) == _lib.LJMD_ERR_NO_DEVICE
for fn in ("ljmd_set_state", "ljmd_get_state"):
    pass
assert lib.ljmd_compute_forces(None, None, None, None) == _lib.LJMD_ERR_INVALID_ARG
assert lib.ljmd_verlet_steps(None, 1, None, None, None, None) == _lib.LJMD_ERR_INVALID_ARG
assert lib.ljmd_snapshot_begin(None) == _lib.LJMD_ERR_INVALID_ARG
assert lib.ljmd_comm_size(None) == 0
assert lib.ljmd_comm_unique_id(None) == _lib.LJMD_ERR_INVALID_ARG
lib.ljmd_destroy(None)
lib.ljmd_stateless_reset()
print("sanitized host code: ok")
""" % str(ROOT)
    env = dict(os.environ, LD_PRELOAD=rt[-1], LJMD_LIBRARY=str(lib),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=24")
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "sanitized host code: ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr
