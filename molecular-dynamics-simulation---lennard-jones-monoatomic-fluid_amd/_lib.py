"""ctypes binding of libljmd.so -- the C ABI declared in include/ljmd.h.

There is no fallback: if the shared library is missing this module raises at import
of the first symbol, and if no HIP device is present every compute call raises
LjmdError(LJMD_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
LIB_PATH = PKG_DIR / "libljmd.so"

LJMD_OK = 0
LJMD_ERR_INVALID_ARG = -1
LJMD_ERR_NO_DEVICE = -2
LJMD_ERR_HIP = -3
LJMD_ERR_STATE = -4
LJMD_ERR_ALLOC = -5

PRECISION_FP64 = 0
PRECISION_FP32_FORCE = 1

R, RU, V, A = 0, 1, 2, 3
PARTIAL_STRIDE = 8
COMM_ID_BYTES = 128

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class LjmdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ljmd: {message} (status {code})")
        self.code = code
        self.message = message


# every exported symbol of include/ljmd.h: name -> (restype, argtypes)
PROTOTYPES = {
    "ljmd_version": (C.c_char_p, []),
    "ljmd_device_count": (C.c_int32, []),
    "ljmd_last_error": (C.c_char_p, [C.c_void_p]),
    "ljmd_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_double, C.c_double, C.c_double,
                              C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ljmd_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32,
                                    C.c_int32, c_int32_p]),
    "ljmd_destroy": (None, [C.c_void_p]),
    "ljmd_set_state": (C.c_int, [C.c_void_p] + [c_double_p] * 6),
    "ljmd_set_accel": (C.c_int, [C.c_void_p] + [c_double_p] * 3),
    "ljmd_set_unwrapped": (C.c_int, [C.c_void_p] + [c_double_p] * 3),
    "ljmd_get_state": (C.c_int, [C.c_void_p] + [c_double_p] * 12),
    "ljmd_compute_forces": (C.c_int, [C.c_void_p] + [c_double_p] * 3),
    "ljmd_verlet_steps": (C.c_int, [C.c_void_p, C.c_int32] + [c_double_p] * 4),
    "ljmd_multi_migrations": (C.c_int32, [C.c_void_p]),
    "ljmd_migrate": (C.c_int, [C.c_void_p]),
    "ljmd_migrate_pack": (C.c_int, [C.c_void_p]),
    "ljmd_migrate_buffer": (C.c_void_p, [C.c_void_p, c_int64_p, c_int64_p, c_int64_p]),
    "ljmd_migrate_deal": (C.c_int, [C.c_void_p]),
    "ljmd_particle_ids": (C.c_int, [C.c_void_p, c_int32_p]),
    "ljmd_enqueue_steps": (C.c_int, [C.c_void_p, C.c_int32]),
    "ljmd_enqueue_steps_sampled": (C.c_int, [C.c_void_p, C.c_int32]),
    "ljmd_set_observables": (C.c_int, [C.c_void_p, C.c_int32]),
    "ljmd_collect_steps": (C.c_int, [C.c_void_p, C.c_int32] + [c_double_p] * 4),
    "ljmd_snapshot_begin": (C.c_int, [C.c_void_p]),
    "ljmd_snapshot_end": (C.c_int, [C.c_void_p] + [c_double_p] * 12),
    "ljmd_kinetic_energy": (C.c_int, [C.c_void_p, c_double_p]),
    "ljmd_compute_lj_potential_energy": (C.c_int, [C.c_int32, C.c_double, C.c_double]
                                         + [c_double_p] * 9),
    "ljmd_verlet_step": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double] + [c_double_p] * 13),
    "ljmd_stateless_reset": (None, []),
    "ljmd_rdf_histogram": (C.c_int, [C.c_int32, c_double_p, c_double_p, c_double_p, C.c_double, C.c_int32,
                                     C.c_double, C.POINTER(C.c_uint64)]),
    "ljmd_time_origin_average": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int32,
                                           C.c_int32, c_double_p]),
    "ljmd_shard_range": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "ljmd_exchange_buffer": (C.c_void_p, [C.c_void_p, c_int64_p, c_int64_p, c_int64_p]),
    "ljmd_device_ptr": (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32]),
    "ljmd_stream": (C.c_void_p, [C.c_void_p]),
    "ljmd_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ljmd_comm_init": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ljmd_allgather_positions": (C.c_int, [C.c_void_p]),
    "ljmd_comm_size": (C.c_int32, [C.c_void_p]),
    "ljmd_synchronize": (C.c_int, [C.c_void_p]),
    "ljmd_memcpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ljmd_step_begin": (C.c_int, [C.c_void_p]),
    "ljmd_step_finish": (C.c_int, [C.c_void_p]),
    "ljmd_forces_partial": (C.c_int, [C.c_void_p]),
    "ljmd_step_forces": (C.c_int, [C.c_void_p]),
    "ljmd_force_buffers": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), c_int64_p,
                                     C.POINTER(C.c_void_p), c_int64_p]),
    "ljmd_read_partials": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "ljmd_combine_scalars": (C.c_int, [C.c_void_p, c_double_p, C.c_int32] + [c_double_p] * 4),
    "ljmd_set_tail_corrections": (C.c_int, [C.c_void_p, C.c_int32]),
    "ljmd_stateless_set_tail_corrections": (None, [C.c_int32]),
    "ljmd_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "ljmd_pair_kernel_name": (C.c_char_p, [C.c_void_p]),
    "ljmd_profile_read": (C.c_int, [C.c_void_p, c_double_p, c_int32_p]),
    "ljmd_profile_read_ex": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int32_p]),
    "ljmd_profile_read_rank": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_int32_p]),
    "ljmd_profile_read_stats": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p, c_double_p, c_int32_p]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libljmd.so (once) and applies the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("LJMD_LIBRARY", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: build it with `make -C {PKG_DIR / 'csrc'}` or "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().ljmd_last_error(handle)
    return msg.decode() if msg else ""


def check(status: int, handle=None) -> None:
    if status != LJMD_OK:
        raise LjmdError(status, last_error(handle) or last_error(None))
