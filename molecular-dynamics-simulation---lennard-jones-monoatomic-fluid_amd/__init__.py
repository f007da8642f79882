"""ljmd_amd -- MI355X (gfx950) drop-in for the Lennard-Jones force/energy +
velocity-Verlet hot path of Ledicia/Molecular-Dynamics-Simulation---Lennard-Jones-monoatomic-fluid.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/ljmd.h) -> libljmd.so
  fortran/   ISO_C_BINDING shim modules with the reference's module/procedure names
             + the thin Fortran driver
  *.py       the host-side mirror of the reference interface used by tests and bench
"""
from ._lib import LjmdError, load as load_library  # noqa: F401
from .md_types import SimParams, SimState, init_params, init_state, compute_derived_params  # noqa: F401
from .physics import (Engine, compute_lj_potential_energy, verlet_step, minimum_image,  # noqa: F401
                      observables)
