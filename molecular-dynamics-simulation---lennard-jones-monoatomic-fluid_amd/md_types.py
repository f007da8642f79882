"""Host-side mirror of the reference's parameter/state containers.

  type(sim_params), init_params, compute_derived_params   scripts/base/md_types.f90:27-50,105-169
  type(sim_state),  init_state                            scripts/base/md_types.f90:56-60,175-230

Same field names, same derived-constant expressions, same guards (the reference
`stop`s with a message; here the same message is raised as ValueError).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PI = 3.1415926535897932384626433832795  # md_types.f90:22


@dataclass
class SimParams:
    n: int = 0
    num_cells: int = 0
    box_length: float = 0.0
    inv_box_length: float = 0.0
    volume: float = 0.0
    density: float = 0.0
    dt: float = 0.0
    dt_half: float = 0.0
    dt_square_half: float = 0.0
    rc: float = 0.0
    rc_square: float = 0.0


def compute_derived_params(p: SimParams) -> None:
    """md_types.f90:132-169, expression for expression."""
    if p.box_length > 0.0:
        p.inv_box_length = 1.0 / p.box_length
        p.volume = p.box_length * p.box_length * p.box_length
        if p.n > 0:
            p.density = p.n / p.volume
    else:
        raise ValueError("compute_derived_params(): box_length must be > 0.")
    if p.rc > 0.0:
        p.rc_square = p.rc * p.rc
    else:
        raise ValueError("compute_derived_params(): rc (cutoff_radius) must be > 0.")
    if p.rc >= 0.5 * p.box_length:
        raise ValueError("compute_derived_params(): rc (cutoff_radius) must be < L/2 (minimum image convention).")
    if p.dt > 0.0:
        p.dt_half = 0.5 * p.dt
        p.dt_square_half = p.dt_half * p.dt
    else:
        raise ValueError("compute_derived_params(): dt must be > 0.")


def init_params(n: int, box_length: float, dt: float, rc: float, num_cells: int = 0) -> SimParams:
    """md_types.f90:105-120."""
    p = SimParams(n=int(n), box_length=float(box_length), dt=float(dt), rc=float(rc),
                  num_cells=int(num_cells))
    compute_derived_params(p)
    return p


@dataclass
class SimState:
    rx: np.ndarray = field(default=None)
    ry: np.ndarray = field(default=None)
    rz: np.ndarray = field(default=None)
    vx: np.ndarray = field(default=None)
    vy: np.ndarray = field(default=None)
    vz: np.ndarray = field(default=None)
    ax: np.ndarray = field(default=None)
    ay: np.ndarray = field(default=None)
    az: np.ndarray = field(default=None)

    FIELDS = ("rx", "ry", "rz", "vx", "vy", "vz", "ax", "ay", "az")

    def allocated(self) -> bool:
        return self.rx is not None

    def copy(self) -> "SimState":
        return SimState(**{k: getattr(self, k).copy() for k in self.FIELDS})


def init_state(p: SimParams) -> SimState:
    """allocate_state + zero_state, md_types.f90:175-230."""
    if p.n <= 0:
        raise ValueError("allocate_state(): params%n must be > 0.")
    return SimState(**{k: np.zeros(p.n, dtype=np.float64) for k in SimState.FIELDS})
