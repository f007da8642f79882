"""Parser of inputs/input_simulation_parameters.txt -- same grammar and the same
validation as read_simulation_parameters (scripts/base/read_input_files.f90:27-173).

Grammar (read_input_files.f90:87-159): blank lines and lines whose FIRST column is '#'
are skipped; every other line is tried as the next missing numeric block with a
list-directed read -- block 1 = 4 integers (k, total_steps, output_interval,
warmup_steps), block 2 = 3 reals (dt, L, rc_over_L), block 3 = 1 real
(target_total_energy); a line that fails the read is silently skipped (that is how the
header-word lines are ignored).  Fortran `d` exponents (1.d-4) are accepted.
N = 4 k^3 (:168); rc = rc_over_L * L (:171).
"""
from __future__ import annotations

import re
from dataclasses import dataclass

from .md_types import SimParams, init_params

_INT_RE = re.compile(r"^[+-]?\d+$")
_REAL_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eEdD][+-]?\d+)?$")


@dataclass
class RunControl:
    params: SimParams
    total_steps: int
    output_interval: int
    warmup_steps: int
    rc_over_L: float
    target_total_energy: float


def _tokens(line: str):
    # list-directed input: blanks or commas separate items; a '/' ends the record
    line = line.split("/")[0]
    return [t for t in re.split(r"[,\s]+", line.strip()) if t]


def _as_int(tok: str):
    return int(tok) if _INT_RE.match(tok) else None


def _as_real(tok: str):
    if not _REAL_RE.match(tok):
        return None
    return float(tok.replace("d", "e").replace("D", "e"))


def parse_simulation_parameters(text: str) -> RunControl:
    got1 = got2 = got3 = False
    k = total = oi = warm = 0
    dt = L = rcl = etarget = 0.0
    for line in text.splitlines():
        if len(line.strip()) == 0 or line[0] == "#":
            continue
        toks = _tokens(line)
        if not got1:
            vals = [_as_int(t) for t in toks[:4]]
            if len(vals) == 4 and all(v is not None for v in vals):
                k, total, oi, warm = vals
                if k <= 0:
                    raise ValueError("read_simulation_parameters(): k must be > 0.")
                if total <= 0:
                    raise ValueError("read_simulation_parameters(): total_steps must be > 0.")
                if oi <= 0:
                    raise ValueError("read_simulation_parameters(): output_interval must be > 0.")
                if warm < 0:
                    raise ValueError("read_simulation_parameters(): warmup_steps must be >= 0.")
                got1 = True
            continue
        if not got2:
            vals = [_as_real(t) for t in toks[:3]]
            if len(vals) == 3 and all(v is not None for v in vals):
                dt, L, rcl = vals
                if dt <= 0.0:
                    raise ValueError("read_simulation_parameters(): dt must be > 0.")
                if L <= 0.0:
                    raise ValueError("read_simulation_parameters(): L must be > 0.")
                if rcl <= 0.0:
                    raise ValueError("read_simulation_parameters(): rc_over_L must be > 0.")
                if rcl > 0.5:
                    raise ValueError("read_simulation_parameters(): rc_over_L must be <= 0.5 (minimum image).")
                got2 = True
            continue
        if not got3:
            v = _as_real(toks[0]) if toks else None
            if v is not None:
                etarget = v
                got3 = True
                break
            continue
    if not got1:
        raise ValueError("read_simulation_parameters(): missing Block 1 numeric line.")
    if not got2:
        raise ValueError("read_simulation_parameters(): missing Block 2 numeric line.")
    if not got3:
        raise ValueError("read_simulation_parameters(): missing Block 3 numeric line.")
    n = 4 * k * k * k
    params = init_params(n, L, dt, rcl * L, num_cells=k)
    return RunControl(params, total, oi, warm, rcl, etarget)


def read_simulation_parameters(filename) -> RunControl:
    try:
        with open(filename, "r") as f:
            text = f.read()
    except OSError as exc:
        raise ValueError("read_simulation_parameters(): cannot open input file.") from exc
    return parse_simulation_parameters(text)
