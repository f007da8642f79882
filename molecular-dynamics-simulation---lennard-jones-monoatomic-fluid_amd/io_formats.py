"""Binary / text formats on the edges of the hot path.

  outputs/rv_init.dat                       md_initial_config_program.f90:275-290 (write)
                                            md_simulation_program.f90:573-588     (read)
  outputs/one_run/rva.dat                   md_simulation_program.f90:254-257 (header), :384-387 (records)
  outputs/one_run/instantaneous_energies.dat   md_simulation_program.f90:294 (header), :374 (rows)

Fortran unformatted sequential records as written by gfortran / flang: a 4-byte
little-endian byte count before and after every record payload.
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

ENERGIES_HEADER = "# time   epot   ekin   etot   T   P"


def _write_record(f, payload: bytes) -> None:
    mark = struct.pack("<i", len(payload))
    f.write(mark)
    f.write(payload)
    f.write(mark)


def _read_record(f) -> bytes:
    head = f.read(4)
    if len(head) < 4:
        raise EOFError("end of unformatted file")
    (nbytes,) = struct.unpack("<i", head)
    payload = f.read(nbytes)
    tail = f.read(4)
    if len(payload) != nbytes or tail != head:
        raise ValueError("corrupt Fortran unformatted record")
    return payload


def _pack3(x, y, z) -> bytes:
    return np.concatenate([np.asarray(a, dtype="<f8").ravel() for a in (x, y, z)]).tobytes()


def write_rv_init(path, rx, ry, rz, vx, vy, vz) -> None:
    with open(path, "wb") as f:
        _write_record(f, _pack3(rx, ry, rz))
        _write_record(f, _pack3(vx, vy, vz))


def read_rv_init(path, n: int):
    """-> (r[3, n], v[3, n])"""
    with open(path, "rb") as f:
        try:
            r = np.frombuffer(_read_record(f), dtype="<f8")
            v = np.frombuffer(_read_record(f), dtype="<f8")
        except EOFError as exc:
            raise ValueError("read_rv_init(): truncated rv_init file.") from exc
    if r.size != 3 * n or v.size != 3 * n:
        raise ValueError(f"read_rv_init(): record holds {r.size // 3} particles, expected {n}.")
    return r.reshape(3, n).copy(), v.reshape(3, n).copy()


class RvaWriter:
    """rva.dat: header (n:int32, L:f8, dt:f8, output_interval:int32, n_snapshots:int32) then per
    sample four records r, ru, v, a, each x||y||z of n fp64."""

    def __init__(self, path, n: int, box_length: float, dt: float, output_interval: int, n_snapshots: int):
        self._f = open(path, "wb")
        self.n = n
        _write_record(self._f, struct.pack("<iddii", n, box_length, dt, output_interval, n_snapshots))

    def write_snapshot(self, r, ru, v, a) -> None:
        for trio in (r, ru, v, a):
            _write_record(self._f, _pack3(*trio))

    def close(self) -> None:
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_rva(path):
    """-> (header dict, snapshots[n_snap, 4, 3, n]) ; layout spec = md_one_run_analysis.py:345-397."""
    with open(path, "rb") as f:
        n, L, dt, oi, nsnap = struct.unpack("<iddii", _read_record(f))
        snaps = []
        while True:
            try:
                recs = [np.frombuffer(_read_record(f), dtype="<f8").reshape(3, n) for _ in range(4)]
            except EOFError:
                break
            snaps.append(np.stack(recs))
    header = dict(n=n, box_length=L, dt=dt, output_interval=oi, n_snapshots_expected=nsnap)
    return header, (np.stack(snaps) if snaps else np.empty((0, 4, 3, n)))


def fortran_1pe13_6(x: float) -> str:
    """One field of the edit descriptor 1pe13.6 (md_simulation_program.f90:374)."""
    s = f"{x:.6E}"
    mant, exp = s.split("E")
    e = int(exp)
    if abs(e) >= 100:                   # Fortran drops the 'E' for 3-digit exponents
        s = f"{mant}{e:+04d}"
    return s.rjust(13) if len(s) <= 13 else "*" * 13


def energies_row(time, epot, ekin, etot, temp, press) -> str:
    """'(1pe13.6,5(2x,1pe13.6))'"""
    return "  ".join(fortran_1pe13_6(v) for v in (time, epot, ekin, etot, temp, press))


def read_energies(path) -> np.ndarray:
    rows = [list(map(float, ln.split())) for ln in Path(path).read_text().splitlines()
            if ln.strip() and not ln.startswith("#")]
    return np.array(rows, dtype=np.float64).reshape(-1, 6)
