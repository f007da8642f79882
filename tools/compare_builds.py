#!/usr/bin/env python3
"""Bitwise comparison of two builds of libljmd.so on the same inputs (forces at n = 262144, a 25-step trajectory at
n = 20000 with a partially filled last tile, the sharded form, the mixed-precision mode): each build runs in its own child process and writes
its arrays; used to check that a refactoring of the pair kernel did not change a single bit.  usage:
compare_builds.py <libA.so> <libB.so>"""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import ljmd_amd
from ljmd_amd import Engine, synthetic
out = {}
p, r, v = synthetic.make_config(262144)
with Engine(p) as e:
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    out["sc262144"] = np.array(e.compute_forces()); out["a262144"] = np.stack(e.get_state(("a",))["a"])
p, r, v = synthetic.make_config(20000, seed=3)
with Engine(p) as e:
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2]); e.compute_forces()
    out["traj20000"] = np.stack(e.verlet_steps(25)); out["v20000"] = np.stack(e.get_state(("v",))["v"])
p, r, v = synthetic.make_config(65536, seed=5)
with Engine(p, devices=[0, 0, 0, 0]) as e:
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2]); e.compute_forces()
    out["multi65536"] = np.stack(e.verlet_steps(12))
p, r, v = synthetic.make_config(262144)
with Engine(p, precision_mode=1) as e:            # mixed: fp32 far kernel + fp64 near kernel
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    out["mixed_sc262144"] = np.array(e.compute_forces()); out["mixed_a262144"] = np.stack(e.get_state(("a",))["a"])
p, r, v = synthetic.make_config(20000, seed=3)
with Engine(p, precision_mode=1) as e:
    e.set_state(r[0], r[1], r[2], v[0], v[1], v[2]); e.compute_forces()
    out["mixed_traj20000"] = np.stack(e.verlet_steps(25))
np.savez(sys.argv[1], **out)
""" % str(ROOT)
res = []
with tempfile.TemporaryDirectory() as t:
    for k, lib in enumerate(sys.argv[1:3]):
        f = Path(t) / f"out{k}.npz"
        subprocess.run([sys.executable, "-c", CHILD, str(f)], check=True, env=dict(os.environ, LJMD_LIBRARY=str(Path(lib).resolve())))
        res.append(dict(np.load(f)))
ok = True
for key in res[0]:
    same = np.array_equal(res[0][key].view(np.uint64), res[1][key].view(np.uint64))
    ok = ok and same
    print(f"{key:12s} bitwise equal: {same}" + ("" if same else f"   max |diff| = {np.abs(res[0][key] - res[1][key]).max():.3e}"))
sys.exit(0 if ok else 1)
