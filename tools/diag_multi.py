import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ljmd_amd
from ljmd_amd import Engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p, r, v = synthetic.make_config(n)
with Engine(p) as one:
    one.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    one.compute_forces()
    s1 = np.stack(one.verlet_steps(steps), axis=1)
    v1 = np.stack(one.get_state(("v",))["v"])
with Engine(p, devices=[0]*8) as m:
    m.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    m.compute_forces()
    s8 = np.stack(m.verlet_steps(steps), axis=1)
    v8 = np.stack(m.get_state(("v",))["v"])
S = n // 8
print(n, steps, "scalars rel", np.abs(s8 - s1).max(axis=0) / np.abs(s1).max(axis=0), "max|dv|", np.abs(v8-v1).max(),
      "per-rank max|dv|", [float(np.abs(v8[:, g*S:(g+1)*S]-v1[:, g*S:(g+1)*S]).max()) for g in range(8)],
      "sum v8", np.abs(v8.sum(axis=1)).max(), "sum v1", np.abs(v1.sum(axis=1)).max(), flush=True)
