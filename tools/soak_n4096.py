import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import ljmd_amd
from ljmd_amd import Engine, synthetic
p, r, v = synthetic.make_config(4096)
with Engine(p) as eng:
    eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
    e0 = eng.compute_forces()[0] + eng.kinetic_energy()
    t0 = time.perf_counter()
    means = []
    for blk in range(40):
        e, k, d, dd = eng.verlet_steps(5000)
        assert np.all(np.isfinite(e + k))
        means.append((e + k).mean())
    dt = time.perf_counter() - t0
    vel = np.stack(eng.get_state(("v",))["v"])
print(f"200000 steps at n=4096: {200000 / dt:.0f} steps/s; Etot(0) {e0:.6f}; block means min {min(means):.6f} max {max(means):.6f} "
      f"(rel range {(max(means) - min(means)) / abs(e0):.2e}); |sum v| per particle {np.abs(vel.sum(axis=1)).max() / 4096:.2e}")
