import sys, time
sys.path.insert(0, '/root/repo')
import ljmd_amd
from ljmd_amd import Engine, synthetic
for n in (1024, 4096, 16384):
    p, r, v = synthetic.make_config(n)
    with Engine(p) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.verlet_steps(100)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.enqueue_steps(2000)
        t1 = time.perf_counter()
        eng.collect_steps(2000)
        t2 = time.perf_counter()
        print(f"n={n}: enqueue 2000 steps: {1e6 * (t1 - t0) / 2000:.1f} us/step of CPU; until done: {1e6 * (t2 - t0) / 2000:.1f} us/step", flush=True)
