import os, sys, time
sys.path.insert(0, '/root/repo')
import ljmd_amd
from ljmd_amd import Engine, synthetic
for n in (65536, 262144):
    p, r, v = synthetic.make_config(n)
    os.environ["LJMD_MULTI_MIGRATE_EVERY"] = "10"
    with Engine(p, devices=[0] * 8) as eng:
        eng.set_state(r[0], r[1], r[2], v[0], v[1], v[2])
        eng.compute_forces()
        eng.advance(10)            # steps_since_migration = 10 afterwards
        eng.synchronize()
        t0 = time.perf_counter(); eng.advance(1); eng.synchronize(); t1 = time.perf_counter()   # migration + 1 step
        eng.advance(1); eng.synchronize(); t2 = time.perf_counter()                               # 1 step
        print(f"n={n}: migration + 1 step {1e3*(t1-t0):.1f} ms, 1 step {1e3*(t2-t1):.1f} ms, migrations {eng.migrations()}", flush=True)
