"""Step rate at small particle counts, where a step is launch-bound (development aid)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, default_params
for n in (3000, 30000, 100000):
    p = default_params(flags=SPH_FLAG_SURFACE_TENSION)
    pos, vel = scenes.random_block(n, tank_size=float(p.tank_size), seed=1)
    sim = capi.Simulation(p, n_max=n); sim.upload(pos, vel); sim.step(200); sim.sync()
    for chunk in (1, 1000):
        t = time.perf_counter()
        for _ in range(2000 // chunk): sim.step(chunk)
        sim.sync(); dt = time.perf_counter() - t
        print("n %6d  sph_step(%4d) x %4d: %.1f us/step" % (n, chunk, 2000 // chunk, dt / 2000 * 1e6), flush=True)
    sim.close()
