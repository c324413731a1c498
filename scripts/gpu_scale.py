"""Throughput at larger particle counts on one GPU (development aid): dam break and uniform cube."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
for name, side in (("dam_break", 100), ("dam_break", 200), ("uniform_cube", 128), ("uniform_cube", 256)):
    t = time.time()
    pos, vel, p = getattr(scenes, name)(side)
    p.flags = SPH_FLAG_SURFACE_TENSION
    n = pos.shape[0]
    tg = time.time() - t
    sim = capi.Simulation(p, n_max=n); sim.upload(pos, vel); sim.enable_timing(True)
    sim.step(20); sim.sync(); sim.reset_timings()
    t = time.time(); steps = 100; sim.step(steps); sim.sync(); dt = time.time() - t
    tm = sim.timings()
    gp, gv = sim.download()
    print("%s side %d n %d grid %s gen %.1fs | ms/step %.3f pu/s %.3e | hash %.3f scan %.3f reorder %.3f search %.3f dens %.3f force %.3f | finite %s" % (
        name, side, n, sim.grid_dims(), tg, dt/steps*1e3, n*steps/dt, tm['hash_ms']/steps, tm['scan_ms']/steps, tm['reorder_ms']/steps,
        tm['search_ms']/steps, tm['density_ms']/steps, tm['force_ms']/steps, bool(np.isfinite(gp).all())), flush=True)
    sim.close()
