"""Development aid (GPU): 3000 steps of the 1M dam break; positions stay finite and inside the tank."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
pos, vel, p = scenes.dam_break(100)
p.flags = SPH_FLAG_SURFACE_TENSION
with capi.Simulation(p, n_max=pos.shape[0]) as sim:
    sim.upload(pos, vel)
    for k in range(6):
        sim.step(500)
        gp, gv = sim.download()
        print(k, np.isfinite(gp).all(), float(gp.min()), float(gp.max()), float(np.abs(gv).max()), flush=True)
