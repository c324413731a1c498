"""One dam-break step (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
pos, vel, p = scenes.dam_break(int(sys.argv[1]))
p.flags = SPH_FLAG_SURFACE_TENSION
sim = capi.Simulation(p, n_max=pos.shape[0]); sim.upload(pos, vel); sim.step(int(sys.argv[2])); sim.sync(); sim.close()
