"""Short run for counter collection: dam break, a few steps at a chosen phase (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, SPH_FLAG_LANE_SWEEP, SPH_FLAG_REFERENCE_ARITH
side, pre, steps, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
pos, vel, p = scenes.dam_break(side)
p.flags = SPH_FLAG_SURFACE_TENSION | {"lane": SPH_FLAG_LANE_SWEEP, "ref": SPH_FLAG_REFERENCE_ARITH}.get(mode, 0)
sim = capi.Simulation(p, n_max=pos.shape[0]); sim.upload(pos, vel)
sim.step(pre + steps); sim.sync(); sim.close()
