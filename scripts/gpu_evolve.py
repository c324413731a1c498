"""Step-time evolution of the dam-break scene, per sweep flavour (development aid)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, SPH_FLAG_KEEP_DEBUG, SPH_FLAG_LANE_SWEEP
side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pos, vel, p = scenes.dam_break(side)
n = pos.shape[0]
only = sys.argv[3] if len(sys.argv) > 3 else None
for name, extra in (("masked", 0), ("lane", SPH_FLAG_LANE_SWEEP)):
    if only and name != only: continue
    p.flags = SPH_FLAG_SURFACE_TENSION | extra
    sim = capi.Simulation(p, n_max=n); sim.upload(pos, vel); sim.enable_timing(True)
    for w in range(windows):
        sim.reset_timings(); t=time.time(); sim.step(20); sim.sync(); dt=time.time()-t
        tm = sim.timings()
        print("%s steps %3d-%3d ms/step %.3f | hash %.3f scan %.3f reorder %.3f search %.3f dens %.3f force %.3f" % (name, w*20, w*20+20, dt/20*1e3, tm['hash_ms']/20, tm['scan_ms']/20, tm['reorder_ms']/20, tm['search_ms']/20, tm['density_ms']/20, tm['force_ms']/20))
    sim.close()
