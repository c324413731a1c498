"""Step-time evolution of the dam-break scene (development aid)."""
import sys, time, os
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch; torch.cuda.init()
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, SPH_FLAG_KEEP_DEBUG
pos, vel, p = scenes.dam_break(100)
p.flags = SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG
n = pos.shape[0]
sim = capi.Simulation(p, n_max=n); sim.upload(pos, vel); sim.enable_timing(True)
for w in range(12):
    sim.reset_timings(); t=time.time(); sim.step(20); sim.sync(); dt=time.time()-t
    tm = sim.timings(); rho, f, nc, cell = sim.download_debug()
    gp, gv = sim.download()
    occ = np.bincount(cell); 
    print("steps %3d-%3d ms/step %.3f force %.3f dens %.3f | ncount mean %.1f max %d | cell occ mean(nonempty) %.2f max %d | vmax %.2f ymean %.3f" % (w*20, w*20+20, dt/20*1e3, tm['force_ms']/20, tm['density_ms']/20, nc.mean(), nc.max(), occ[occ>0].mean(), occ.max(), np.abs(gv).max(), gp[:,1].mean()))
