"""Neighbour-count distribution of the dam break over time (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, SPH_FLAG_KEEP_DEBUG
pos, vel, p = scenes.dam_break(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
p.flags = SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG
sim = capi.Simulation(p, n_max=pos.shape[0]); sim.upload(pos, vel)
for k in range(0, 400, 10):
    sim.step(10)
    nc = sim.download_debug()[2]
    print("step %3d  mean %.1f  p50 %d p90 %d p99 %d p99.9 %d max %d  frac>64 %.4f >96 %.4f >128 %.4f >192 %.5f >256 %.5f" % (
        k + 10, nc.mean(), *np.percentile(nc, [50, 90, 99, 99.9]).astype(int), nc.max(),
        (nc > 64).mean(), (nc > 96).mean(), (nc > 128).mean(), (nc > 192).mean(), (nc > 256).mean()), flush=True)
