"""Quick on-GPU sanity run: HIP path vs CPU oracle (development aid, not a test)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import default_params, SPH_FLAG_SURFACE_TENSION, SPH_FLAG_EXACT_ORDER, SPH_FLAG_KEEP_DEBUG
from oracle import Oracle

o = Oracle()
def beq(a, b): return np.array_equal(a.view(np.uint32), b.view(np.uint32))
for n in (1, 2, 64, 3000):
    pos, vel = scenes.random_block(n, seed=n)
    for exact in (True, False):
        p = default_params(flags=SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG | (SPH_FLAG_EXACT_ORDER if exact else 0))
        sim = capi.Simulation(p, n_max=n)
        sim.upload(pos, vel)
        po, vo = pos.copy(), vel.copy()
        for s in range(10):
            sim.step(1)
            gp, gv = sim.download()
            rho, f, nc, cell = sim.download_debug()
            nc_o = o.neighbor_counts(p, po); cell_o = o.cell_ids(p, po)
            ro, fo = o.step_allpairs(p, po, vo, True)
            if exact:
                print(n, "exact", s, "rho", beq(rho, ro), "F", beq(f, fo), "pos", beq(gp, po), "vel", beq(gv, vo), "nc", np.array_equal(nc, nc_o), "cell", np.array_equal(cell, cell_o))
            else:
                print(n, "fast ", s, "rho %.2e F %.2e pos %.2e vel %.2e" % (np.abs(rho-ro).max()/np.abs(ro).max(), np.abs(f-fo).max()/np.abs(fo).max(), np.abs(gp-po).max()/np.abs(po).max(), np.abs(gv-vo).max()/max(np.abs(vo).max(),1e-30)), "nc", np.array_equal(nc, nc_o), "cell", np.array_equal(cell, cell_o))
        sim.close()
# throughput glance
for side in (100,):
    pos, vel, p = scenes.dam_break(side)
    n = pos.shape[0]
    sim = capi.Simulation(p, n_max=n)
    sim.upload(pos, vel)
    sim.step(5); sim.sync()
    sim.enable_timing(True)
    t = time.time(); sim.step(20); sim.sync(); dt = time.time() - t
    print("dam_break", n, "ms/step", dt/20*1e3, "pu/s %.3e" % (n*20/dt), sim.timings(), sim.grid_dims())
    sim.close()
