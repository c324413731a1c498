"""Cost of the slab machinery without any transport: k slabs of one dam break, all on ONE GPU in one process,
stepped one after the other, against the same scene as a single domain (development aid)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph_amd import capi, scenes, slab
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pos, vel, p = scenes.dam_break(side)
p.flags = SPH_FLAG_SURFACE_TENSION
n = pos.shape[0]
sim = capi.Simulation(p, n_max=n); sim.upload(pos, vel); sim.step(10); sim.sync()
t = time.perf_counter(); sim.step(steps); sim.sync(); t1 = time.perf_counter() - t
print("single domain: %.3f ms/step" % (t1 / steps * 1e3), flush=True); sim.close()
gn = capi.global_cells(p)
for k in (1, 2, 4):
    cellw = np.float32(p.h) * np.float32(1.000244140625)
    col = np.minimum((pos[:, 0] / cellw).astype(np.int64), gn - 1)
    cuts = slab.partition_columns(np.bincount(col, minlength=gn), k)
    sims = []
    for r in range(k):
        m = (col >= cuts[r]) & (col < cuts[r + 1])
        s = capi.SlabSimulation(p, n_max=int(2.5 * n / k) + 100000, col_lo=cuts[r], col_hi=cuts[r + 1], cols_capacity=gn)
        s.upload(pos[m], vel[m], np.nonzero(m)[0].astype(np.int32)); sims.append(s)
    ex = slab.LocalExchange(k)
    for _ in range(10): slab.step_local_slabs(sims, ex)
    for s in sims: s.sync()
    t = time.perf_counter()
    for _ in range(steps): slab.step_local_slabs(sims, ex)
    for s in sims: s.sync()
    tk = time.perf_counter() - t
    print("%d slabs on one GPU: %.3f ms/step for all (%.2fx the single domain), particles per slab %s" % (
        k, tk / steps * 1e3, tk / t1, [s.n for s in sims]), flush=True)
    for s in sims: s.close()
