"""Load balance of k slabs over a dam break (all slabs on one GPU, in-process exchange): worst slab's share of
the particles against the mean, for several rebalancing intervals (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes, slab
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
side, k, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pos, vel, p = scenes.dam_break(side)
p.flags = SPH_FLAG_SURFACE_TENSION
n = pos.shape[0]
gn = capi.global_cells(p)
cellw = np.float32(p.h) * np.float32(1.000244140625)
col = np.minimum((pos[:, 0] / cellw).astype(np.int64), gn - 1)
for every in (4, 2, 1):
    cuts = slab.partition_columns(np.bincount(col, minlength=gn), k)
    sims = []
    for r in range(k):
        m = (col >= cuts[r]) & (col < cuts[r + 1])
        s = capi.SlabSimulation(p, n_max=int(3.0 * n / k) + 200000, col_lo=cuts[r], col_hi=cuts[r + 1], cols_capacity=gn)
        s.upload(pos[m], vel[m], np.nonzero(m)[0].astype(np.int32)); sims.append(s)
    ex = slab.LocalExchange(k)
    worst, acc = 0.0, 0.0
    for t in range(steps):
        slab.step_local_slabs(sims, ex)
        for s in sims: s.sync()
        if (t + 1) % every == 0:
            cuts = slab.rebalance_local_slabs(sims, cuts)
        if (t + 1) % 10 == 0:
            share = max(s.n for s in sims) * k / n
            worst = max(worst, share); acc += share
    print("rebalance every %d: worst slab / mean = %.2f at worst, %.2f on average; final cuts %s; lost %d; max shift %.1f" % (
        every, worst, acc / (steps // 10), cuts, sum(s.lost() for s in sims), max(s.max_column_shift() for s in sims)), flush=True)
    for s in sims: s.close()
