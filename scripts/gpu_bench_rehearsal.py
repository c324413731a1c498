"""Development aid (GPU): the multi-GPU bench's window (pre-roll 60, warm-up 5, 20 steps) for N slabs with the bench's own
sizing rules (n_max, rec_capacity, cols_capacity of sph_amd/slab_bench.py), all slabs on ONE GPU over the local
transport: checks that no capacity is exceeded anywhere in the window, nothing is lost or missed, and how far the
largest slab strays from the mean.   python scripts/gpu_bench_rehearsal.py <N> [side_per_gpu]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import capi, scenes, slab
from sph_amd.params import SPH_FLAG_SURFACE_TENSION

world = int(sys.argv[1])
side = int(round((int(sys.argv[2]) if len(sys.argv) > 2 else 100) * world ** (1.0 / 3.0)))
p = scenes.dam_break_params(side)
p.flags = SPH_FLAG_SURFACE_TENSION
n_total = side ** 3
gn = capi.global_cells(p)
plane_x = (np.arange(side) + 0.5) * scenes.REST_SPACING
cell = np.float32(p.h) * np.float32(1.000244140625)
plane_col = np.minimum((plane_x / float(cell)).astype(np.int64), gn - 1)
hist = np.bincount(plane_col, minlength=gn) * side * side
cuts = slab.partition_columns(hist, world)
n_max = int(2.0 * n_total / world) + 4 * side * side * 8
rec_capacity = max(1 << 16, 48 * side * side)
backends = []
for r in range(world):
    mine = np.nonzero((plane_col >= cuts[r]) & (plane_col < cuts[r + 1]))[0]
    i0, i1 = (int(mine[0]), int(mine[-1]) + 1) if mine.size else (0, 0)
    b = capi.SlabSimulation(p, n_max=n_max, col_lo=cuts[r], col_hi=cuts[r + 1], cols_capacity=gn, rec_capacity=1024)
    b.fill_lattice(side, side, side, (i0, i1))
    backends.append(b)
for r, b in enumerate(backends):
    b.attach_local(backends[r - 1] if r > 0 else None, backends[r + 1] if r < world - 1 else None, rec_capacity)
slab.halo_local_slabs(backends)
worst_n, worst_own = 0, 0
for s in range(85):
    slab.step_local_slabs_overlapped(backends)
    if s % 2 == 1 and world > 1:
        cuts = slab.rebalance_next_local_slabs(backends, cuts)
    worst_n = max(worst_n, max(b.n for b in backends))
    worst_own = max(worst_own, max(b.last_own() for b in backends))
own = [b.last_own() for b in backends]
print(json.dumps({"slabs": world, "side": side, "particles": n_total, "n_max": n_max, "rec_capacity": rec_capacity,
                  "largest_particle_set_seen": worst_n, "largest_own_seen": worst_own, "mean_own": n_total / world,
                  "own_at_end": own, "lost": sum(b.lost() for b in backends), "missed": sum(b.missed() for b in backends),
                  "conserved": sum(own) == n_total, "cuts": [int(c) for c in cuts]}))
for b in backends:
    b.close()
