"""Development/measurement aid (GPU): K slabs of the dam break in one process; how evenly is the TIME of a step spread
over the slabs when the cut planes balance particle counts, and when they balance work (sum of 9 + neighbours)?
   python scripts/gpu_balance_work.py <side> <slabs> <steps> <count|work>
Per-slab device time = hash + scan + reorder + search + force of the slab's own kernels (HIP events), summed over the
last 20 steps; prints worst / mean."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import slab, slab_local

side, nslabs, steps, by = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
p, backends, cuts = slab_local.build("dam_break", side, nslabs)
slab.halo_local_slabs(backends)
window = 20
wall = [0.0] * nslabs  # host clock around begin + sync of each slab in the window (the kernels' brackets overlap in time
                       # where a slab runs its face and interior columns side by side)
for s in range(steps):
    if s == steps - window:
        for b in backends:
            b.enable_timing(True)
            b.reset_timings()
    if s >= steps - window:  # one slab at a time, so that no other slab's kernels share the GPU with the timed ones
        for k, b in enumerate(backends):
            t0 = time.perf_counter()
            b.step_begin()
            b.sync()
            wall[k] += time.perf_counter() - t0
        for b in backends:
            b.step_end()
    else:
        slab.step_local_slabs_overlapped(backends)
    if s % 2 == 1:
        cuts = slab.rebalance_next_local_slabs(backends, cuts, by=by)
for b in backends:
    b.sync()
t = []
cats = {}
for b in backends:
    tm = b.timings()
    for k, v in tm.items():
        if k.endswith("_ms"):
            cats[k] = cats.get(k, 0.0) + v / window
    t.append(tm["hash_ms"] + tm["scan_ms"] + tm["reorder_ms"] + tm["search_ms"] + tm["density_ms"] + tm["force_ms"])
own = [b.last_own() for b in backends]
bad = {"lost": sum(b.lost() for b in backends), "missed": sum(b.missed() for b in backends)}
t = np.array(t) / window
print(json.dumps({"balance_by": by, "particles": side ** 3, "slabs": nslabs, "steps": steps,
                  "slab_ms_per_step": [round(float(v), 3) for v in t],
                  "slab_wall_ms_per_step": [round(w / window * 1e3, 3) for w in wall], "worst_over_mean_time": float(t.max() / t.mean()),
                  "sum_over_slabs_ms_per_step": {k: round(v, 4) for k, v in cats.items()}, "own": own, "worst_over_mean_particles": float(max(own) / np.mean(own)), "cuts": [int(c) for c in cuts],
                  "lost_or_missed": bad}))
for b in backends:
    b.close()
