"""Development/measurement aid (GPU): K slabs of the dam break in one process, records moved by the library.
   python scripts/gpu_slab_overlap.py <side> <slabs> <steps> [rebalance_every]
Prints ms/step of the slab run and of the single context on the same field (the cost of the slab machinery when
transport is a device copy).  Under `rocprofv3 --hip-trace --stats` the API summary shows the host waits per step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, slab, slab_local

side, nslabs, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p, backends, cuts = slab_local.build("dam_break", side, nslabs)
slab.halo_local_slabs(backends)
for _ in range(5):
    slab.step_local_slabs_overlapped(backends)
for b in backends:
    b.sync()
t0 = time.perf_counter()
for s in range(steps):
    slab.step_local_slabs_overlapped(backends)
    if every and s % every == every - 1:
        cuts = slab.rebalance_next_local_slabs(backends, cuts)
for b in backends:
    b.sync()
dt_slab = (time.perf_counter() - t0) / steps
own = [b.last_own() for b in backends]
bad = sum(b.lost() + b.missed() for b in backends)
for b in backends:
    b.close()
with capi.Simulation(p, n_max=side ** 3) as single:
    single.fill_lattice(side, side, side)
    single.step(5)
    single.sync()
    t0 = time.perf_counter()
    single.step(steps)
    single.sync()
    dt_one = (time.perf_counter() - t0) / steps
print(json.dumps({"particles": side ** 3, "slabs": nslabs, "steps": steps, "rebalance_every": every,
                  "slab_ms_per_step": dt_slab * 1e3, "single_context_ms_per_step": dt_one * 1e3,
                  "ratio": dt_slab / dt_one, "own": own, "lost_or_missed": bad}))
