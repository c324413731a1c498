"""Development aid (GPU): the 64M-particle dam break as eight slabs on one GPU with the force pass in one launch (0) and
split into faces and interior (1): what the split costs when nothing travels over a link."""
import sys, json, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import slab, slab_local
for mode in (0, 1):
    p, backends, cuts = slab_local.build("dam_break", 400, 8)
    for b in backends:
        b.set_overlap(mode)
    slab.halo_local_slabs(backends)
    for s in range(4):
        slab.step_local_slabs_overlapped(backends)
    for b in backends: b.sync()
    t0 = time.perf_counter()
    for s in range(16):
        slab.step_local_slabs_overlapped(backends)
    for b in backends: b.sync()
    dt = (time.perf_counter() - t0) / 16
    print(json.dumps({"overlap_mode": mode, "ms_per_step": dt * 1e3}), flush=True)
    for b in backends: b.close()
