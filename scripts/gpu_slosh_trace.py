"""Development aid (GPU): the sloshing tank on K slabs; particles and work per slab every 20 steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import scenes, slab, slab_local
side, nslabs, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = sys.argv[4] if len(sys.argv) > 4 else "sloshing"
p, backends, cuts = slab_local.build(scene, side, nslabs, n_max_factor=6.0)
slab.halo_local_slabs(backends)
for s in range(steps):
    if scene == "sloshing":
        g = scenes.sloshing_gravity(s)
        for b in backends:
            q = b.params.copy()
            q.gravity[0], q.gravity[1], q.gravity[2] = float(g[0]), float(g[1]), float(g[2])
            b.set_params(q)
    slab.step_local_slabs_overlapped(backends)
    if s % 2 == 1:
        st = {}
        cuts = slab.rebalance_next_local_slabs(backends, cuts, stats=st)
        if s % 20 == 19:
            own = [b.last_own() for b in backends]
            load = st["load"]
            print(json.dumps({"step": s + 1, "cuts": [int(c) for c in cuts], "own": own,
                              "own_worst": round(max(own) * nslabs / sum(own), 2),
                              "work_worst": round(max(load) * nslabs / sum(load), 2), "shift": round(st["shift"], 2),
                              "lost": sum(b.lost() for b in backends), "missed": sum(b.missed() for b in backends)}), flush=True)
