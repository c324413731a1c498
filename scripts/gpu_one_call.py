"""Development/measurement aid (GPU): the one-call entry (host arrays in, host arrays out: upload, one step, download)
timed per call, pageable numpy memory as a host program would hold it, or the same arrays page-locked in place
(sph_host_register, what the drop-in initParticles does).   python scripts/gpu_one_call.py <n> <calls> [pinned]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION

n, calls = int(sys.argv[1]), int(sys.argv[2])
if n == 3000:
    params = None                                   # the reference's own configuration
    pos, vel = capi.fill_random_block(n)
else:
    side = round(n ** (1.0 / 3.0))
    n = side ** 3
    pos, vel, params = scenes.dam_break(side)
    params.flags = SPH_FLAG_SURFACE_TENSION
pos = np.ascontiguousarray(pos, np.float32)
vel = np.ascontiguousarray(vel, np.float32)
pinned = len(sys.argv) > 3 and sys.argv[3] == "pinned"
resident = len(sys.argv) > 3 and sys.argv[3] == "resident"
if pinned:
    capi.host_register(pos)
    capi.host_register(vel)
if resident:
    capi.one_call_mode(1)
for _ in range(5):
    capi.update_particles(pos, vel, params)
t = []
for _ in range(calls):
    t0 = time.perf_counter()
    capi.update_particles(pos, vel, params)
    t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print(json.dumps({"particles": n, "host_memory": "page-locked in place" if pinned else "pageable",
                  "one_call_mode": 1 if resident else 0, "calls": calls, "ms_per_call_median": float(np.median(t)), "ms_per_call_mean": float(t.mean()),
                  "particle_updates_per_s": n / (float(np.median(t)) * 1e-3), "bytes_over_pcie_per_call": 48 * n}))
