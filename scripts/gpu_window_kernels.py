"""Measurement aid (GPU): one plain context, a scene generated in HBM, per-kernel device time over a window of steps,
and the mean neighbour count at the end of the window (the phase the window was in).
   python scripts/gpu_window_kernels.py <dam_break|uniform_cube> <side> <first_step> <steps> [reference]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_KEEP_DEBUG, SPH_FLAG_REFERENCE_ARITH, SPH_FLAG_SURFACE_TENSION, default_params

scene, side, first, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reference = len(sys.argv) > 5 and sys.argv[5] == "reference"
p = scenes.dam_break_params(side) if scene == "dam_break" else default_params(tank_size=side * scenes.REST_SPACING)
p.flags = SPH_FLAG_SURFACE_TENSION | (SPH_FLAG_REFERENCE_ARITH if reference else 0)
n = side ** 3
with capi.Simulation(p, n_max=n) as sim:
    sim.fill_lattice(side, side, side)
    sim.step(max(0, first - 1))
    if first > 0:  # neighbour counts of the step right before the window
        q = p.copy()
        q.flags |= SPH_FLAG_KEEP_DEBUG
        sim.set_params(q)
        sim.step(1)
        nc0 = sim.download_debug(force=False)[2]
        sim.set_params(p)
    else:
        nc0 = None
    sim.sync()
    sim.enable_timing(True)
    sim.reset_timings()
    ms = sim.step_profile(steps)
    tm = sim.timings()
    sim.enable_timing(False)
    q = p.copy()
    q.flags |= SPH_FLAG_KEEP_DEBUG
    sim.set_params(q)
    sim.step(1)
    nc = sim.download_debug(force=False)[2]
    dims = sim.grid_dims()
per = {k[:-3]: v / steps for k, v in tm.items() if k.endswith("_ms")}
print(json.dumps({"scene": scene, "side": side, "particles": n, "grid": list(dims), "window": [first, first + steps],
                  "arithmetic": "reference" if reference else "fast",
                  "ms_per_step": float(np.mean(ms)), "particle_updates_per_s": n / (float(np.mean(ms)) * 1e-3),
                  "mean_neighbours_before": float(nc0.mean()) if nc0 is not None else None,
                  "mean_neighbours": float(nc.mean()), "max_neighbours": int(nc.max()),
                  "kernel_ms_per_step": {k: round(v, 4) for k, v in per.items()},
                  "kernel_ms_per_million": {k: round(v / (n / 1e6), 4) for k, v in per.items()},
                  "force_frac_of_hbm_peak": 52.0 * n / (per["force"] * 1e-3) / 8e12}))
