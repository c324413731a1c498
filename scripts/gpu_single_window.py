"""Development/measurement aid (GPU): one context, dam break of side^3 particles generated in HBM; device time per step
over a window.   python scripts/gpu_single_window.py <side> <first_step> <steps>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION
side, first, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = scenes.dam_break_params(side)
p.flags = SPH_FLAG_SURFACE_TENSION
with capi.Simulation(p, n_max=side ** 3) as sim:
    sim.fill_lattice(side, side, side)
    sim.step(first)
    sim.sync()
    sim.enable_timing(True)
    sim.reset_timings()
    ms = sim.step_profile(steps)
    tm = sim.timings()
print(json.dumps({"particles": side ** 3, "window": [first, first + steps], "ms_per_step": float(np.mean(ms)),
                  "kernels_ms_per_step": {k: round(v / steps, 4) for k, v in tm.items() if k.endswith("_ms")}}))
