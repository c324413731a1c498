"""Diagnosis aid (GPU, WHATIF build of the library only): time ONE step's search and force kernels with parts of their
work switched off (results of that step are meaningless; the state is discarded afterwards).
   SPH_AMD_LIB=<whatif build> python scripts/gpu_whatif.py <side> <first_step>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION

side, first = int(sys.argv[1]), int(sys.argv[2])
MODES = [("base", 0), ("both: no neighbour loop at all (prologue and epilogue only)", 128), ("force: all neighbours from the LDS tile", 1), ("force: no pair arithmetic", 2),
         ("force: LDS only, no arithmetic", 3), ("force: no loads", 4), ("force: no loads, no arithmetic", 6),
         ("force: no tile (all gathered)", 8), ("force: no tile, no arithmetic", 10),
         ("search: one address per wave", 16), ("search: no distance arithmetic", 32),
         ("search: no loads", 64), ("search: no loads, no arithmetic", 96), ("base again", 0)]
for name, mode in MODES:
    p = scenes.dam_break_params(side)
    p.flags = SPH_FLAG_SURFACE_TENSION
    with capi.Simulation(p, n_max=side ** 3) as sim:
        sim.fill_lattice(side, side, side)
        sim.step(first)
        sim.sync()
        q = p.copy()
        q.flags |= mode << 24
        sim.set_params(q)
        sim.enable_timing(True)
        sim.reset_timings()
        sim.step_profile(1)
        tm = sim.timings()
    print(json.dumps({"mode": name, "search_us": round(tm["search_ms"] * 1e3, 1), "force_us": round(tm["force_ms"] * 1e3, 1)}), flush=True)
