"""Development helper (GPU): error of the fast-arithmetic default path against the golden vectors and the CPU grid
twin, and kernel times of both arithmetic flavours on the 1M dam break."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden_case_names, load_golden, max_rel
from oracle import Oracle
from sph_amd import capi, scenes
from sph_amd.params import *

DBG = SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG
orc = Oracle()
for name in golden_case_names():
    g = load_golden(name)
    p = default_params(flags=DBG)
    with capi.Simulation(p, n_max=g["pos0"].shape[0]) as sim:
        sim.upload(g["pos0"], g["vel0"])
        sim.step(1)
        rho, f, nc, cell = sim.download_debug()
        pos, vel = sim.download()
        print(name, "nc", np.array_equal(nc, g["ncount0"]), "rho %.2e F %.2e pos %.2e vel %.2e" % (
            max_rel(rho, g["rho1"]), max_rel(f, g["force1"]), max_rel(pos, g["pos_1"]), max_rel(vel, g["vel_1"])), flush=True)
        for k in range(2, 11):
            sim.step(1)
            if k in (2, 5, 10):
                pos, vel = sim.download()
                print("   k=%d pos %.2e (noise %.2e) vel %.2e (noise %.2e)" % (k, max_rel(pos, g[f"pos_{k}"]), float(g[f"noise_pos_{k}"]),
                      max_rel(vel, g[f"vel_{k}"]), float(g[f"noise_vel_{k}"])), flush=True)

pos, vel, p = scenes.dam_break(100)
for label, presteps in (("rest", 0), ("compressed", 70)):
    q = p.copy(); q.flags = DBG
    with capi.Simulation(q, n_max=pos.shape[0]) as sim:
        sim.upload(pos, vel)
        sim.step(presteps)
        mp, mv = sim.download()
        po, vo = mp.copy(), mv.copy()
        sim.upload(mp, mv)
        rho_o, f_o, nc_o = orc.step_grid(q, po, vo, order=0, want_debug=True)
        sim.step(1)
        rho, f, nc, _ = sim.download_debug()
        gp, gv = sim.download()
    print("1M", label, "nc", np.array_equal(nc, nc_o), "mean nc %.1f" % nc.mean(), "rho %.2e F %.2e pos %.2e vel %.2e" % (
        max_rel(rho, rho_o), max_rel(f, f_o), max_rel(gp, po), max_rel(gv, vo)), flush=True)

for label, flags in (("fast", SPH_FLAG_SURFACE_TENSION), ("reference-arith", SPH_FLAG_SURFACE_TENSION | SPH_FLAG_REFERENCE_ARITH)):
    q = p.copy(); q.flags = flags
    with capi.Simulation(q, n_max=pos.shape[0]) as sim:
        sim.upload(pos, vel)
        sim.enable_timing(True)
        for phase, nsteps in (("0-20", 20), ("20-60", 40), ("60-100", 40), ("100-220", 120)):
            sim.reset_timings()
            t0 = time.perf_counter(); sim.step(nsteps); sim.sync(); dt = time.perf_counter() - t0
            tm = sim.timings()
            print(label, phase, "ms/step %.3f" % (dt / nsteps * 1e3), " ".join("%s %.3f" % (k[:-3], tm[k] / nsteps) for k in
                  ("hash_ms", "scan_ms", "reorder_ms", "search_ms", "density_ms", "force_ms")), flush=True)
