#!/usr/bin/env python3
"""Benchmark of the SPH hot path (one step = one updateParticles of the reference,
reference compute/compute.cpp:120-320) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

N = 1 : the 1M-particle dam-break block (100^3 jittered lattice in the corner of a tank twice
        its width), fp32, inputs resident in HBM before the timed region.
N > 1 : launched by torch.distributed.run, one rank per GPU; the dam-break block grows along x
        with N (weak scaling: per-GPU particle count fixed) and is slab-decomposed along x.

Prints ONE JSON line on rank 0 with the metric (particle-updates/s, whole job), a roofline
object for the dominant kernel (HIP-event durations measured inside the timed region on the
library's own stream) and a cpu_baseline object (the reference algorithm timed on this box's
host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sph_amd import capi, scenes  # noqa: E402
from sph_amd.params import SPH_FLAG_SURFACE_TENSION  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)
VALU_CLOCK_HZ = 2.4e9  # MI355X peak engine clock

# Algorithmic HBM bytes per particle and kernel: every array touched once per pass, neighbour
# re-reads assumed served on-chip.
BYTES_PER_PARTICLE = {
    "hash": 16,      # R pos 12, W cell 4
    "reorder": 56,   # R key 4 (histogram) ; R key 4 + R pos,vel 24 + W pos,vel 24 (scatter)
    "search": 12,    # R pos 12 (+ candidate bitmask W, 4 B per 32 candidates)
    "density": 16,   # R pos 12, W rho 4
    "force": 52,     # R pos 12, vel 12, rho 4 ; W pos 12, vel 12
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--side", type=int, default=100, help="dam-break block is side^3 particles per GPU")
    ap.add_argument("--workload", default="dam_break", choices=["dam_break", "uniform_cube"])
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rebalance-every", type=int, default=2, help="multi-GPU: steps between cut-plane moves")
    ap.add_argument("--cpu-sample-side", type=int, default=28,
                    help="CPU baseline runs the all-pairs reference algorithm on a side^3 sub-block")
    return ap.parse_args()


def cpu_baseline(args):
    """The reference algorithm (all-pairs) on host cores, on a bounded sample of the workload."""
    from oracle import Oracle, Ref  # checker only: timed here as the reported CPU baseline
    side = args.cpu_sample_side
    pos, vel, p = scenes.dam_break(side)
    n = pos.shape[0]
    orc = Oracle()
    threads = orc.num_threads()
    if Ref.available():
        ref = Ref(p)
        kind = "reference"

        def step():
            ref.step(pos, vel)
    else:
        kind = "port"

        def step():
            orc.step_allpairs(p, pos, vel)
    step()  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    reps = 0
    while True:
        step()
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 20:
            break
    per_step = dt / reps
    return {
        "value": n / per_step,
        "unit": "particle-updates/s",
        "cores": threads,
        "kind": kind,
        "sample": (f"{reps} all-pairs steps of a {side}^3 = {n}-particle sub-block of the dam-break field "
                   f"({per_step:.3f} s/step; the algorithm is O(N^2): extrapolated "
                   f"{per_step * (1e6 / n) ** 2 / 3600.0:.2f} h/step at 1M particles)"),
    }


def cpu_grid_twin(args):
    """This repository's uniform-grid CPU twin (NOT the reference algorithm: same arithmetic and results, but
    O(N) neighbour search) on the full workload, for a CPU figure at the size the GPU number is quoted on."""
    from oracle import Oracle  # checker only: timed here as a reported CPU figure
    pos, vel, p = scenes.dam_break(args.side)
    orc = Oracle()
    orc.step_grid(p, pos, vel, order=0)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < 5.0 and reps < 20):
        orc.step_grid(p, pos, vel, order=0)
        reps += 1
    per_step = (time.perf_counter() - t0) / reps
    return {"value": pos.shape[0] / per_step, "unit": "particle-updates/s", "cores": orc.num_threads(),
            "kind": "port, uniform-grid search (not the reference's all-pairs algorithm)",
            "sample": f"{reps} steps of the full {pos.shape[0]}-particle dam break ({per_step:.3f} s/step)"}


def from_profile(workload_name, key):
    """A per-launch figure of the dominant kernel from the committed rocprofv3 --pmc summary of this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload") == workload_name:
            return t.get(key)
    except (OSError, ValueError):
        pass
    return None


def run_single(args):
    import torch
    if args.workload == "dam_break":
        pos, vel, p = scenes.dam_break(args.side)
        workload = f"dam_break_{args.side}^3"
    else:
        pos, vel, p = scenes.uniform_cube(args.side)
        workload = f"uniform_cube_{args.side}^3"
    p.flags = SPH_FLAG_SURFACE_TENSION
    n = pos.shape[0]
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    sim = capi.Simulation(p, n_max=n, device=dev)
    sim.upload(pos, vel)          # inputs resident in HBM before the timed region
    sim.step(args.warmup)
    sim.sync()
    sim.enable_timing(not args.no_events)
    sim.reset_timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(args.steps)
    sim.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = sim.timings()
    gp, _ = sim.download()
    assert np.isfinite(gp).all(), "non-finite positions after the timed region"
    sim.close()
    return dict(n=n, dt=dt, timings=tm, workload=workload, params=p)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 or world > 1:
        from sph_amd import slab_bench
        return slab_bench.main(args)

    r = run_single(args)
    n, dt, tm = r["n"], r["dt"], r["timings"]
    steps = args.steps
    value = n * steps / dt
    out = {
        "metric": "particle-updates/sec",
        "value": value,
        "unit": "particle-updates/s",
        "n_gpus": 1,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": r["workload"], "particles": n, "tank_size": float(r["params"].tank_size),
                   "h": float(r["params"].h), "surface_tension": True, "parallelism": "1 GPU"},
    }
    if not args.no_events and tm["steps"] > 0:
        k = tm["steps"]
        per = {name: tm[name + "_ms"] / k for name in ("hash", "scan", "reorder", "search", "density", "force")}
        force_s = per["force"] * 1e-3
        achieved = BYTES_PER_PARTICLE["force"] * n / force_s / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_force", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": from_profile(r["workload"], "force_bytes_per_launch"),
            "bytes_per_particle": BYTES_PER_PARTICLE["force"], "avg_launch_ms": per["force"],
        }
        df = (BYTES_PER_PARTICLE["density"] + BYTES_PER_PARTICLE["force"]) * n / (
            (per["search"] + per["density"] + per["force"]) * 1e-3) / 1e9
        out["roofline_density_force"] = {"bound": "hbm", "achieved": df, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": df / HBM_PEAK_GBS, "bytes_per_particle": 68}
        # The sweeps are bound by vector-instruction issue, not by HBM: a wave64 instruction holds one of the
        # chip's 1024 SIMDs for 4 cycles.  Instruction count per launch from the committed counter profile,
        # duration live.
        insts = from_profile(r["workload"], "force_valu_wave_insts_per_launch")
        if insts:
            peak = 1024 * VALU_CLOCK_HZ / 4.0
            out["roofline_valu"] = {"bound": "valu-issue", "kernel": "k_force", "achieved": insts / force_s, "peak": peak,
                                    "unit": "wave-instructions/s", "frac": insts / force_s / peak,
                                    "wave_insts_per_launch": insts}
        step_bytes = sum(BYTES_PER_PARTICLE.values()) - BYTES_PER_PARTICLE["search"]  # the 140 B pipeline model
        whole = step_bytes * n / (dt / steps) / 1e9
        out["roofline_step"] = {"bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": whole / HBM_PEAK_GBS, "bytes_per_particle": step_bytes}
        out["kernel_ms_per_step"] = per
    if not args.no_cpu_baseline and rank == 0:
        out["cpu_baseline"] = cpu_baseline(args)
        if args.workload == "dam_break":
            out["cpu_grid_twin"] = cpu_grid_twin(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
