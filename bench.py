#!/usr/bin/env python3
"""Benchmark of the SPH hot path (one step = one updateParticles of the reference,
reference compute/compute.cpp:120-320) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

N = 1 : the 1M-particle dam-break block (100^3 jittered lattice in the corner of a tank twice
        its width), fp32, inputs resident in HBM before the timed region; the 16.8M-particle uniform
        cube and the 8M-particle dam break (the per-GPU share of the 64M one) follow as extra_configs.
N > 1 : one rank per GPU (launched by torch.distributed.run, or by this script itself when it is
        started plainly: the parent makes no GPU call and starts the ranks as children); the block
        holds 200^3 particles per GPU by default - 8 GPUs run the 64M-particle dam break - and is
        slab-decomposed along x (sph_amd/slab_bench.py).

The dam break is not a steady state: a step costs 0.25 ms while the block is still at rest density
(13 neighbours per particle) and more than twice that at peak compression (53 neighbours).  So
that one number means one thing whatever --steps/--warmup are, the timed window always starts at
the same PHASE: an untimed pre-roll of --preroll steps (default 60: peak compression of the
collapse, the most expensive phase), then W warm-up steps, then K timed steps.  The rest-density
figure (steps 5-25 from the initial block) is printed as a labelled extra.

Prints ONE JSON line on rank 0 with the metric (particle-updates/s, whole job), a roofline
object for the dominant kernel (HIP-event durations measured inside the timed region on the
library's own stream) and a cpu_baseline object (the reference algorithm timed on this box's
host cores on bounded samples).
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
from sph_amd.slab_bench import DEFAULT_SIDE_PER_GPU, peak_preroll  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)

# Algorithmic HBM bytes per particle and kernel: every array touched once per pass, neighbour
# re-reads assumed served on-chip.
BYTES_PER_PARTICLE = {
    "hash": 16,      # R pos 12, W cell 4
    "reorder": 56,   # R key 4 (histogram) ; R key 4 + R pos,vel 24 + W pos,vel 24 (scatter)
    "search": 16,    # search + density in one kernel: R pos 12, W rho 4 (+ candidate bitmask W, 4 B per 32 candidates)
    "force": 52,     # R pos 12, vel 12, rho 4 ; W pos 12, vel 12
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=None,
                    help="untimed steps before the warm-up: the phase of the collapse the timed window starts in "
                         "(default: peak compression, 60 steps at side 100, scaled with the square root of the side)")
    ap.add_argument("--side", type=int, default=None,
                    help="the block is side^3 particles per GPU (default 100 on one GPU, 200 per GPU on several)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="single GPU: skip the uniform-cube and 8M dam-break lines and the reference-arithmetic figure")
    ap.add_argument("--workload", default="dam_break", choices=["dam_break", "uniform_cube", "sloshing"])
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rest-figure", action="store_true", help="skip the rest-density extra (steps 5-25)")
    ap.add_argument("--rebalance-every", type=int, default=2, help="multi-GPU: steps between cut-plane moves")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the multi-GPU code path even with one rank (rehearsal of the slab driver on a 1-GPU box)")
    ap.add_argument("--transport", default="library", choices=["library", "torch"],
                    help="multi-GPU: halo exchange issued by the library (RCCL from C++) or by torch.distributed")
    return ap.parse_args()


CPU_SAMPLE_SIDES = (14, 27, 37)  # 2 744, 19 683 and 50 653 particles: the sizes the all-pairs fit is made on


def cpu_baseline(n_bench):
    """The reference algorithm (all-pairs, O(N^2)) on host cores, on bounded samples of the workload: side^3
    sub-blocks of the dam-break field at three sizes, a t = c*N^2 fit through them, and its value at the
    benchmark's particle count."""
    from oracle import Oracle, Ref  # checker only: timed here as the reported CPU baseline
    orc = Oracle()
    # A launcher may have pinned the ranks to one OpenMP thread each (torch.distributed.run does when the variable is
    # unset): the baseline is timed on this box's share of host cores, 16 per GPU at most.
    if orc.num_threads() < 2:
        orc.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    threads = orc.num_threads()
    kind = "reference" if Ref.available() else "port"
    points = []
    for side in CPU_SAMPLE_SIDES:
        pos, vel, p = scenes.dam_break(side)
        n = pos.shape[0]
        if kind == "reference":
            ref = Ref(p)

            def step():
                ref.step(pos, vel)
        else:
            def step():
                orc.step_allpairs(p, pos, vel)
        step()  # warm-up (thread pool, page faults)
        budget = 1.0 if side == CPU_SAMPLE_SIDES[0] else 6.0
        t0 = time.perf_counter()
        reps = 0
        while True:
            step()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= 50:
                break
        points.append((n, dt / reps, reps))
    ns = np.array([q[0] for q in points], np.float64)
    ts = np.array([q[1] for q in points], np.float64)
    slope = float(np.polyfit(np.log(ns[1:]), np.log(ts[1:]), 1)[0])  # measured exponent between the two large sizes
    c = float((ts * ns ** 2).sum() / (ns ** 4).sum())               # least-squares t = c N^2
    t_bench = c * float(n_bench) ** 2
    return {
        "value": n_bench / t_bench,
        "unit": "particle-updates/s",
        "cores": threads,
        "kind": kind,
        "sample": ("all-pairs steps of side^3 sub-blocks of the dam-break field: " +
                   ", ".join(f"N={n}: {t:.4f} s/step ({r} steps)" for n, t, r in points) +
                   f"; fit t = {c:.3e}*N^2 s (measured exponent between the two largest: {slope:.2f}); value = the fit "
                   f"at N={n_bench}: {t_bench / 3600.0:.2f} h/step"),
        "points": [{"particles": int(n), "s_per_step": t, "particle_updates_per_s": n / t} for n, t, _ in points],
        "fit_c_seconds_per_particle_squared": c,
        "fit_exponent_measured": slope,
    }


def cpu_grid_twin(args):
    """This repository's uniform-grid CPU twin (NOT the reference algorithm: same terms, but O(N) neighbour
    search) on the full workload, for a CPU figure at the size the GPU number is quoted on."""
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
            "sample": f"{reps} steps of the full {pos.shape[0]}-particle dam break from rest ({per_step:.3f} s/step)"}


def profile_figures(workload_name, args):
    """Counter-derived per-launch figures of the committed rocprofv3 --pmc profile, if and only if that profile was
    taken over this very window (same workload, pre-roll, warm-up and step count)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    w = t.get("window", {})
    if (t.get("workload") == workload_name and w.get("preroll") == args.preroll and w.get("warmup") == args.warmup
            and w.get("steps") == args.steps):
        return t
    return None


def run_single(args):
    import torch
    if args.workload == "uniform_cube":
        pos, vel, p = scenes.uniform_cube(args.side)
        workload = f"uniform_cube_{args.side}^3"
    else:
        pos, vel, p = scenes.dam_break(args.side)
        workload = f"{args.workload}_{args.side}^3"
    p.flags = SPH_FLAG_SURFACE_TENSION
    n = pos.shape[0]
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)

    def window(timing_level, with_rest_figure):
        """Fresh context, pre-roll, warm-up, then the timed steps with the given event level (0 none, 1 every kernel,
        2 the force pass only).  The state is a pure function of the step number, so a repeat sees the same window."""
        sim = capi.Simulation(p, n_max=n, device=dev)
        sim.upload(pos, vel)          # inputs resident in HBM before any timed region
        step0 = [0]

        def advance(k):
            """k steps; the sloshing workload turns gravity every step (the reference's a_gravity, compute.cpp:12)."""
            if args.workload == "sloshing":
                for _ in range(k):
                    sim.set_gravity(scenes.sloshing_gravity(step0[0]))
                    sim.step(1)
                    step0[0] += 1
            else:
                sim.step(k)
                step0[0] += k

        rest = None
        done = 0
        if with_rest_figure and args.preroll >= 25 and args.workload != "sloshing":
            advance(5)
            sim.sync()
            ms = sim.step_profile(20)
            done = 25
            step0[0] += 20
            rest = {"window_steps": [5, 25], "ms_per_step": float(ms.mean()), "value": n / (float(ms.mean()) * 1e-3),
                    "note": "block still at rest density (~13 neighbours per particle): the cheapest phase of the workload"}
        advance(max(0, args.preroll - done))
        advance(args.warmup)
        sim.sync()
        sim.enable_timing(timing_level)
        sim.reset_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.workload == "sloshing":
            advance(args.steps)
            sim.sync()
            step_ms = None
        else:
            step_ms = sim.step_profile(args.steps)  # returns when the steps have completed
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = sim.timings()
        gp, _ = sim.download()
        assert np.isfinite(gp).all(), "non-finite positions after the timed region"
        sim.close()
        return dt, tm, step_ms, rest

    # The timed region brackets only the force pass with events (what the roofline figure needs): an event record is a
    # barrier packet on the queue, and bracketing all six kernel categories costs the step 8 %.  The per-kernel
    # breakdown comes from a repeat of the same window with every category bracketed.
    dt, tm, step_ms, rest = window(0 if args.no_events else 2, not args.no_rest_figure)
    breakdown = None
    if not args.no_events:
        _, breakdown, _, _ = window(1, False)
    return dict(n=n, dt=dt, timings=tm, breakdown=breakdown, workload=workload, params=p, step_ms=step_ms, rest=rest)


def generated_window(scene, side, preroll, warmup, steps, reference_arith=False):
    """One more configuration on the same GPU: the scene is generated in HBM, advanced to its window, and the timed
    steps bracket the force pass with events (as the main window does).  Returns a bench-line-shaped dict."""
    from sph_amd.params import SPH_FLAG_REFERENCE_ARITH, default_params
    p = scenes.dam_break_params(side) if scene == "dam_break" else default_params(tank_size=side * scenes.REST_SPACING)
    p.flags = SPH_FLAG_SURFACE_TENSION | (SPH_FLAG_REFERENCE_ARITH if reference_arith else 0)
    n = side ** 3
    with capi.Simulation(p, n_max=n, device=int(os.environ.get("LOCAL_RANK", "0"))) as sim:
        sim.fill_lattice(side, side, side)
        sim.step(preroll + warmup)
        sim.sync()
        sim.enable_timing(2)
        sim.reset_timings()
        t0 = time.perf_counter()
        ms = sim.step_profile(steps)
        dt = time.perf_counter() - t0
        tm = sim.timings()
    force_ms = tm["force_ms"] / steps
    achieved = BYTES_PER_PARTICLE["force"] * n / (force_ms * 1e-3) / 1e9
    return {"workload": f"{scene}_{side}^3", "particles": n, "tank_size": float(p.tank_size),
            "arithmetic": "reference (double coefficient products, IEEE division and square root)" if reference_arith
                          else "fast",
            "phase": {"preroll_steps": preroll, "timed_steps": [preroll + warmup, preroll + warmup + steps]},
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": n * steps / dt, "unit": "particle-updates/s",
            "ms_per_step_device": {"min": float(ms.min()), "mean": float(ms.mean()), "max": float(ms.max())},
            "roofline": {"bound": "hbm", "kernel": "k_force", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_particle": BYTES_PER_PARTICLE["force"], "avg_launch_ms": force_ms}}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: this process makes no GPU call; the ranks are fresh children under torch.distributed.run
        from sph_amd import slab_bench
        sys.exit(slab_bench.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.gpus > 1 or world > 1 or args.force_slab:
        from sph_amd import slab_bench
        return slab_bench.main(args)
    if args.side is None:
        args.side = 100
    if args.preroll is None:
        args.preroll = peak_preroll(args.side) if args.workload != "uniform_cube" else 20

    r = run_single(args)
    n, dt, tm = r["n"], r["dt"], r["timings"]
    steps = args.steps
    value = n * steps / dt
    first = args.preroll + args.warmup
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
                   "h": float(r["params"].h), "surface_tension": True, "parallelism": "1 GPU",
                   "arithmetic": "fast (fp32 coefficients, refined rcp/rsq); neighbour sets exact"},
        "phase": {"preroll_steps": args.preroll, "timed_steps": [first, first + steps],
                  "description": ("the window starts at the peak compression of the dam-break collapse (~50 neighbours per "
                                  "particle; its first 20 steps are the most expensive 20 of the run)"
                                  if args.workload == "dam_break" and args.preroll == peak_preroll(args.side) else
                                  f"{args.preroll} steps after the initial state")},
    }
    if r["step_ms"] is not None and len(r["step_ms"]):
        sm = r["step_ms"]
        out["ms_per_step_device"] = {"min": float(sm.min()), "mean": float(sm.mean()), "max": float(sm.max())}
    if r["rest"]:
        out["rest_density_extra"] = r["rest"]
    if not args.no_events and tm["steps"] > 0:
        force_ms = tm["force_ms"] / tm["steps"]  # live, over the timed region
        prof = profile_figures(r["workload"], args)
        achieved = BYTES_PER_PARTICLE["force"] * n / (force_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_force", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": prof.get("force_bytes_per_launch") if prof else None,
            "bytes_per_particle": BYTES_PER_PARTICLE["force"], "avg_launch_ms": force_ms,
        }
        step_bytes = 140  # hash 16 + sort 56 + density 16 + force 52: the pipeline model
        whole = step_bytes * n / (dt / steps) / 1e9
        out["roofline_step"] = {"bound": "hbm", "achieved": whole, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": whole / HBM_PEAK_GBS, "bytes_per_particle": step_bytes}
        bd = r["breakdown"]
        if bd and bd["steps"] > 0:
            k = bd["steps"]
            per = {name: bd[name + "_ms"] / k for name in ("hash", "scan", "reorder", "search", "density", "force")}
            sweeps_ms = per["search"] + per["density"] + per["force"]
            df = 68.0 * n / (sweeps_ms * 1e-3) / 1e9
            out["roofline_density_force"] = {"bound": "hbm", "achieved": df, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": df / HBM_PEAK_GBS, "bytes_per_particle": 68,
                                             "kernels": "k_neighbours (search + density) + k_force"}
            out["kernel_ms_per_step"] = per
            out["kernel_ms_per_step_note"] = ("from a repeat of the same window with every kernel category bracketed by "
                                              "events (the timed region brackets the force pass only)")
        if prof and prof.get("issue_side"):
            # what actually bounds the gathering sweeps, from counters of this same window (see profiles/)
            out["issue_side"] = prof["issue_side"]
    if not args.no_extra_configs and not args.no_events and args.workload == "dam_break":
        out["extra_configs"] = [
            # steady-state neighbour search + force load: the 16.8M-particle cube that fills its tank
            generated_window("uniform_cube", 256, 20, 5, 20),
            # the per-GPU share of the 64M-particle dam break: the denominator of the multi-GPU line's efficiency
            generated_window("dam_break", DEFAULT_SIDE_PER_GPU, peak_preroll(DEFAULT_SIDE_PER_GPU), 5, 20),
        ]
        ref = generated_window("dam_break", args.side, args.preroll, args.warmup, min(steps, 20), reference_arith=True)
        out["reference_arithmetic_extra"] = {k: ref[k] for k in ("workload", "arithmetic", "phase", "steps", "ms_per_step",
                                                                 "value", "unit", "roofline")}
    if not args.no_cpu_baseline and rank == 0:
        out["cpu_baseline"] = cpu_baseline(n)
        if args.workload == "dam_break":
            out["cpu_grid_twin"] = cpu_grid_twin(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
