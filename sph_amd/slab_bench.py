"""Multi-GPU leg of bench.py: one rank per GPU (torch.distributed.run), slab decomposition along x.

Weak scaling: the dam-break block holds `side`^3 particles PER GPU (N * side^3 in total, as a cube of
side round((N)^(1/3) * side)), in a tank twice the block's width.  `--side 200 --gpus 8` is the
64M-particle dam break.  Every rank generates only its own lattice planes.
"""
import contextlib
import json
import os
import sys
import time

import numpy as np


@contextlib.contextmanager
def _stdout_to_stderr():
    """Native libraries that chat on file descriptor 1 are sent to stderr: stdout carries the one JSON line."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(keep, 1)
        os.close(keep)


def main(args):
    import torch
    import torch.distributed as dist

    from . import capi, scenes, slab
    from .params import SPH_FLAG_SURFACE_TENSION

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} ranks (launch with torch.distributed.run); "
                         f"got WORLD_SIZE={world}")
    shared_gpu = os.environ.get("SPH_BENCH_SHARED_GPU") == "1"  # rehearsal: all ranks on GPU 0, gloo transport
    library = getattr(args, "transport", "library") == "library" and not shared_gpu
    dev_index = 0 if shared_gpu else local
    if shared_gpu:
        torch.set_num_threads(1)  # host-staged records: many ranks' intra-op thread pools would fight for the cores
    torch.cuda.set_device(dev_index)
    # library transport: the records travel over RCCL issued from C++; torch.distributed (gloo, host) only hands out the
    # communicator id and reduces the column histogram when cut planes move.  torch transport: everything through it.
    with _stdout_to_stderr():  # gloo reports its connections on stdout
        dist.init_process_group("gloo" if (shared_gpu or library) else "nccl", rank=rank, world_size=world)
    device = torch.device("cuda", dev_index)

    side = int(round(args.side * world ** (1.0 / 3.0)))
    p = scenes.dam_break_params(side)
    p.flags = SPH_FLAG_SURFACE_TENSION
    n_total = side ** 3
    gn = capi.global_cells(p)
    # initial cut planes from the analytic column histogram of the lattice
    plane_x = (np.arange(side) + 0.5) * scenes.REST_SPACING
    cell = np.float32(p.h) * np.float32(1.000244140625)
    plane_col = np.minimum((plane_x / float(cell)).astype(np.int64), gn - 1)
    hist = np.bincount(plane_col, minlength=gn) * side * side
    cuts = slab.partition_columns(hist, world)
    # lattice planes whose (unjittered) column falls into this rank's slab; jitter may move a particle
    # across the face, which the first exchange handles like any other migrant
    mine = np.nonzero((plane_col >= cuts[rank]) & (plane_col < cuts[rank + 1]))[0]
    i0, i1 = (int(mine[0]), int(mine[-1]) + 1) if mine.size else (0, 0)
    n_max = int(2.0 * n_total / world) + 4 * side * side * 8
    rec_capacity = max(1 << 16, 48 * side * side)
    sim = capi.SlabSimulation(p, n_max=n_max, col_lo=cuts[rank], col_hi=cuts[rank + 1], cols_capacity=gn,
                              device=dev_index, rec_capacity=1024 if library else rec_capacity)
    sim.fill_lattice(side, side, side, (i0, i1))  # every rank generates its own planes in HBM
    preroll = getattr(args, "preroll", 0)
    if library:
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        with _stdout_to_stderr():  # RCCL announces its version on stdout when the communicator is created
            sim.attach_rccl(uid[0], rank, world, rec_capacity)
        runner = slab.LibraryRunner(sim, rank, world, cuts, rebalance_every=args.rebalance_every)
        runner.prime()
    else:
        stream = torch.cuda.Stream(device=device)
        sim.use_stream(stream.cuda_stream)   # kernels and RCCL transfers are ordered on one stream
        runner = slab.SlabRunner(sim, slab.DistExchange(rank, world, device=device), rank, world, cuts,
                                 rebalance_every=args.rebalance_every)
        torch.cuda.set_stream(stream)
    for _ in range(preroll + args.warmup):
        runner.step()
    sim.sync()
    sim.enable_timing(0 if args.no_events else 2)  # the force pass only: event records are not free (see bench.py)
    sim.reset_timings()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    sim.sync()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if library:
        runner.drain()
    tm = sim.timings()
    n_own = sim.last_own() if library else sim.n
    sent = 0.0 if library else float(runner.exchange.bytes_sent)
    stats = torch.tensor([dt, float(n_own), float(sim.lost() + (sim.missed() if library else 0)), tm["force_ms"],
                          tm["search_ms"] + tm["density_ms"], sent], dtype=torch.float64)
    gathered = [torch.zeros_like(stats) for _ in range(world)] if rank == 0 else None
    if shared_gpu or library:
        dist.gather(stats, gathered, dst=0)
    else:
        g = [torch.zeros_like(stats, device=device) for _ in range(world)] if rank == 0 else None
        dist.gather(stats.to(device), g, dst=0)
        gathered = [x.cpu() for x in g] if rank == 0 else None
    if rank == 0:
        g = torch.stack(gathered).numpy()
        dt_max = float(g[:, 0].max())
        n_now = int(g[:, 1].sum())
        assert n_now == n_total, f"particles lost: {n_now} of {n_total}"
        assert g[:, 2].sum() == 0, "particles left a local grid"
        out = {
            "metric": "particle-updates/sec",
            "value": n_total * args.steps / dt_max,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"dam_break_{side}^3", "particles": n_total, "particles_per_gpu": n_total // world,
                       "tank_size": float(p.tank_size), "h": float(p.h), "surface_tension": True,
                       "parallelism": (f"{world} slabs along x, 2-column halo, " +
                                       ("RCCL send/recv issued by the library, overlapped with the interior columns"
                                        if library else "torch.distributed send/recv")),
                       "phase": {"preroll_steps": preroll, "timed_steps": [preroll + args.warmup, preroll + args.warmup + args.steps]},
                       "final_cuts": [int(c) for c in runner.cuts],
                       "particles_per_rank": [int(v) for v in g[:, 1]]},
        }
        if not args.no_events:
            # the slowest rank's force kernel bounds the step
            k = int(np.argmax(g[:, 3]))
            force_ms = g[k, 3] / args.steps
            n_k = g[k, 1]
            achieved = 52.0 * n_k / (force_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_force", "achieved": achieved, "peak": 8000.0,
                               "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                               "bytes_per_particle": 52, "avg_launch_ms": force_ms, "rank": k}
        if not library:
            out["halo_bytes_per_step_per_rank"] = float(g[:, 5].mean() / args.steps)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
