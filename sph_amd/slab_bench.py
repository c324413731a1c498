"""Multi-GPU leg of bench.py: one rank per GPU, slab decomposition along x.

Weak scaling: the block holds `side`^3 particles PER GPU (N * side^3 in total, as a cube of side
round(N^(1/3) * side)).  The default per-GPU side of 200 makes `--gpus 8` the 64M-particle dam break (400^3 in a
tank twice the block's width).  Every rank generates only its own lattice planes, in HBM.

Workloads (all on the same slab machinery):
  dam_break     block in the corner of a tank twice its width, gravity fixed;
  sloshing      the same block and tank, the gravity vector turned every step on every rank - how the reference's
                mutable a_gravity (compute/compute.cpp:12) is meant to be driven;
  uniform_cube  the lattice fills the tank (steady state).

`python bench.py --gpus N` started as a plain script launches its own ranks (launch_command below): the parent makes
no GPU call, starts `python -m torch.distributed.run` as a child and relays rank 0's JSON line.
"""
import contextlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

DEFAULT_SIDE_PER_GPU = 200   # 8M particles per GPU: 8 GPUs hold the 64M-particle dam break
HANG_DEADLINE_S = 1500       # a rank still running after this long prints every thread's stack and exits
PEAK_PREROLL_AT_SIDE_100 = 60  # steps until the 100^3 dam break is at peak compression


def peak_preroll(side_total):
    """Steps from rest to the peak compression of the collapse.  The block falls freely over its own height, so the
    time scales with the square root of the side: 60 steps at side 100, 120 at side 400."""
    return int(round(PEAK_PREROLL_AT_SIDE_100 * math.sqrt(side_total / 100.0)))


def total_side(side_per_gpu, world):
    return int(round(side_per_gpu * world ** (1.0 / 3.0)))


def resolve_sizes(args, world):
    """(per-GPU side, total side, pre-roll steps) of a multi-GPU run from the command line's values or defaults."""
    side = args.side if args.side else DEFAULT_SIDE_PER_GPU
    st = total_side(side, world)
    if args.preroll is not None:
        pre = args.preroll
    elif args.workload == "dam_break":
        pre = peak_preroll(st)
    else:
        pre = 20
    return side, st, pre


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(script, argv, gpus, visible_gpus, port):
    """What a plain `python bench.py --gpus N ...` starts: (command, extra environment).  With fewer visible GPUs than
    ranks the ranks share GPU 0 (rehearsal of the slab driver on a one-GPU box; records staged through gloo)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = {}
    if visible_gpus < gpus:
        env["SPH_BENCH_SHARED_GPU"] = "1"
    return cmd, env


def self_launch(script, argv, gpus):
    """Parent side of `python bench.py --gpus N` without a launcher: no GPU call is made here."""
    import torch  # device_count() does not initialise the GPU
    cmd, extra = launch_command(script, argv, gpus, torch.cuda.device_count(), free_port())
    env = dict(os.environ, **extra)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in res.stdout.splitlines():
        if line.startswith("{"):
            print(line)
        elif line.strip():
            print(line, file=sys.stderr)
    sys.stdout.flush()
    return res.returncode


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


def scene_params(workload, side_total):
    from . import scenes
    from .params import SPH_FLAG_SURFACE_TENSION, default_params
    if workload == "uniform_cube":
        p = default_params(tank_size=side_total * scenes.REST_SPACING)
    else:  # dam_break, sloshing: block in the corner of a tank twice its width
        p = scenes.dam_break_params(side_total)
    p.flags = SPH_FLAG_SURFACE_TENSION
    return p


def all_agree(dist, torch, ok):
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def arm_deadline():
    """A transport that never completes (a peer that died, a link that does not come up) would otherwise hang the run
    without a word: past the deadline (SPH_BENCH_DEADLINE_S, 0 = none) the rank prints every thread's stack to stderr
    and exits with status 1, and the launcher ends the other ranks."""
    import faulthandler
    deadline = float(os.environ.get("SPH_BENCH_DEADLINE_S", HANG_DEADLINE_S))
    if deadline > 0:
        faulthandler.dump_traceback_later(deadline, exit=True)


def disarm_deadline():
    import faulthandler
    faulthandler.cancel_dump_traceback_later()


def main(args):
    arm_deadline()
    # RCCL shares device memory between the ranks' processes through dmabuf handles; the legacy IPC mode is not
    # supported by the host driver of these machines.  Must be in the environment before the HIP runtime starts.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from . import capi, scenes, slab

    if "WORLD_SIZE" not in os.environ:  # --force-slab without a launcher: a world of one
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "RANK": "0", "WORLD_SIZE": "1"})
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and not (args.force_slab and world == 1):
        raise SystemExit(f"--gpus {args.gpus} under a launcher needs WORLD_SIZE={args.gpus}; got WORLD_SIZE={world}")
    shared_gpu = os.environ.get("SPH_BENCH_SHARED_GPU") == "1"  # rehearsal: all ranks on GPU 0, gloo transport
    library = getattr(args, "transport", "library") == "library" and not shared_gpu
    dev_index = 0 if shared_gpu else local
    if shared_gpu:
        torch.set_num_threads(1)  # host-staged records: many ranks' intra-op thread pools would fight for the cores
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # Control plane over gloo (host): communicator id, rebalancing all-reduce, barriers.  The records travel over RCCL
    # issued from C++ (library transport) or over a second, RCCL-backed torch group (torch transport).
    with _stdout_to_stderr():  # gloo reports its connections on stdout
        dist.init_process_group("gloo", rank=rank, world_size=world)

    side, side_t, preroll = resolve_sizes(args, world)
    p = scene_params(args.workload, side_t)
    n_total = side_t ** 3
    gn = capi.global_cells(p)
    # initial cut planes from the analytic column histogram of the lattice
    plane_x = (np.arange(side_t) + 0.5) * scenes.REST_SPACING
    cell = np.float32(p.h) * np.float32(1.000244140625)
    plane_col = np.minimum((plane_x / float(cell)).astype(np.int64), gn - 1)
    hist = np.bincount(plane_col, minlength=gn) * side_t * side_t
    cuts = slab.partition_columns(hist, world)
    # lattice planes whose (unjittered) column falls into this rank's slab; jitter may move a particle
    # across the face, which the first exchange handles like any other migrant
    mine = np.nonzero((plane_col >= cuts[rank]) & (plane_col < cuts[rank + 1]))[0]
    i0, i1 = (int(mine[0]), int(mine[-1]) + 1) if mine.size else (0, 0)
    n_max = int(2.0 * n_total / world) + 4 * side_t * side_t * 8
    rec_capacity = max(1 << 16, 64 * side_t * side_t)  # as the one-GPU runs of the same configurations
    sim = capi.SlabSimulation(p, n_max=n_max, col_lo=cuts[rank], col_hi=cuts[rank + 1], cols_capacity=gn,
                              device=dev_index, rec_capacity=1024 if library else rec_capacity)
    sim.fill_lattice(side_t, side_t, side_t, (i0, i1))  # every rank generates its own planes in HBM

    transport_note = None
    if library:
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = True
        try:
            with _stdout_to_stderr():  # RCCL announces its version on stdout when the communicator is created
                sim.attach_rccl(uid[0], rank, world, rec_capacity)
        except capi.SphError as e:
            print(f"rank {rank}: attaching the RCCL transport failed: {e}", file=sys.stderr)
            ok = False
        if not all_agree(dist, torch, ok):
            # every rank falls back together: records packed by the library, moved by torch.distributed
            library = False
            transport_note = "library transport could not be attached on every rank; fell back to torch.distributed"
            sim.close()
            sim = capi.SlabSimulation(p, n_max=n_max, col_lo=cuts[rank], col_hi=cuts[rank + 1], cols_capacity=gn,
                                      device=dev_index, rec_capacity=rec_capacity)
            sim.fill_lattice(side_t, side_t, side_t, (i0, i1))
    if library:
        runner = slab.LibraryRunner(sim, rank, world, cuts, rebalance_every=args.rebalance_every)
        runner.prime()
    else:
        data_group = None
        if not shared_gpu:
            with _stdout_to_stderr():
                data_group = dist.new_group(backend="nccl")  # RCCL: record tensors stay in HBM
        stream = torch.cuda.Stream(device=device)
        sim.use_stream(stream.cuda_stream)   # kernels and RCCL transfers are ordered on one stream
        runner = slab.SlabRunner(sim, slab.DistExchange(rank, world, device=device, group=data_group), rank, world,
                                 cuts, rebalance_every=args.rebalance_every)
        torch.cuda.set_stream(stream)

    step_no = [0]

    def advance(k):
        for _ in range(k):
            if args.workload == "sloshing":  # the same vector on every rank, from the step number
                sim.set_gravity(scenes.sloshing_gravity(step_no[0]))
            runner.step()
            step_no[0] += 1

    def timed(k):
        sim.sync()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        advance(k)
        sim.sync()
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter() - t0

    # rest-density window (steps 5-25 from the initial block) as a labelled extra, then on to the phase of the main one
    rest_dt = None
    if args.workload == "dam_break" and preroll >= 25 and not args.no_rest_figure:
        advance(5)
        rest_dt = timed(20)
    advance(max(0, preroll - step_no[0]))
    advance(args.warmup)
    sim.sync()
    sim.enable_timing(0 if args.no_events else 2)  # the force pass only: event records are not free (see bench.py)
    sim.reset_timings()
    first_timed = step_no[0]
    dt = timed(args.steps)
    if library:
        runner.drain()
    tm = sim.timings()
    n_own = sim.last_own() if library else sim.n
    sent = 0.0 if library else float(runner.exchange.bytes_sent)
    xstats = sim.exchange_stats() if library else (0, 0, [0, 0], [0, 0])
    stats = torch.tensor([dt, float(n_own), float(sim.lost() + (sim.missed() if library else 0)), tm["force_ms"],
                          tm["search_ms"] + tm["density_ms"], sent, float(rest_dt or 0.0), float(xstats[1]),
                          float(max(xstats[2] + xstats[3]))], dtype=torch.float64)
    gathered = [torch.zeros_like(stats) for _ in range(world)] if rank == 0 else None
    dist.gather(stats, gathered, dst=0)
    if rank == 0:
        g = torch.stack(gathered).numpy()
        dt_max = float(g[:, 0].max())
        n_now = int(g[:, 1].sum())
        assert n_now == n_total, f"particles lost: {n_now} of {n_total}"
        assert g[:, 2].sum() == 0, "particles left a local grid"
        if library:
            how = "RCCL send/recv issued by the library, overlapped with the interior columns"
        elif shared_gpu:
            how = "REHEARSAL: all ranks share one GPU, records staged through gloo on the host"
        else:
            how = "torch.distributed send/recv (RCCL)"
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
            "config": {"workload": f"{args.workload}_{side_t}^3", "particles": n_total,
                       "particles_per_gpu": n_total // world, "side_per_gpu": side,
                       "single_gpu_denominator": f"bench.py --gpus 1 reports dam_break_{side}^3 under extra_configs: "
                                                 "the same particle count per GPU on one GPU",
                       "tank_size": float(p.tank_size), "h": float(p.h), "surface_tension": True,
                       "arithmetic": "fast (fp32 coefficients, refined rcp/rsq); neighbour sets exact",
                       "parallelism": f"{world} slabs along x, 2-column halo, " + how,
                       "final_cuts": [int(c) for c in runner.cuts],
                       "particles_per_rank": [int(v) for v in g[:, 1]]},
            "phase": {"preroll_steps": preroll, "timed_steps": [first_timed, first_timed + args.steps],
                      "description": ("peak compression of the dam-break collapse (pre-roll scaled with the square root of "
                                      "the block's side)" if args.workload == "dam_break" and args.preroll is None else
                                      f"{first_timed} steps after the initial state")},
        }
        if transport_note:
            out["config"]["transport_note"] = transport_note
        if g[:, 6].max() > 0:
            r_dt = float(g[:, 6].max())
            out["rest_density_extra"] = {"window_steps": [5, 25], "ms_per_step": r_dt / 20 * 1e3,
                                         "value": n_total * 20 / r_dt,
                                         "note": "block still at rest density: the cheapest phase of the workload"}
        if library:
            out["exchange"] = {"remainder_rounds_per_rank": [int(v) for v in g[:, 7]],
                               "largest_next_message_records": int(g[:, 8].max()), "record_capacity": rec_capacity}
        if not args.no_events:
            # the slowest rank's force pass bounds the step
            k = int(np.argmax(g[:, 3]))
            force_ms = g[k, 3] / args.steps
            n_k = g[k, 1]
            achieved = 52.0 * n_k / (force_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_force", "achieved": achieved, "peak": 8000.0,
                               "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                               "bytes_per_particle": 52, "avg_launch_ms": force_ms, "rank": k,
                               "note": "force pass of the slowest rank per step (faces + interior launches summed), "
                                       "HIP events on the library's stream inside the timed region"}
        if not library:
            out["halo_bytes_per_step_per_rank"] = float(g[:, 5].mean() / args.steps)
        if not args.no_cpu_baseline:
            import bench  # the same all-pairs fit as the single-GPU line, at this run's particle count
            out["cpu_baseline"] = bench.cpu_baseline(n_total)
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
    disarm_deadline()
