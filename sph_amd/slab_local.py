"""Several slabs of one simulation in ONE process on ONE GPU, records moved by the library (LOCAL transport).

The configurations the reference's north star quotes on 8 GPUs - the 64M-particle dam break and the 256M-particle
sloshing tank - fit the 288 GB of a single MI355X.  This driver runs them there as K slab contexts, through exactly
the code path a K-GPU run takes (slab contexts, sph_slab_step_begin/_end, cut planes that follow the particles),
with device copies standing in for the xGMI transfers.  It checks what can be checked at that size: every particle
is advanced by exactly one slab, none is lost or missed, positions stay finite and inside the tank, the slabs stay
balanced.

    python -m sph_amd.slab_local --scene dam_break --side 400 --slabs 8 --steps 20
    python -m sph_amd.slab_local --scene sloshing  --side 640 --slabs 8 --steps 20
"""
import argparse
import json
import time

import numpy as np

from . import capi, scenes, slab
from .params import SPH_FLAG_SURFACE_TENSION


def build(scene, side, nslabs, rec_factor=64, n_max_factor=1.3):
    """K slab contexts holding a side^3 jittered block in the corner of a tank twice its width, generated in HBM."""
    p = scenes.dam_break_params(side)
    p.flags = SPH_FLAG_SURFACE_TENSION
    gn = capi.global_cells(p)
    plane_x = (np.arange(side) + 0.5) * scenes.REST_SPACING
    cell = np.float32(p.h) * np.float32(1.000244140625)
    plane_col = np.minimum((plane_x / float(cell)).astype(np.int64), gn - 1)
    hist = np.bincount(plane_col, minlength=gn) * side * side
    cuts = slab.partition_columns(hist, nslabs)
    n_total = side ** 3
    # own particles (+30 % for imbalance between rebalances; cut planes steered by work give a slab of uncompressed
    # fluid up to 2.7 times the particles of a compressed one: long runs ask for more) + four ghost columns of ~1.5
    # lattice planes each
    n_max = int(n_max_factor * n_total / nslabs) + 8 * side * side
    backends = []
    for r in range(nslabs):
        mine = np.nonzero((plane_col >= cuts[r]) & (plane_col < cuts[r + 1]))[0]
        i0, i1 = (int(mine[0]), int(mine[-1]) + 1) if mine.size else (0, 0)
        # room for the cut planes to wander (8 B per cell: a full-width grid for every slab would be tens of GB)
        b = capi.SlabSimulation(p, n_max=n_max, col_lo=cuts[r], col_hi=cuts[r + 1],
                                cols_capacity=min(gn, cuts[r + 1] - cuts[r] + max(64, gn // 3)),
                                rec_capacity=1024)  # the pack/append buffers of the caller-driven protocol are not used
        b.fill_lattice(side, side, side, (i0, i1))
        backends.append(b)
    rec_capacity = max(1 << 16, rec_factor * side * side)
    for r, b in enumerate(backends):
        b.attach_local(backends[r - 1] if r > 0 else None, backends[r + 1] if r < nslabs - 1 else None, rec_capacity)
    return p, backends, cuts


def run(scene, side, nslabs, steps, rebalance_every=2, check=True, n_max_factor=1.3):
    p, backends, cuts = build(scene, side, nslabs, n_max_factor=n_max_factor)
    n_total = side ** 3
    slab.halo_local_slabs(backends)
    for b in backends:
        b.sync()
    t0 = time.perf_counter()
    worst, worst_work, work_trace = 1.0, 1.0, []
    for s in range(steps):
        if scene == "sloshing":  # the reference's mutable a_gravity (compute.cpp:12), turned every step
            g = scenes.sloshing_gravity(s)
            for b in backends:
                q = b.params.copy()
                q.gravity[0], q.gravity[1], q.gravity[2] = float(g[0]), float(g[1]), float(g[2])
                b.set_params(q)
        slab.step_local_slabs_overlapped(backends)
        if rebalance_every and s % rebalance_every == rebalance_every - 1:
            st = {}
            cuts = slab.rebalance_next_local_slabs(backends, cuts, stats=st)
            own = [b.last_own() for b in backends]
            worst = max(worst, max(own) * nslabs / float(sum(own)))
            w = max(st["load"]) * nslabs / float(sum(st["load"]))  # what the cut planes steer: sum of (9 + neighbours)
            worst_work = max(worst_work, w)
            work_trace.append(round(w, 3))
    for b in backends:
        b.sync()
    dt = time.perf_counter() - t0
    own = [b.last_own() for b in backends]
    out = {"scene": scene, "particles": n_total, "slabs": nslabs, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "particle_updates_per_s": n_total * steps / dt, "own_particles": own,
           "lost": sum(b.lost() for b in backends), "missed": sum(b.missed() for b in backends),
           "worst_imbalance": worst, "worst_work_imbalance": worst_work,
           "work_imbalance_every_20th_rebalance": work_trace[::20], "cuts": [int(c) for c in cuts]}
    if check:
        lo, hi = np.inf, -np.inf
        finite = True
        for b in backends:
            bp, bv, bid = b.download()
            k = b.last_own()
            finite = finite and bool(np.isfinite(bp[:k]).all() and np.isfinite(bv[:k]).all())
            lo, hi = min(lo, float(bp[:k].min())), max(hi, float(bp[:k].max()))
        out.update({"finite": finite, "min_coordinate": lo, "max_coordinate": hi, "tank_size": float(p.tank_size)})
    for b in backends:
        b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dam_break", choices=["dam_break", "sloshing"])
    ap.add_argument("--side", type=int, default=400)
    ap.add_argument("--slabs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rebalance-every", type=int, default=2)
    ap.add_argument("--n-max-factor", type=float, default=1.3,
                    help="particle capacity of a slab in units of the mean slab (work-steered cut planes give a slab "
                         "of uncompressed fluid more particles than one of compressed fluid)")
    a = ap.parse_args()
    print(json.dumps(run(a.scene, a.side, a.slabs, a.steps, a.rebalance_every, n_max_factor=a.n_max_factor)))


if __name__ == "__main__":
    main()
