"""Worker of tests/test_slab_cpu.py::test_gloo_two_ranks_match_single_domain (one process per rank).

Runs SlabRunner + DistExchange over the gloo backend with the CPU grid twin as compute backend,
including rebalancing; rank 0 gathers by particle id and compares with the single-domain twin."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import Oracle  # noqa: E402
from sph_amd import slab  # noqa: E402
from sph_amd.params import default_params  # noqa: E402
from slab_backends import OracleSlabBackend  # noqa: E402
from test_slab_cpu import by_id, moving_block  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    p = default_params()
    pos, vel, ids = moving_block()
    gn = orc.consts(p).ncell
    cols = orc.cell_ids(p, pos) // (gn * gn)
    cuts = slab.partition_columns(np.bincount(cols, minlength=gn), world)
    # start from a deliberately unbalanced cut so that rebalancing has work to do
    cuts[1] = min(cuts[1] + 3, gn - slab.MIN_SLAB_COLUMNS)
    b = OracleSlabBackend(orc, p, cuts[rank], cuts[rank + 1])
    m = (cols >= cuts[rank]) & (cols < cuts[rank + 1])
    b.upload(pos[m], vel[m], ids[m])
    # argv[2] = "tight": messages sized with no headroom at all, so that most steps need the overflow round
    tight = len(sys.argv) > 2 and sys.argv[2] == "tight"
    ex = slab.DistExchange(rank, world, headroom=0.5, min_capacity=1) if tight else slab.DistExchange(rank, world)
    runner = slab.SlabRunner(b, ex, rank, world, cuts, rebalance_every=2)
    nsteps = 8
    for _ in range(nsteps):
        runner.step()
    mine = b.download()
    gathered = [None] * world
    dist.gather_object(mine, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        got = by_id(*[np.concatenate(x) for x in zip(*gathered)])
        rp, rv = pos.copy(), vel.copy()
        for _ in range(nsteps):
            orc.step_grid(p, rp, rv, ids, order=0)
        ok = (np.array_equal(got[2], ids) and np.array_equal(got[0].view(np.uint32), rp.view(np.uint32))
              and np.array_equal(got[1].view(np.uint32), rv.view(np.uint32)))
        print("cuts", cuts, "->", runner.cuts, "sent bytes", runner.exchange.bytes_sent, "rounds", runner.exchange.rounds)
        # first step: counts + payload; afterwards one round per step unless a message outgrew its size
        assert runner.exchange.rounds > nsteps + 1 if tight else runner.exchange.rounds == nsteps + 1
        print("BITWISE_MATCH" if ok else "MISMATCH")
        assert runner.cuts != cuts, "rebalancing never moved the cut"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
