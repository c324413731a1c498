"""Worker of tests/test_slab_cpu.py::test_library_runner_rebalances_consistently_over_gloo (one process per rank).

LibraryRunner keeps torch.distributed off the data path: it only all-reduces the column histogram and the largest
column shift when the cut planes move.  Here two ranks run it over gloo around a stand-in for the slab context (the
real one needs a GPU) that records what the runner asks of it; both ranks must arrive at the same cuts, move each
face by no more columns per call than the fastest particle of ANY rank crosses per step (2.5 here: three columns), pass the largest shift of ALL ranks to every context, and never move the faces
while an exchange is in flight (the all-reduce is collected one step after it was started)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sph_amd import slab  # noqa: E402


class StandIn:
    def __init__(self, hist, cuts, rank, shift):
        self.hist, self.cuts, self.rank, self.shift = hist, list(cuts), rank, shift
        self.calls, self.face_shift, self.next, self.in_flight = [], None, None, False

    def step_begin(self):
        self.calls.append("begin")
        self.in_flight = True

    def step_end(self):
        self.calls.append("end")
        self.in_flight = False
        if self.next is not None:  # new faces take effect with the step after the one whose records they shaped
            self.cuts[self.rank], self.cuts[self.rank + 1] = self.next
            self.next = None

    def load_work(self):
        lo, hi = self.cuts[self.rank], self.cuts[self.rank + 1]
        counts = self.hist[lo:hi].astype(np.int32)
        return lo, counts, counts * 22, self.shift  # every particle at the rest density's 13 neighbours: work = 22 each

    def set_face_shift(self, v):
        self.face_shift = v

    def set_next_columns(self, lo, hi):
        assert not self.in_flight, "faces moved while an exchange was in flight"
        self.next = (lo, hi)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncols = 40
    hist = np.zeros(ncols, np.int64)
    hist[2:14] = 1000  # all the particles sit in the left third: the cut must travel left
    cuts = [0, 30, ncols]
    b = StandIn(hist, cuts, rank, shift=0.5 + 2.0 * rank)
    runner = slab.LibraryRunner(b, rank, world, cuts, rebalance_every=1)
    seen = [list(cuts)]
    for _ in range(12):
        runner.step()
        b.cuts = list(runner.cuts) if b.next is None else b.cuts
        seen.append(list(runner.cuts))
    moves = [abs(a[1] - c[1]) for a, c in zip(seen, seen[1:])]
    assert max(moves) == 3, moves  # ceil of the larger rank's shift of 2.5 columns per step
    assert seen[-1][1] < 30 - 8, seen[-1]
    # the larger of the two ranks' shifts plus the runner's one column of margin, on both ranks
    assert abs(b.face_shift - 3.5) < 1e-2, b.face_shift
    assert b.calls[:2] == ["begin", "end"]
    gathered = [None] * world
    dist.all_gather_object(gathered, seen)
    assert gathered[0] == gathered[1], "ranks disagree about the cut planes"
    if rank == 0:
        print("CONSISTENT", seen[-1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
