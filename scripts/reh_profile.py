"""Per-phase wall time of SlabRunner.step in a multi-rank run (development aid): wraps bench.py."""
import os, sys, time, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import slab
acc = {"pack": 0.0, "exchange": 0.0, "append": 0.0, "step": 0.0, "rebalance": 0.0, "n": 0}
def step(self):
    b = self.backend
    t0 = time.perf_counter(); sl, sr = b.pack()
    t1 = time.perf_counter(); fl, fr = self.exchange.exchange(sl, sr)
    t2 = time.perf_counter(); b.append(fl); b.append(fr)
    t3 = time.perf_counter(); b.step(); b.sync()
    t4 = time.perf_counter()
    self.steps_done += 1
    if self.rebalance_every and self.steps_done % self.rebalance_every == 0 and self.nranks > 1:
        self.rebalance()
    t5 = time.perf_counter()
    for k, v in zip(("pack", "exchange", "append", "step", "rebalance"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += v
    acc["n"] += 1
    if acc["n"] % 20 == 0 and self.rank == 0:
        print("rank0 per step ms:", {k: round(acc[k] / acc["n"] * 1e3, 3) for k in acc if k != "n"},
              "rounds", getattr(self.exchange, "rounds", None), "send", sl.shape[0], sr.shape[0], flush=True)
slab.SlabRunner.step = step
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
