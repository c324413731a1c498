"""cProfile of rank 0's SlabRunner.step calls in a multi-rank run (development aid): wraps bench.py."""
import os, sys, runpy, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import slab
prof = cProfile.Profile()
orig = slab.SlabRunner.step
calls = [0]
def step(self):
    calls[0] += 1
    if self.rank == 0 and calls[0] > 15:
        prof.enable(); orig(self); prof.disable()
        if calls[0] == 45:
            s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue(), flush=True)
    else:
        orig(self)
slab.SlabRunner.step = step
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
