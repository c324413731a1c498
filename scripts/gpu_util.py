"""Lane utilisation the load binning can reach: per block of 256 sorted slots, neighbour counts sorted descending
and cut into waves; utilisation = sum(nnb) / sum(64 * max per wave) (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_SURFACE_TENSION, SPH_FLAG_KEEP_DEBUG
pos, vel, p = scenes.dam_break(100)
p.flags = SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG
sim = capi.Simulation(p, n_max=pos.shape[0]); sim.upload(pos, vel)
done = 0
for target in (1, 30, 70, 120, 200):
    sim.step(target - done); done = target
    _, _, nc, cell = sim.download_debug()
    order = np.lexsort((np.arange(nc.size), cell))       # slot order: cell, then id
    k = nc[order].astype(np.int64)
    n = k.size // 256 * 256
    for blk in (64, 256, 1024):
        b = k[: k.size // blk * blk].reshape(-1, blk)
        nat = b.reshape(-1, 64)
        srt = -np.sort(-b, axis=1); srt = srt.reshape(-1, 64)
        print("step %3d block %4d: natural order %.3f  sorted in block %.3f   (mean nnb %.1f)" % (
            target, blk, nat.sum() / (64.0 * nat.max(axis=1).sum()), srt.sum() / (64.0 * srt.max(axis=1).sum()), k.mean()), flush=True)
