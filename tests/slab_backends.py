"""Test-only slab backend: the CPU grid twin (oracle) behind the interface SlabRunner drives.

Lets the multi-rank orchestration in sph_amd/slab.py (cut planes, record routing, ownership,
rebalancing, torch.distributed exchange) run on CPU with gloo.  Never used by the product.
"""
import numpy as np
import torch

from sph_amd.capi import RECORD_WORDS


class OracleSlabBackend:
    def __init__(self, oracle, params, col_lo, col_hi):
        self.oracle, self.params = oracle, params
        self.gn = oracle.consts(params).ncell
        self.lo, self.hi = col_lo, col_hi
        self.pos = np.zeros((0, 3), np.float32)
        self.vel = np.zeros((0, 3), np.float32)
        self.ids = np.zeros(0, np.int32)
        self.last_counts = None
        self.debug = None
        self.stray = 0

    def _cols(self, pos):
        if pos.shape[0] == 0:
            return np.zeros(0, np.int64)
        return self.oracle.cell_ids(self.params, np.ascontiguousarray(pos)).astype(np.int64) // (self.gn * self.gn)

    def upload(self, pos, vel, ids):
        self.pos, self.vel, self.ids = pos.copy(), vel.copy(), np.asarray(ids, np.int32).copy()

    def columns(self):
        return self.lo, self.hi, self.gn

    def set_columns(self, lo, hi):
        self.lo, self.hi = lo, hi

    @property
    def n(self):
        return self.ids.size

    def _records(self, mask):
        k = int(mask.sum())
        rec = np.empty((k, RECORD_WORDS), np.float32)
        rec[:, 0:3] = self.pos[mask]
        rec[:, 3:6] = self.vel[mask]
        rec[:, 6] = self.ids[mask].view(np.float32)
        return torch.from_numpy(rec)

    def pack(self):
        c = self._cols(self.pos)
        left = (c < self.lo + 2) if self.lo > 0 else np.zeros(c.size, bool)
        right = (c >= self.hi - 2) if self.hi < self.gn else np.zeros(c.size, bool)
        return self._records(left), self._records(right)

    def append(self, rec):
        if rec is None or rec.shape[0] == 0:
            return
        r = rec.cpu().numpy()
        c = self._cols(np.ascontiguousarray(r[:, 0:3]))
        self.stray += int(((c < self.lo - 2) | (c >= self.hi + 2)).sum())
        self.pos = np.concatenate([self.pos, r[:, 0:3]])
        self.vel = np.concatenate([self.vel, r[:, 3:6]])
        self.ids = np.concatenate([self.ids, np.ascontiguousarray(r[:, 6]).view(np.int32)])

    def step(self):
        c = self._cols(self.pos)
        own = (c >= self.lo) & (c < self.hi)
        self.last_counts = np.bincount(c[own] - self.lo, minlength=self.hi - self.lo).astype(np.int32)
        pos, vel = np.ascontiguousarray(self.pos), np.ascontiguousarray(self.vel)
        rho, f, nc = self.oracle.step_grid(self.params, pos, vel, self.ids, order=0, want_debug=True)
        self.debug = (rho[own], f[own], nc[own], self.ids[own])
        self.pos, self.vel, self.ids = pos[own], vel[own], self.ids[own]

    def column_counts(self):
        return self.last_counts

    def lost(self):
        return self.stray

    def max_column_shift(self):
        if self.vel.shape[0] == 0:
            return 0.0
        cell = np.float32(self.params.h) * np.float32(1.000244140625)
        return float(np.abs(self.vel[:, 0]).max() * np.float32(self.params.timestep) / cell)

    def download(self):
        return self.pos, self.vel, self.ids

    def empty_records(self, n):
        return torch.empty((n, RECORD_WORDS), dtype=torch.float32)
