"""CPU tests of the multi-rank path: cut planes, record routing, ownership, rebalancing and the
torch.distributed exchange (gloo, world_size 2), with the CPU grid twin standing in for the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits_equal

from sph_amd import scenes, slab
from sph_amd.params import default_params
from slab_backends import OracleSlabBackend


def moving_block(n_side=12, seed=3, speed=0.8):
    """Jittered block in the middle of the default tank with velocities up to ~speed m/s in x:
    0.8 m/s * 0.005 s = 0.4 cell per step, so particles migrate across cut planes within a few steps."""
    pos, vel, ids = scenes.jittered_lattice(n_side, n_side, n_side, origin=(0.06, 0.05, 0.05), seed=seed,
                                            return_ids=True)
    rng = np.random.default_rng(seed)
    vel = ((rng.random(vel.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(2 * speed)).astype(np.float32)
    return pos, vel, ids


def by_id(pos, vel, ids):
    o = np.argsort(ids)
    return pos[o], vel[o], ids[o]


# ---- cut planes ----------------------------------------------------------------------------------------

def test_partition_balances_and_respects_min_width():
    hist = np.zeros(140, np.int64)
    hist[:70] = 14286  # dam break: half the tank is empty
    cuts = slab.partition_columns(hist, 8)
    assert cuts[0] == 0 and cuts[-1] == 140 and len(cuts) == 9
    widths = np.diff(cuts)
    assert widths.min() >= slab.MIN_SLAB_COLUMNS
    loads = [hist[cuts[r]:cuts[r + 1]].sum() for r in range(8)]
    assert max(loads) <= 1.2 * hist.sum() / 8
    with pytest.raises(ValueError):
        slab.partition_columns(np.ones(10), 4)


def test_partition_with_empty_histogram_still_valid():
    cuts = slab.partition_columns(np.zeros(40, np.int64), 4)
    assert np.diff(cuts).min() >= slab.MIN_SLAB_COLUMNS and cuts[-1] == 40


def test_rebalance_moves_one_column_at_a_time():
    hist = np.zeros(60, np.int64)
    hist[:20] = 100
    cuts = [0, 30, 60]
    seen = [cuts]
    for _ in range(30):
        cuts = slab.rebalance_cuts(cuts, hist)
        assert abs(cuts[1] - seen[-1][1]) <= 1
        seen.append(cuts)
    assert cuts[1] == 10  # converged to the balanced cut


def test_min_width_follows_the_column_shift_and_cuts_move_one_column():
    assert slab.min_columns_for_shift(0.0) == slab.MIN_SLAB_COLUMNS
    assert slab.min_columns_for_shift(0.4) == 4
    assert slab.min_columns_for_shift(2.2) == 6
    hist = np.full(40, 100, np.int64)
    cuts = [0, 4, 8, 40]  # far from balanced (13/13/14) and narrower than 7 columns
    new = slab.rebalance_cuts(cuts, hist, min_cols=7)
    assert all(abs(a - b) <= 1 for a, b in zip(new, cuts)), "a face moves one column per call at most"
    for _ in range(40):
        new = slab.rebalance_cuts(new, hist, min_cols=7)
    assert min(np.diff(new)) >= 7 and new == slab.partition_columns(hist, 3, 7)


def fast_jet(n_side=10, speed=5.0, seed=5):
    """Block moving in +x at ~speed m/s: 5 m/s * 0.005 s = 2.5 cell columns per step."""
    pos, vel, ids = scenes.jittered_lattice(n_side, n_side, n_side, origin=(0.02, 0.05, 0.05), seed=seed,
                                            return_ids=True)
    rng = np.random.default_rng(seed)
    vel = (rng.random(vel.shape, dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    vel[:, 0] += np.float32(speed)
    return pos, vel, ids


def test_particles_crossing_several_columns_per_step(oracle):
    """A jet that crosses 2.5 columns per step: slabs as wide as min_columns_for_shift asks for hand every
    particle to the right owner with the one-hop exchange (nothing strays), bit-identical to one domain."""
    p = default_params()
    pos, vel, ids = fast_jet()
    gn = oracle.consts(p).ncell
    cols = oracle.cell_ids(p, pos) // (gn * gn)
    shift = float(np.abs(vel[:, 0]).max() * 0.005 / 0.01)
    cuts = slab.partition_columns(np.ones(gn, np.int64), 3, slab.min_columns_for_shift(shift))
    backends = []
    for r in range(3):
        b = OracleSlabBackend(oracle, p, cuts[r], cuts[r + 1])
        m = (cols >= cuts[r]) & (cols < cuts[r + 1])
        b.upload(pos[m], vel[m], ids[m])
        backends.append(b)
    ex = slab.LocalExchange(3)
    rp, rv = pos.copy(), vel.copy()
    for s in range(5):
        slab.step_local_slabs(backends, ex)
        cuts = slab.rebalance_local_slabs(backends, cuts)
        oracle.step_grid(p, rp, rv, ids, order=0)
        assert sum(b.n for b in backends) == ids.size
    assert all(b.lost() == 0 for b in backends)
    got = by_id(*[np.concatenate(x) for x in zip(*[b.download() for b in backends])])
    assert bits_equal(got[0], rp) and bits_equal(got[1], rv)
    assert backends[1].n + backends[2].n > 0, "the jet must have crossed a cut plane"


def test_too_narrow_slabs_report_stray_particles(oracle):
    """The same jet over 4-column slabs: particles jump over a whole slab; the receiver counts them."""
    p = default_params()
    pos, vel, ids = fast_jet(speed=16.0)
    gn = oracle.consts(p).ncell
    cols = oracle.cell_ids(p, pos) // (gn * gn)
    cuts = [0, 4, 8, 12, gn]
    backends = []
    for r in range(4):
        b = OracleSlabBackend(oracle, p, cuts[r], cuts[r + 1])
        m = (cols >= cuts[r]) & (cols < cuts[r + 1])
        b.upload(pos[m], vel[m], ids[m])
        backends.append(b)
    ex = slab.LocalExchange(4)
    for s in range(3):
        slab.step_local_slabs(backends, ex)
    assert sum(b.lost() for b in backends) > 0


# ---- several slabs in one process -------------------------------------------------------------------------

@pytest.mark.parametrize("nslabs,rebalance", [(2, False), (3, False), (3, True)])
def test_local_slabs_match_single_domain_bitwise(oracle, nslabs, rebalance):
    p = default_params()
    pos, vel, ids = moving_block()
    gn = oracle.consts(p).ncell
    cols = oracle.cell_ids(p, pos) // (gn * gn)
    cuts = slab.partition_columns(np.bincount(cols, minlength=gn), nslabs)
    backends = []
    for r in range(nslabs):
        b = OracleSlabBackend(oracle, p, cuts[r], cuts[r + 1])
        m = (cols >= cuts[r]) & (cols < cuts[r + 1])
        b.upload(pos[m], vel[m], ids[m])
        backends.append(b)
    ex = slab.LocalExchange(nslabs)
    rp, rv = pos.copy(), vel.copy()
    migrated = 0
    owner_before = np.searchsorted(cuts, cols, side="right") - 1
    for s in range(8):
        slab.step_local_slabs(backends, ex)
        if rebalance:
            cuts = slab.rebalance_local_slabs(backends, cuts)
        oracle.step_grid(p, rp, rv, ids, order=0)
        assert sum(b.n for b in backends) == ids.size, "particles lost or duplicated"
    got = by_id(*[np.concatenate(x) for x in zip(*[b.download() for b in backends])])
    assert np.array_equal(got[2], ids)
    assert bits_equal(got[0], rp) and bits_equal(got[1], rv)
    owner_after = np.empty(ids.size, np.int64)
    for r, b in enumerate(backends):
        owner_after[b.download()[2]] = r
    migrated = int((owner_after != owner_before).sum())
    assert migrated > 0, "the scene must exercise migration across cut planes"


# ---- torch.distributed, gloo, world_size 2 -------------------------------------------------------------------

@pytest.mark.parametrize("sizing", ["headroom", "tight"])
def test_gloo_two_ranks_match_single_domain(tmp_path, sizing):
    script = os.path.join(ROOT, "tests", "slab_gloo_worker.py")
    port = 29500 + (os.getpid() % 2000) + (7 if sizing == "tight" else 0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, script, str(tmp_path), sizing], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [pr.communicate(timeout=240)[0] for pr in procs]
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{outs[r]}"
    assert "BITWISE_MATCH" in outs[0], outs[0]


def test_library_runner_rebalances_consistently_over_gloo():
    script = os.path.join(ROOT, "tests", "library_runner_gloo_worker.py")
    port = 31500 + (os.getpid() % 2000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [pr.communicate(timeout=240)[0] for pr in procs]
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{outs[r]}"
    assert "CONSISTENT" in outs[0], outs[0]


def test_rebalance_respects_column_capacities():
    """A slab whose context can hold only so many columns is never asked to grow beyond that: the surplus goes to its
    neighbours, and when no admissible move exists the planes stay."""
    hist = np.zeros(60, np.int64)
    hist[50:] = 100  # everything on the right: slab 0 would like to swallow the empty left half
    cuts = [0, 12, 24, 60]
    for _ in range(40):
        cuts = slab.rebalance_cuts(cuts, hist, max_shift=3, min_cols=4, max_cols=[20, 30, 40])
        assert cuts[1] - cuts[0] <= 20 and cuts[2] - cuts[1] <= 30 and cuts[3] - cuts[2] <= 40
    assert cuts[1] == 20 and cuts[2] == 50          # as far right as the capacities allow
    free = slab.rebalance_cuts([0, 12, 24, 60], hist, max_shift=3, min_cols=4)
    assert free[1] == 15 and free[2] == 27         # without capacities: three columns per call towards the load


def test_face_moves_follow_the_front_and_yield_to_narrow_slabs():
    """Planes travel as many columns per step as the fastest particle crosses (within the exchange's limit), every
    column of travel widens the minimum slab, and where the narrowest slab cannot afford that the planes slow down so
    that slabs under pressure can narrow by a column per call."""
    assert slab.face_move_for_shift(0.3) == 1 and slab.face_move_for_shift(2.2) == 3
    assert slab.face_move_for_shift(50.0) == slab.MAX_FACE_MOVE
    assert slab.min_columns_for_shift(2.2, face_move=3) == 3 + 2 + 3
    assert slab.min_columns_for_shift(0.0) == slab.MIN_SLAB_COLUMNS
    wide = [0, 40, 80, 120]
    assert slab.face_move_for_shift(3.5, wide) == 4
    tight = [0, 40, 50, 120]                       # a 10-column slab at a shift of 3.5 columns per step
    assert slab.face_move_for_shift(3.5, tight) == 3   # 4 + 2 + 3 = 9 < 10: the slab may still narrow
    assert slab.face_move_for_shift(3.5, [0, 40, 47, 120]) == 1


def test_library_cut_planner_agrees_with_the_python_one():
    """sph_slab_plan_partition / sph_slab_plan_next (what a native driver calls) against partition_columns /
    face_move_for_shift / min_columns_for_shift / rebalance_cuts on random column loads, shifts and capacities."""
    from sph_amd import capi
    rng = np.random.default_rng(7)
    cases = 0
    for trial in range(400):
        ncols = int(rng.integers(8, 200))
        nslabs = int(rng.integers(1, max(2, ncols // 8) + 1))
        kind = trial % 4
        if kind == 0:
            hist = rng.integers(0, 1000, ncols)
        elif kind == 1:   # a block in one corner: most columns empty
            hist = np.zeros(ncols, np.int64)
            hist[: ncols // 3] = rng.integers(1, 500, ncols // 3)
        elif kind == 2:   # fluid piled up against the far wall
            hist = (np.arange(ncols) ** 3).astype(np.int64)
        else:
            hist = np.zeros(ncols, np.int64)
        min_cols = int(rng.integers(1, max(2, ncols // nslabs) + 1))
        if nslabs * min_cols > ncols:
            with pytest.raises(capi.SphError):
                capi.plan_partition(hist, nslabs, min_cols)
            continue
        cuts = slab.partition_columns(hist, nslabs, min_cols)
        assert capi.plan_partition(hist, nslabs, min_cols) == cuts
        # planes somewhere else than the balanced partition, a front of some speed, contexts of some capacity
        other = slab.partition_columns(rng.integers(0, 1000, ncols), nslabs, min_cols)
        shift = float(np.float32(rng.choice([0.0, 0.3, 1.0, 1.7, 2.5, 4.2, 9.5])))
        margin = float(rng.choice([0.0, 1.0]))
        widths = np.diff(other)
        caps = None if trial % 3 == 0 else [int(w + rng.integers(0, 12)) for w in widths]
        move = slab.face_move_for_shift(shift, other)
        want = slab.rebalance_cuts(other, hist, max_shift=move,
                                   min_cols=slab.min_columns_for_shift(shift + margin, move), max_cols=caps)
        got, got_move, got_shift = capi.plan_next(hist, other, shift, margin, caps)
        assert got_move == move and got == want, (trial, ncols, nslabs, other, shift, margin, caps)
        assert got_shift == pytest.approx(shift + margin)
        assert got[0] == 0 and got[-1] == ncols and all(abs(a - b) <= move for a, b in zip(got, other))
        cases += 1
    assert cases > 200


def test_library_cut_planner_refuses_bad_cuts():
    from sph_amd import capi
    hist = np.ones(32, np.int64)
    for bad in ([1, 16, 32], [0, 16, 31], [0, 20, 20, 32], [0, 20, 10, 32]):
        with pytest.raises(capi.SphError):
            capi.plan_next(hist, bad, 1.0)
    with pytest.raises(capi.SphError):
        capi.plan_next(-hist, [0, 16, 32], 1.0)
    with pytest.raises(capi.SphError):
        capi.plan_next(hist, [0, 16, 32], -1.0)


def test_column_of_is_the_sort_s_column():
    """sph_column_of against the oracle's cell index (reference compute.cpp has no grid: the oracle's cell ids are the
    ones the GPU sort is tested against) on random coordinates and on the cell faces themselves."""
    from oracle import Oracle
    from sph_amd import capi
    p = default_params()
    gn = capi.global_cells(p)
    cell = np.float32(p.h) * np.float32(1.000244140625)
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-0.05, p.tank_size + 0.05, 500).astype(np.float32),
                         (np.arange(0, gn + 1, dtype=np.float32) * cell),
                         np.nextafter(np.arange(1, gn + 1, dtype=np.float32) * cell, np.float32(0))])
    pos = np.zeros((xs.size, 3), np.float32)
    pos[:, 0] = xs
    want = Oracle().cell_ids(p, pos) // (gn * gn)
    got = np.array([capi.column_of(p, x) for x in xs])
    assert np.array_equal(got, want)
