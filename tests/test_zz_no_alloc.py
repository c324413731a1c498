"""The entry point that allocates nothing (sph_update_particles_noalloc) against the one-call entry point it is a
variant of (sph_update_particles): same step, so the same bits, on two calls in a row, the second on a smaller
input; sizes beyond its fixed scratch are refused."""
import numpy as np
import pytest

from conftest import bits_equal

from sph_amd import capi, scenes
from sph_amd.params import SPH_FLAG_KEEP_DEBUG, SPH_FLAG_SURFACE_TENSION, default_params

pytestmark = pytest.mark.gpu


def _step_both(torch, pos, vel, p, steps):
    """`steps` updateParticles through both entry points; returns ((pos, vel) existing, (pos, vel) no-alloc)."""
    hp, hv = pos.copy(), vel.copy()
    dp, dv = torch.from_numpy(pos.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(steps):
        capi.update_particles(hp, hv, p)
        capi.update_particles_noalloc(dp.data_ptr(), dv.data_ptr(), pos.shape[0], p, stream)
    torch.cuda.synchronize()
    return (hp, hv), (dp.cpu().numpy(), dv.cpu().numpy())


def test_no_alloc_entry_point_matches_update_particles():
    torch = pytest.importorskip("torch")
    p = default_params(flags=SPH_FLAG_SURFACE_TENSION)
    big, big_v = scenes.random_block(20000, seed=11)
    rng = np.random.default_rng(3)
    big_v = ((rng.random(big_v.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.4)).astype(np.float32)
    (hp, hv), (gp, gv) = _step_both(torch, big, big_v, p, 3)
    assert bits_equal(gp, hp) and bits_equal(gv, hv)
    # second call in a row on a smaller input, in a smaller tank: the shared array holds the first call's leftovers
    q = default_params(flags=SPH_FLAG_SURFACE_TENSION, tank_size=0.12)
    small, small_v = scenes.random_block(700, tank_size=0.12, seed=12)
    (hp, hv), (gp, gv) = _step_both(torch, small, small_v, q, 2)
    assert bits_equal(gp, hp) and bits_equal(gv, hv)
    # and back up again
    (hp, hv), (gp, gv) = _step_both(torch, big, big_v, p, 1)
    assert bits_equal(gp, hp) and bits_equal(gv, hv)


def test_no_alloc_entry_point_against_the_oracle(oracle):
    """One step through the no-allocation entry point against the CPU restatement of the reference's algorithm
    (all-pairs, the reference's own summation order): the default arithmetic's one-step bound of 1e-5 on positions
    and velocities, scaled by the field maximum."""
    torch = pytest.importorskip("torch")
    from conftest import max_rel
    p = default_params(flags=SPH_FLAG_SURFACE_TENSION)
    pos, vel = scenes.random_block(3000, seed=5)  # the reference's default scene size (main.cpp:8)
    po, vo = pos.copy(), vel.copy()
    oracle.step_allpairs(p, po, vo)
    dp, dv = torch.from_numpy(pos.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    capi.update_particles_noalloc(dp.data_ptr(), dv.data_ptr(), 3000, p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert max_rel(dp.cpu().numpy(), po) <= 1e-5 and max_rel(dv.cpu().numpy(), vo) <= 1e-5


def test_no_alloc_entry_point_allocates_nothing_and_can_be_captured():
    """What the entry point is for: device memory in use does not change across calls (first call on the device
    included: its scratch is a static array of the library), and a call can be recorded into a HIP graph and replayed -
    something no allocating call survives - with the bits of the direct calls."""
    torch = pytest.importorskip("torch")
    p = default_params(flags=SPH_FLAG_SURFACE_TENSION)
    pos, vel = scenes.random_block(5000, seed=21)
    dp, dv = torch.from_numpy(pos.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    gp, gv = dp.clone(), dv.clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        capi.update_particles_noalloc(dp.data_ptr(), dv.data_ptr(), 5000, p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the amount of device memory in use"
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        capi.update_particles_noalloc(gp.data_ptr(), gv.data_ptr(), 5000, p, side.cuda_stream)
        graph.capture_end()
    torch.cuda.synchronize()
    # capturing executes nothing: gp, gv still hold the initial state; three replays = the three direct calls
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert bits_equal(gp.cpu().numpy(), dp.cpu().numpy()) and bits_equal(gv.cpu().numpy(), dv.cpu().numpy())


def test_no_alloc_entry_point_rejects_what_does_not_fit():
    torch = pytest.importorskip("torch")
    p = default_params(flags=SPH_FLAG_SURFACE_TENSION)
    n = capi.NOALLOC_MAX_PARTICLES + 1
    d = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(capi.SphError) as e:
        capi.update_particles_noalloc(d.data_ptr(), d.data_ptr(), n, p, None)
    assert e.value.code == 4 and "exceed" in str(e.value)
    wide = default_params(flags=SPH_FLAG_SURFACE_TENSION, tank_size=0.7)  # 70^3 cells > 64^3
    with pytest.raises(capi.SphError) as e:
        capi.update_particles_noalloc(d.data_ptr(), d.data_ptr(), 100, wide, None)
    assert e.value.code == 4
    dbg = default_params(flags=SPH_FLAG_SURFACE_TENSION | SPH_FLAG_KEEP_DEBUG)
    with pytest.raises(capi.SphError) as e:
        capi.update_particles_noalloc(d.data_ptr(), d.data_ptr(), 100, dbg, None)
    assert e.value.code == 1
    capi.update_particles_noalloc(d.data_ptr(), d.data_ptr(), 0, p, None)  # nothing to do is fine
