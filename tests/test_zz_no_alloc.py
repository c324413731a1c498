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
