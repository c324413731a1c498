"""CPU-only, build container only: the oracle against the LIVE reference build
(oracle/_ref/libsph_ref.so = the reference's unmodified compute.cpp) on fresh seeds and
non-default parameters.  Skipped where the reference build is absent (the GPU box)."""
import numpy as np
import pytest

from conftest import bits_equal

from oracle import Ref
from sph_amd import scenes
from sph_amd.params import default_params

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (no reference checkout here)")


@pytest.mark.parametrize("n,seed", [(7, 11), (200, 12), (1500, 13)])
def test_random_blocks(oracle, n, seed):
    p = default_params()
    ref = Ref(p)
    pos, vel = scenes.random_block(n, seed=seed)
    a, b = (pos.copy(), vel.copy()), (pos.copy(), vel.copy())
    for _ in range(6):
        rr, fr = ref.step(*a, want_debug=True)
        ro, fo = oracle.step_allpairs(p, *b, want_debug=True)
        assert bits_equal(rr, ro) and bits_equal(fr, fo)
        assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1])


def test_non_default_parameters(oracle):
    """Tilted gravity, other h / tank / viscosity: the restatement follows the globals."""
    p = default_params()
    p.h = 0.0125
    p.tank_size = 0.3
    p.dyn_visc = 0.35
    p.rho0 = 550.0
    p.sound_speed = 1.3
    p.timestep = 0.004
    p.gravity[0], p.gravity[1], p.gravity[2] = 2.5, -9.6, 0.4
    ref = Ref(p)
    pos, vel = scenes.jittered_lattice(9, 9, 9, seed=21)
    a, b = (pos.copy(), vel.copy()), (pos.copy(), vel.copy())
    for _ in range(4):
        rr, fr = ref.step(*a, want_debug=True)
        ro, fo = oracle.step_allpairs(p, *b, want_debug=True)
        assert bits_equal(rr, ro) and bits_equal(fr, fo)
        assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1])
    k = oracle.consts(p)
    assert bits_equal(np.array([k.h2, k.h6, k.h9], np.float32), ref.consts())
