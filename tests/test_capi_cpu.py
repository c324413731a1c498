"""CPU-only checks of the C-ABI library: it loads, exports every symbol that include/sph_amd.h
declares, validates arguments, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from sph_amd import capi
from sph_amd.params import SphParams, default_params


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sph_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sph_amd.h but not exported"
    assert sorted(capi.exported_symbols()) == names, "capi.py binding list and header disagree"


def test_default_params_match_reference_globals():
    lib = capi.load()
    p = SphParams()
    assert lib.sph_params_default(C.byref(p)) == 0
    q = default_params()
    # reference compute/compute.cpp:8-17
    assert (p.sound_speed, p.rho0, p.h) == (1.0, 600.0, np.float32(0.01))
    assert p.mass == np.float32(0.00020543) and p.dyn_visc == np.float32(0.2)
    assert tuple(p.gravity) == (0.0, -10.0, 0.0)
    assert p.timestep == np.float32(0.005) and p.tank_size == np.float32(0.2154)
    assert p.surf_tension == np.float32(0.0728)
    assert bytes(p) == bytes(q), "python default_params() and sph_params_default() disagree"


def test_version_and_error_strings():
    lib = capi.load()
    assert b"gfx950" in lib.sph_version()
    assert lib.sph_params_default(None) != 0
    assert b"NULL" in lib.sph_last_error()


def test_argument_validation_without_device():
    lib = capi.load()
    ctx = C.c_void_p()
    p = default_params()
    p.h = -1.0
    assert lib.sph_create(C.byref(p), 10, 0, C.byref(ctx)) == 1  # SPH_ERR_INVALID
    p = default_params()
    assert lib.sph_create(C.byref(p), 0, 0, C.byref(ctx)) == 1
    p.cell_size = 0.001  # smaller than h: the 27-cell stencil would miss neighbours
    assert lib.sph_create(C.byref(p), 10, 0, C.byref(ctx)) == 1
    assert lib.sph_step(None, 1) == 1
    assert lib.sph_upload(None, None, None, 0) == 1


def test_no_cpu_fallback():
    """Without a GPU every compute entry point fails loudly instead of computing on the host."""
    lib = capi.load()
    if lib.sph_device_count() > 0:
        pytest.skip("a GPU is visible")
    ctx = C.c_void_p()
    p = default_params()
    assert lib.sph_create(C.byref(p), 100, 0, C.byref(ctx)) == 2  # SPH_ERR_NO_DEVICE
    assert b"no CPU path" in lib.sph_last_error()
    pos = np.zeros((4, 3), np.float32)
    with pytest.raises(capi.SphError):
        capi.update_particles(pos, pos.copy())


def test_fill_random_block_is_reference_shaped():
    """Counterpart of initParticles' fill (reference compute.cpp:108-117): [0, tank/2]^3, v = 0."""
    pos, vel = capi.fill_random_block(5000, seed=3)
    assert pos.min() >= 0.0 and pos.max() <= np.float32(0.2154) / 2
    assert np.all(vel == 0.0)
    assert abs(pos.mean() - 0.2154 / 4) < 2e-3
    pos2, _ = capi.fill_random_block(5000, seed=3)
    assert np.array_equal(pos, pos2)
    pos3, _ = capi.fill_random_block(5000, seed=4)
    assert not np.array_equal(pos, pos3)
