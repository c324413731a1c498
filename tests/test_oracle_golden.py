"""CPU-only: the oracle (oracle/sph_oracle.c) against the golden vectors generated from the
reference's own compute.cpp (tests/golden/make_golden.py), plus hand-derivable known answers."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, bits_equal, golden_case_names, load_golden, max_rel

STEPS = (1, 2, 5, 10)


@pytest.mark.parametrize("name", golden_case_names())
def test_allpairs_bit_exact_vs_reference(oracle, params, name):
    g = load_golden(name)
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for k in range(1, 11):
        rho, force = oracle.step_allpairs(params, pos, vel, want_debug=(k == 1))
        if k == 1:
            assert bits_equal(rho, g["rho1"]), "densities differ from the reference's"
            assert bits_equal(force, g["force1"]), "f_total differs from the reference's"
        if k in STEPS:
            assert bits_equal(pos, g[f"pos_{k}"]), f"positions differ after {k} steps"
            assert bits_equal(vel, g[f"vel_{k}"]), f"velocities differ after {k} steps"


@pytest.mark.parametrize("name", golden_case_names())
def test_grid_twin_id_order_bit_exact(oracle, params, name):
    """Restricting the sums to the 27-cell stencil is exact: non-neighbour terms are +-0."""
    g = load_golden(name)
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for k in range(1, 11):
        rho, force, nc = oracle.step_grid(params, pos, vel, order=1, want_debug=True)
        if k == 1:
            assert bits_equal(rho, g["rho1"]) and bits_equal(force, g["force1"])
            assert np.array_equal(nc, g["ncount0"])
        if k in STEPS:
            assert bits_equal(pos, g[f"pos_{k}"]) and bits_equal(vel, g[f"vel_{k}"])


@pytest.mark.parametrize("name", golden_case_names())
def test_grid_twin_stencil_order_within_tolerance(oracle, params, name):
    """Stencil summation order differs from the reference only by fp32 reassociation."""
    g = load_golden(name)
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    rho, force, nc = oracle.step_grid(params, pos, vel, order=0, want_debug=True)
    assert np.array_equal(nc, g["ncount0"])
    assert max_rel(rho, g["rho1"]) < 1e-5
    assert max_rel(force, g["force1"]) < 1e-5
    assert max_rel(pos, g["pos_1"]) < 1e-5
    assert max_rel(vel, g["vel_1"]) < 1e-5


@pytest.mark.parametrize("name", golden_case_names())
def test_neighbor_counts_and_cells(oracle, params, name):
    g = load_golden(name)
    assert np.array_equal(oracle.neighbor_counts(params, g["pos0"]), g["ncount0"])
    assert np.array_equal(oracle.cell_ids(params, g["pos0"]), g["cell0"])


def test_kernel_functions_bit_exact(oracle, params):
    with np.load(os.path.join(GOLDEN_DIR, "kernels.npz")) as z:
        r = z["r"]
        for name in ("lapl_viscosity", "grad_spiky", "w_poly6", "grad_poly6", "lapl_poly6"):
            got = np.array([oracle.kernel(name, params, float(x)) for x in r], np.float32)
            assert bits_equal(got, z[name]), name
        k = oracle.consts(params)
        assert bits_equal(np.array([k.h2, k.h6, k.h9], np.float32), z["consts"])
        for a, b in zip(z["vecnorm_in"], z["vecnorm_out"]):
            assert bits_equal(oracle.vec_normalize(a), b)


# ---- known answers that can be derived by hand -------------------------------------------------

def test_isolated_particle_density_and_free_fall(oracle, params):
    pos = np.array([[0.1, 0.1, 0.1]], np.float32)
    vel = np.zeros((1, 3), np.float32)
    rho, force = oracle.step_allpairs(params, pos, vel, want_debug=True)
    assert abs(float(rho[0]) - 321.843) < 1e-2           # m * Wpoly6(0)
    # force density g*rho divided by rho: v_y = -10*dt, y += v_y*dt   (gravity only; the
    # self colour-field gradient is exactly zero so the tension term vanishes)
    assert vel[0, 1] == pytest.approx(-10.0 * 0.005, rel=1e-6)
    assert pos[0, 1] == pytest.approx(0.1 - 10.0 * 0.005 * 0.005, rel=1e-6)
    assert vel[0, 0] == 0.0 and vel[0, 2] == 0.0


def test_kernels_vanish_outside_support(oracle, params):
    for name in ("lapl_viscosity", "grad_spiky", "w_poly6", "grad_poly6", "lapl_poly6"):
        assert oracle.kernel(name, params, 0.0100001) == 0.0
        assert oracle.kernel(name, params, 1.0) == 0.0
    assert oracle.kernel("grad_poly6", params, 0.0) == 0.0
    k = oracle.consts(params)
    assert oracle.kernel("lapl_poly6", params, 0.0) == pytest.approx(-k.c_poly6_lapl * k.h2 * k.h2, rel=1e-6)
    assert oracle.kernel("grad_spiky", params, 0.0) == pytest.approx(k.c_spiky_grad * 0.01 ** 2, rel=1e-6)


def test_wall_reflection(oracle, params):
    t = np.float32(params.tank_size)
    pos = np.array([[0.0001, 0.1, float(t) - 0.0001]], np.float32)
    vel = np.array([[-1.0, 0.0, 1.0]], np.float32)
    oracle.step_allpairs(params, pos, vel)
    assert pos[0, 0] == 0.0 and vel[0, 0] == 1.0          # clamped to the wall, velocity mirrored
    assert pos[0, 2] == t and vel[0, 2] == -1.0


def test_coincident_particles_feel_no_pressure(oracle, params):
    pos = np.array([[0.05, 0.05, 0.05], [0.05, 0.05, 0.05]], np.float32)
    vel = np.zeros((2, 3), np.float32)
    rho, force = oracle.step_allpairs(params, pos, vel, want_debug=True)
    assert rho[0] == rho[1]
    assert force[0, 0] == 0.0 and force[0, 2] == 0.0      # only gravity (y) acts
    assert bits_equal(force[0], force[1])


def test_r2_threshold_is_not_h_squared(oracle, params):
    """sqrtf(x) <= h holds for one more fp32 value than x <= h*h at h = 0.01."""
    k = oracle.consts(params)
    h2 = np.float32(params.h) * np.float32(params.h)
    assert np.float32(k.r2_max) == np.nextafter(h2, np.float32(1))
    assert np.sqrt(np.float32(k.r2_max)) <= np.float32(params.h)
    assert np.sqrt(np.nextafter(np.float32(k.r2_max), np.float32(1))) > np.float32(params.h)
