"""Synthetic particle fields (host side, numpy) for parity tests and benchmarks.

The reference's only scene is ``initParticles`` (reference compute/compute.cpp:97-118): N
particles uniform in [0, tank/2]^3 with zero velocity, seeded from the wall clock.  Parity runs
need reproducible inputs, so every generator here takes a seed.

Perfect lattices are deliberately not offered: the reference normalises the colour-field
gradient with no magnitude threshold (compute.cpp:255), so on a symmetric lattice the
surface-tension direction of interior particles is pure rounding noise and no two summation
orders agree.  Lattice scenes are therefore jittered.
"""
import numpy as np

from .params import default_params

REST_SPACING = float((np.float64(0.00020543) / 600.0) ** (1.0 / 3.0))  # (m/rho0)^(1/3) = 0.0069958 m


def random_block(n, tank_size=0.2154, seed=12345):
    """Reference-style block: uniform in [0, tank/2]^3, v = 0 (compute.cpp:108-117)."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3), dtype=np.float32) * np.float32(tank_size / 2.0)).astype(np.float32)
    vel = np.zeros((n, 3), dtype=np.float32)
    return pos, vel


def jittered_lattice(nx, ny, nz, spacing=REST_SPACING, jitter=0.1, origin=(0.0, 0.0, 0.0),
                     seed=12345, chunk=1 << 22):
    """nx*ny*nz block on a simple-cubic lattice, each coordinate jittered by +-jitter*spacing.

    Particle (i,j,k) sits at origin + (i+0.5, j+0.5, k+0.5)*spacing + jitter, x fastest.
    Generated in chunks so that 16M+ particle fields do not need float64 temporaries of full size.
    """
    n = nx * ny * nz
    pos = np.empty((n, 3), dtype=np.float32)
    rng = np.random.default_rng(seed)
    o = np.asarray(origin, dtype=np.float64)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = np.arange(a, b, dtype=np.int64)
        ijk = np.stack([idx % nx, (idx // nx) % ny, idx // (nx * ny)], axis=1).astype(np.float64)
        jit = (rng.random((b - a, 3)) * 2.0 - 1.0) * (jitter * spacing)
        pos[a:b] = (o + (ijk + 0.5) * spacing + jit).astype(np.float32)
    vel = np.zeros((n, 3), dtype=np.float32)
    return pos, vel


def dam_break(side, seed=12345, jitter=0.1):
    """side^3 jittered block at rest spacing in the corner of a tank twice as wide.

    side=100 is the 1M-particle single-GPU configuration, side=400 the 64M one.
    Returns (pos, vel, params).
    """
    block = side * REST_SPACING
    params = default_params(tank_size=2.0 * block)
    pos, vel = jittered_lattice(side, side, side, seed=seed, jitter=jitter)
    return pos, vel, params


def uniform_cube(side, seed=12345, jitter=0.1):
    """side^3 jittered lattice filling the whole tank (steady-state neighbour search load)."""
    params = default_params(tank_size=side * REST_SPACING)
    pos, vel = jittered_lattice(side, side, side, seed=seed, jitter=jitter)
    return pos, vel, params


def dam_break_slab_x(side_x, side_yz, seed=12345, jitter=0.1):
    """Dam-break block of side_x * side_yz^2 particles (x is the slab-decomposition axis)."""
    params = default_params(tank_size=2.0 * max(side_x, side_yz) * REST_SPACING)
    pos, vel = jittered_lattice(side_x, side_yz, side_yz, seed=seed, jitter=jitter)
    return pos, vel, params


def wall_contact(n, tank_size=0.2154, seed=7):
    """Particles on, at and beyond the tank walls with velocities pointing outward.

    Exercises the reflect-and-clamp boundary (compute.cpp:286-319) and cell-index clamping.
    """
    rng = np.random.default_rng(seed)
    t = np.float32(tank_size)
    pos = (rng.random((n, 3), dtype=np.float32) * t).astype(np.float32)
    vel = ((rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(4.0)).astype(np.float32)
    k = n // 6
    pos[0 * k:1 * k, 0] = 0.0
    pos[1 * k:2 * k, 1] = t
    pos[2 * k:3 * k, 2] = -np.float32(0.003)      # outside, below
    pos[3 * k:4 * k, 0] = t + np.float32(0.004)   # outside, above
    pos[4 * k:5 * k, 1] = np.nextafter(t, np.float32(0))
    return pos, vel


def sloshing_gravity(step, period=400, amplitude_deg=15.0, g=10.0):
    """Gravity vector of the sloshing scene: rotated about z by a sinusoid of the step index."""
    ang = np.deg2rad(amplitude_deg) * np.sin(2.0 * np.pi * step / period)
    return np.array([g * np.sin(ang), -g * np.cos(ang), 0.0], dtype=np.float32)
