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


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (counter-based: no sequential state)."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def jitter_uniform(index, seed, axis):
    """Uniform [0,1) float64 that depends only on (particle index, seed, axis).

    Counter-based, so any rank can generate any subset of a field and get the same particles.
    """
    with np.errstate(over="ignore"):
        key = index.astype(np.uint64) * np.uint64(3) + np.uint64(axis) + np.uint64(seed) * np.uint64(0x100000001B3)
        bits = _splitmix64(_splitmix64(key))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def jittered_lattice(nx, ny, nz, spacing=REST_SPACING, jitter=0.1, origin=(0.0, 0.0, 0.0),
                     seed=12345, x_range=None, chunk=1 << 22, return_ids=False):
    """nx*ny*nz block on a simple-cubic lattice, each coordinate jittered by +-jitter*spacing.

    Particle (i,j,k) has index (i*ny + j)*nz + k (x-planes are contiguous: x is the slab
    axis) and sits at origin + (i+0.5, j+0.5, k+0.5)*spacing + jitter.  x_range=(i0, i1)
    generates only the planes i0 <= i < i1, bit-identical to the same particles of the full
    field.  Generated in chunks so that 16M+ particle fields need no full-size float64 copies.
    """
    i0, i1 = (0, nx) if x_range is None else x_range
    first, n = i0 * ny * nz, (i1 - i0) * ny * nz
    pos = np.empty((n, 3), dtype=np.float32)
    o = np.asarray(origin, dtype=np.float64)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = np.arange(first + a, first + b, dtype=np.int64)
        ijk = np.stack([idx // (ny * nz), (idx // nz) % ny, idx % nz], axis=1).astype(np.float64)
        jit = np.stack([jitter_uniform(idx, seed, ax) for ax in range(3)], axis=1) * 2.0 - 1.0
        pos[a:b] = (o + (ijk + 0.5) * spacing + jit * (jitter * spacing)).astype(np.float32)
    vel = np.zeros((n, 3), dtype=np.float32)
    if return_ids:
        return pos, vel, np.arange(first, first + n, dtype=np.int32)
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


def dam_break_params(side):
    """Parameters of the side^3 dam break: reference defaults, tank twice the block's width."""
    return default_params(tank_size=2.0 * side * REST_SPACING)


def dam_break_planes(side, x_range, seed=12345, jitter=0.1):
    """The lattice x-planes x_range=(i0, i1) of dam_break(side), with global particle ids."""
    return jittered_lattice(side, side, side, seed=seed, jitter=jitter, x_range=x_range, return_ids=True)


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
