"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref/libsph_ref.so, i.e. the reference's
unmodified compute/compute.cpp compiled by oracle/Makefile from /root/reference.  The GPU box
never runs this; it only reads the .npz files written here.

Every case stores
  pos0, vel0                       inputs (float32 [n,3])
  pos_k, vel_k for k in STEPS      state after k calls of the reference's updateParticles
  rho1, force1                     the reference's `densities` / `f_total` temporaries of step 1,
                                   tapped via --wrap=malloc/free (oracle/ref_tap.cpp)
  ncount0, cell0                   neighbour counts (sqrtf(r2) <= h, self included) and linear
                                   cell ids of the inputs.  The reference has neither notion, so
                                   these are derived by the oracle harness with the reference's
                                   r2 arithmetic (oracle/sph_oracle.c).
  noise_pos, noise_vel             reference-vs-permuted-reference max relative difference after
                                   the last stored step: the summation-order noise floor that
                                   calibrates tolerances for order-changing implementations.
kernels.npz holds the five smoothing-kernel functions sampled by calling the reference's own
Wpoly6/gradWspiky/... symbols.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import Oracle, Ref  # noqa: E402
from sph_amd import scenes  # noqa: E402
from sph_amd.params import default_params  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STEPS = (1, 2, 5, 10)


def cases():
    t = 0.2154
    yield "random_n1_s1", scenes.random_block(1, t, seed=1)
    yield "random_n2_s1", scenes.random_block(2, t, seed=1)
    # two particles closer than h, moving
    p = np.array([[0.05, 0.05, 0.05], [0.054, 0.052, 0.049]], np.float32)
    v = np.array([[0.1, 0.0, -0.2], [-0.3, 0.2, 0.0]], np.float32)
    yield "pair_close", (p, v)
    # coincident particles: vecNormalize(0) = 0 => no pressure force between them
    p = np.array([[0.05, 0.05, 0.05], [0.05, 0.05, 0.05], [0.053, 0.05, 0.05]], np.float32)
    yield "coincident", (p, np.zeros_like(p))
    for seed in (1, 2, 3):
        yield f"random_n64_s{seed}", scenes.random_block(64, t, seed=seed)
        # dense enough to have neighbours: 512 particles in a (tank/8)^3 box
        q, w = scenes.random_block(512, t / 4.0, seed=seed)
        yield f"random_n512_dense_s{seed}", (q, w)
    yield "random_n3000_s1", scenes.random_block(3000, t, seed=1)
    yield "lattice_n64_s1", scenes.jittered_lattice(4, 4, 4, seed=1)
    yield "lattice_n512_s2", scenes.jittered_lattice(8, 8, 8, seed=2)
    yield "lattice_n2940_s1", scenes.jittered_lattice(15, 14, 14, seed=1)
    yield "wall_n600_s7", scenes.wall_contact(600, t, seed=7)
    # moving block: non-zero velocities so that the viscosity term is exercised from step 1
    q, w = scenes.jittered_lattice(10, 10, 10, seed=5)
    rng = np.random.default_rng(5)
    w = ((rng.random(w.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.6)).astype(np.float32)
    yield "moving_n1000_s5", (q, w)


def main():
    if not Ref.available():
        raise SystemExit("oracle/_ref/libsph_ref.so missing: run `make -C oracle` where /root/reference exists")
    params = default_params()
    ref = Ref(params)
    orc = Oracle()
    for name, (pos0, vel0) in cases():
        pos0 = np.ascontiguousarray(pos0, np.float32)
        vel0 = np.ascontiguousarray(vel0, np.float32)
        n = pos0.shape[0]
        data = {"pos0": pos0, "vel0": vel0,
                "ncount0": orc.neighbor_counts(params, pos0), "cell0": orc.cell_ids(params, pos0)}
        pos, vel = pos0.copy(), vel0.copy()
        perm = np.random.default_rng(99).permutation(n)
        ppos, pvel = pos0[perm].copy(), vel0[perm].copy()
        for k in range(1, max(STEPS) + 1):
            if k == 1:
                rho, force = ref.step(pos, vel, want_debug=True)
                data["rho1"], data["force1"] = rho, force
            else:
                ref.step(pos, vel)
            ref.step(ppos, pvel)
            if k in STEPS:
                data[f"pos_{k}"], data[f"vel_{k}"] = pos.copy(), vel.copy()
                inv = np.empty(n, np.int64)
                inv[perm] = np.arange(n)
                dp = np.abs(ppos[inv] - pos).max() / max(np.abs(pos).max(), 1e-30)
                dv = np.abs(pvel[inv] - vel).max() / max(np.abs(vel).max(), 1e-30)
                data[f"noise_pos_{k}"], data[f"noise_vel_{k}"] = np.float64(dp), np.float64(dv)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
        print(f"{name:28s} n={n:5d} ncount mean {data['ncount0'].mean():6.2f} max {data['ncount0'].max():3d} "
              f"noise pos@10 {data['noise_pos_10']:.2e} vel@10 {data['noise_vel_10']:.2e}")

    # smoothing kernels sampled through the reference's own symbols
    h = np.float32(params.h)
    r = np.concatenate([
        np.linspace(0.0, float(h), 257, dtype=np.float32),
        np.array([h, np.nextafter(h, np.float32(0)), np.nextafter(h, np.float32(1)), 0.0101, 0.02, 1.0], np.float32),
    ]).astype(np.float32)
    k = {"r": r, "consts": ref.consts()}
    for name in ("lapl_viscosity", "grad_spiky", "w_poly6", "grad_poly6", "lapl_poly6"):
        k[name] = np.array([ref.kernel(name, float(x)) for x in r], np.float32)
    vn_in = np.array([[3, 4, 0], [0, 0, 0], [1e-20, 0, 0], [-1, 2, -2], [0.003, -0.004, 0.0051]], np.float32)
    k["vecnorm_in"] = vn_in
    k["vecnorm_out"] = np.stack([ref.vec_normalize(a) for a in vn_in])
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **k)
    print("kernels.npz written:", len(r), "samples per kernel")


if __name__ == "__main__":
    main()
