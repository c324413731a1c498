"""Measures the summation-order noise floor of the REFERENCE on the benchmark's field and stores it as a fixture.

Runs only in the build container (needs oracle/_ref/libsph_ref.so = the reference's unmodified compute.cpp).
The reference's all-pairs algorithm is O(N^2), so the measurement uses the 40^3-particle corner of the dam-break
field (same lattice, jitter and seed as the 1M-particle benchmark block, same parameters), one time step:
the reference is run on the particles in their original order and on a random permutation of them, and the largest
difference between the two results is recorded for density, force density, position and velocity, each relative to
the field's largest magnitude.  Any implementation that sums the same terms in another order (the GPU path sums in
stencil order) cannot be expected to agree with the reference more closely than the reference agrees with itself;
the GPU tests at benchmark size state their force and velocity tolerances as multiples of these numbers.

Usage:  python tests/golden/make_noise_floor.py      ->  tests/golden/noise_dam_break.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import Ref  # noqa: E402
from sph_amd import scenes  # noqa: E402

SIDE = 40
SEEDS = (1, 2, 3)


def max_rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())


def main():
    if not Ref.available():
        raise SystemExit("oracle/_ref/libsph_ref.so missing: run `make -C oracle` where /root/reference exists")
    pos0, vel0, params = scenes.dam_break(SIDE)
    ref = Ref(params)
    pos_a, vel_a = pos0.copy(), vel0.copy()
    rho_a, f_a = ref.step(pos_a, vel_a, want_debug=True)
    out = {"field": f"dam_break({SIDE}): {SIDE}^3 jittered lattice at rest spacing, one reference step",
           "particles": int(pos0.shape[0]), "definition": "max |a - b| / max |b|, reference vs permuted reference",
           "permutations": []}
    worst = {"rho": 0.0, "force": 0.0, "pos": 0.0, "vel": 0.0}
    for seed in SEEDS:
        perm = np.random.default_rng(seed).permutation(pos0.shape[0])
        pos_b, vel_b = np.ascontiguousarray(pos0[perm]), np.ascontiguousarray(vel0[perm])
        rho_b, f_b = ref.step(pos_b, vel_b, want_debug=True)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        row = {"seed": seed, "rho": max_rel(rho_b[inv], rho_a), "force": max_rel(f_b[inv], f_a),
               "pos": max_rel(pos_b[inv], pos_a), "vel": max_rel(vel_b[inv], vel_a)}
        out["permutations"].append(row)
        for k in worst:
            worst[k] = max(worst[k], row[k])
        print(row, flush=True)
    out["noise"] = worst
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "noise_dam_break.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("noise floor:", worst)


if __name__ == "__main__":
    main()
