"""Generates tests/golden/vec3_vectors.npz from the reference's vec3.h and ray.h compiled unmodified into
oracle/_ref/libref_core.so (`make -C oracle`, only where the reference checkout exists).  Run from the repository root:

    python tests/golden/make_golden_vec3.py

Inputs are drawn here (seeded); outputs are what the reference's own operators, dot, cross, unit_vector, rodrigues and
ray::point_at_parameter return for them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402

p = oracle._p


def main():
    core = oracle.ref_core()
    assert core is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261112)
    n = 2000
    a = rng.normal(size=(n, 3)) * rng.choice([1e-6, 1.0, 1e3], (n, 1))
    b = rng.normal(size=(n, 3)) * rng.choice([1e-3, 1.0, 10.0], (n, 1))
    a[:6] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [3, 4, 0], [1e-300, 0, 0]])
    b[:6] = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0], [0, 0, -2], [0, 0, 5], [1, 1, 1]])
    t = rng.normal(size=n) * rng.choice([1e-3, 1.0, 7.0], n)
    out = np.zeros((n, 33))
    pt = np.zeros((n, 3))
    for i in range(n):
        core.ref_vec3_ops(p(a[i]), p(b[i]), t[i], p(out[i]))
        core.ref_ray_point(p(a[i]), p(b[i]), t[i], p(pt[i]))
    np.savez_compressed(os.path.join(HERE, "vec3_vectors.npz"), a=a, b=b, t=t, ops=out, point=pt)
    print("wrote vec3_vectors.npz")


if __name__ == "__main__":
    main()
