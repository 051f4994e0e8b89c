"""Records in tests/golden/filter_vectors.npz what the reference's own filter.cpp (compiled unmodified into
oracle/_ref/libref_filter.so) returns for seeded random images.  Run from the repository root, where the reference
checkout is:  python tests/golden/make_golden_filter.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402

CASES = [  # width, height, window, sigma_d, sigma_r
    (24, 16, 3, 2.0, 0.2), (17, 9, 0, 1.0, 1.0), (5, 7, 6, 3.0, 0.05), (32, 32, 2, 0.7, 10.0), (1, 1, 2, 1.0, 1.0), (40, 3, 4, 20.0, 1.0),
]


def main():
    L = oracle.ref_filter()
    assert L is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261111)
    out = {"cases": np.array(CASES, dtype=np.float64)}
    for k, (w, h, win, sd, sr) in enumerate(CASES):
        img = rng.random((h, w, 4))
        img[..., 3] = rng.random((h, w))  # alpha is ignored and overwritten with 1
        out["img%d" % k] = img
        out["res%d" % k] = oracle.bilateral(img, win, sd, sr, fn=L.ref_bilateral)
    np.savez_compressed(os.path.join(HERE, "filter_vectors.npz"), **out)
    print("wrote filter_vectors.npz")


if __name__ == "__main__":
    main()
