"""Bilateral filter (reference filter.cpp:13-51).  The reference file compiles unmodified, so this row is pinned by it:
tests/golden/filter_vectors.npz holds its outputs (generator tests/golden/make_golden_filter.py).
Tolerance: the filter is a ratio of sums of exp() values; libm's and the device's exp may differ by one unit in the last
place each, and the exponent (down to about -700) is formed with a reciprocal multiply on the device: 1e-12 relative."""
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-12


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "filter_vectors.npz"))


def test_oracle_restatement_is_bit_identical_to_the_reference(golden):
    for k, (w, h, win, sd, sr) in enumerate(golden["cases"]):
        got = oracle.bilateral(golden["img%d" % k], int(win), sd, sr)
        assert np.array_equal(got, golden["res%d" % k]), "case %d" % k
    L = oracle.ref_filter()
    if L is not None:  # where the compiled reference is present: one more image, directly
        img = np.random.default_rng(1).random((11, 13, 4))
        assert np.array_equal(oracle.bilateral(img, 2, 1.5, 0.3), oracle.bilateral(img, 2, 1.5, 0.3, fn=L.ref_bilateral))


def test_oracle_edge_cases():
    img = np.random.default_rng(2).random((6, 5, 4))
    same = oracle.bilateral(img, 0, 1.0, 1.0)
    assert np.array_equal(same[..., :3], img[..., :3]) and (same[..., 3] == 1.0).all()  # window 0: the pixel itself, weight exp(0)
    assert np.isnan(oracle.bilateral(img, 1, 0.0, 1.0)[..., :3]).all()  # sigma_d = 0: 0/0 in the exponent, as in the reference
    flat = np.full((4, 4, 4), 0.25)
    assert np.allclose(oracle.bilateral(flat, 2, 1.0, 0.1)[..., :3], 0.25, rtol=1e-15)


@pytest.mark.gpu
def test_gpu_filter_against_golden_and_oracle(golden):
    from ray_tracer_amd import capi
    from ray_tracer_amd.filter import BiliteralFilter
    ctx = capi.Context(0)
    for k, (w, h, win, sd, sr) in enumerate(golden["cases"]):
        got = ctx.filter_bilateral(golden["img%d" % k], int(win), sd, sr)
        want = golden["res%d" % k]
        assert got.shape == want.shape and np.allclose(got, want, rtol=RTOL, atol=0.0), "case %d: max rel %g" % (k, np.abs(got / want - 1).max())
        assert (got[..., 3] == 1.0).all()
    img = np.random.default_rng(5).random((45, 70, 4))
    f = BiliteralFilter(ctx, img.reshape(-1), 70, 45)  # the reference's interface: flat vector, width, height
    f.sigma_d, f.sigma_r, f.window = 2.5, 0.15, 5
    f.FilterImage()
    assert np.allclose(f.result.reshape(45, 70, 4), oracle.bilateral(img, 5, 2.5, 0.15), rtol=RTOL, atol=0.0)
    assert np.isnan(ctx.filter_bilateral(img, 1, 0.0, 1.0)[..., :3]).all()
    with pytest.raises(capi.BackendError):
        ctx.filter_bilateral(img, -1, 1.0, 1.0)
    ctx.close()
