"""Properties checked at the full sizes of the benchmark configurations, where the CPU oracle is too slow to run:
permutation invariance, closest-hit / any-hit consistency, oracle agreement on a sample, additivity of sample ranges."""
import os

import numpy as np
import pytest

import oracle
from ray_tracer_amd import capi, scenes, synth

pytestmark = pytest.mark.gpu


def test_ten_million_triangle_soup_properties():
    """10 M triangles (sigma 0.002, the configuration of the traversal benchmark), 4 M incoherent rays"""
    ctx = capi.Context(0)
    mesh = synth.soup(10_000_000, 0.002, seed=12345)
    ctx.upload_mesh(mesh)  # above 2 M triangles the tree is built on the device
    st = ctx.stats()
    assert st["n_triangles"] == 10_000_000 and st["bvh_build_seconds"] < 2.0
    rays = synth.incoherent_rays(1 << 22, seed=777)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    hit = hits["prim"] != capi.MISS
    assert 0.4 < hit.mean() < 0.9
    assert (hits["tri"][hit] == hits["prim"][hit]).all()  # one triangle per polygon in a soup
    assert (t64[hit] >= 1e-3).all() and np.isinf(t64[~hit]).all()
    # the answer of a ray does not depend on its neighbours in the batch
    perm = np.random.default_rng(1).permutation(len(rays))
    h2, t2 = ctx.trace_closest(rays[perm], want_t64=True)
    assert np.array_equal(h2["prim"], hits["prim"][perm]) and np.array_equal(t2, t64[perm])
    # any hit up to t_max  <=>  the closest hit lies within t_max
    short = rays.copy()
    short[:, 7] = 0.25
    assert np.array_equal(ctx.trace_any(short).astype(bool), t64 <= np.float64(np.float32(0.25)))
    # a closest hit found with t_max = t is found again, and nothing closer exists
    again = rays[hit][:200000].copy()
    again[:, 7] = np.nextafter(hits["t"][hit][:200000].astype(np.float32), np.float32(np.inf))
    h3, t3 = ctx.trace_closest(again, want_t64=True)
    assert np.array_equal(h3["prim"], hits["prim"][hit][:200000]) and np.array_equal(t3, t64[hit][:200000])
    # the oracle's exact test on the reported triangle gives the reported t (a sample; the whole soup is beyond it)
    sub = np.nonzero(hit)[0][:2000]
    v = mesh["v"].reshape(-1, 3, 3)[hits["prim"][sub]]
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 1])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    o, d = rays[sub, 0:3].astype(np.float64), rays[sub, 4:7].astype(np.float64)
    t_plane = np.einsum("ij,ij->i", v[:, 0] - o, n) / np.einsum("ij,ij->i", d, n)
    assert np.allclose(t_plane, t64[sub], rtol=1e-9)
    ctx.close()


def test_full_hd_render_properties(cornell_dir):
    """1920x1080, 16 spp of the benchmark workload: sample ranges add up, the film is finite, and its mean agrees with the oracle's
    mean at low resolution (same scene and integrator; Monte-Carlo agreement, the seeds differ)"""
    nx, ny, ns = 1920, 1080, 16
    mesh, records, cam = scenes.cornell(nx, ny)
    ctx = capi.Context(0)
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, 99)
    ctx.render_pass(0, ns)
    ctx.render_sync()
    whole = ctx.download_film()
    ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, 99)
    ctx.render_pass(0, 5)
    ctx.render_pass(5, 11)
    ctx.render_sync()
    parts = ctx.download_film()
    assert np.isfinite(whole).all() and whole.min() >= 0.0
    assert np.allclose(whole, parts, rtol=1e-4, atol=1e-8)
    st = ctx.stats()
    assert st["camera_samples"] == nx * ny * ns
    # the same samples with nothing dropped (camera rays past the scene, hopeless connections, dead paths): the same film
    os.environ["MI355RT_TRACE_ALL_CONNECTIONS"] = "1"
    try:
        ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, 99)
        for s in range(0, ns, 4):
            ctx.render_pass(s, 4)
        everything = ctx.download_film()
        st_all = ctx.stats()
    finally:
        del os.environ["MI355RT_TRACE_ALL_CONNECTIONS"]
    assert np.allclose(whole, everything, rtol=1e-4, atol=1e-8)
    assert st_all["camera_samples"] == st["camera_samples"] and st_all["closest_rays"] > 2.0 * st["closest_rays"]
    mean = whole.sum() / (nx * ny)
    m2, r2, c2, oc = oracle.cornell_setup(cornell_dir, 192, 108)
    sc = oracle.Scene(m2)
    sc.set_materials(r2)
    ofilm, _ = sc.render(oc, 192, 108, 16, 30, oracle.ALG_MIS, seed=5)
    assert abs(mean - ofilm.sum() / (192 * 108)) < 0.05 * mean
    ctx.close()


def test_4k_render_runs():
    """3840x2160 (the largest configuration): indices and queue sizes hold, NEE+MIS and NaiveBDPT"""
    nx, ny = 3840, 2160
    mesh, records, cam = scenes.cornell(nx, ny)
    ctx = capi.Context(0)
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    means = {}
    for alg in (capi.ALG_MIS, capi.ALG_NAIVE_BDPT):
        ctx.render_begin(nx, ny, 2, 30, alg, 7)
        ctx.render_pass(0, 2)
        ctx.render_sync()
        film = ctx.download_film()
        assert np.isfinite(film).all()
        lit = np.nonzero(film.sum(1) > 0)[0]  # pixel index = column * ny + row; the box covers the middle columns of a 16:9 frame
        assert lit.min() // ny < nx // 3 and lit.max() // ny > 2 * nx // 3 and len(lit) > 0.3 * nx * ny
        means[alg] = film.sum() / (nx * ny)
    assert 0.05 < means[capi.ALG_MIS] < 0.09 and 1.3 < means[capi.ALG_NAIVE_BDPT] / means[capi.ALG_MIS] < 2.1
    ctx.close()
