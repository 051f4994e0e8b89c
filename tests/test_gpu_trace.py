"""Closest-hit / any-hit parity of the HIP traversal kernels against the CPU oracle, through the C ABI.

Bar: primitive ids identical and t bit-identical in fp64, except exact ties in t between different
polygons (the reference resolves those by its tree shape, bvh.cpp:83-89; the device picks the lowest
fan-triangle index), which are counted and must be exactly tied in the oracle as well.
"""
import os

import numpy as np
import pytest

import oracle
from ray_tracer_amd import capi, scene_io, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def check_closest(hits, t64, o_poly, o_tri, o_t, sc=None, rays=None, allow_ties=True):
    g_poly = hits["prim"].astype(np.int64)
    g_poly[hits["prim"] == capi.MISS] = -1
    same = g_poly == o_poly
    if not same.all():
        bad = np.nonzero(~same)[0]
        # every disagreement must be an exact tie in t (both are hits with identical t)
        assert allow_ties, "prim mismatch at rays %s" % bad[:10]
        assert np.array_equal(t64[bad], o_t[bad]), "prim mismatch that is not a t-tie at rays %s" % bad[:10]
        assert len(bad) <= max(1, len(g_poly) // 100000), "too many ties: %d" % len(bad)
    assert np.array_equal(t64, o_t), "t differs from the oracle's fp64 t"
    hit = o_poly >= 0
    assert np.array_equal(hits["t"][hit], o_t[hit].astype(np.float32))
    assert np.all(np.isinf(hits["t"][~hit]))
    return int((~same).sum())


def cornell_rays(n, seed):
    rays = synth.incoherent_rays(n, seed=seed)
    rays[:, 0:3] *= 0.95  # origins inside the box
    return rays


def test_cornell_closest_vs_reference_bvh(ctx, cornell_dir):
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 64, 64)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = cornell_rays(200000, 777)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_BVH_REFERENCE)
    check_closest(hits, t64, o_poly, o_tri, o_t)
    same = hits["prim"].astype(np.int64) == o_poly
    assert np.array_equal(hits["tri"][same & (o_poly >= 0)], o_tri[same & (o_poly >= 0)].astype(np.uint32))
    # object ids follow the obj object order
    ob = mesh["object_begin"].astype(np.int64)
    hit = o_poly >= 0
    assert np.array_equal(hits["object"][hit], (np.searchsorted(ob, o_poly[hit], side="right") - 1).astype(np.uint32))
    assert 0.5 < hit.mean() <= 1.0


def test_cornell_quads_fan_triangles(ctx, cornell_dir):
    """cornell_box.obj holds quads: prim is the polygon, tri the fan triangle that accepted the hit"""
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 64, 64, obj_name="cornell_box.obj")
    sc = oracle.Scene(mesh)
    assert sc.n_triangles > sc.n_polygons
    ctx.upload_mesh(mesh)
    rays = cornell_rays(100000, 3)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_BVH_REFERENCE)
    check_closest(hits, t64, o_poly, o_tri, o_t)
    hit = (o_poly >= 0) & (hits["prim"].astype(np.int64) == o_poly)
    assert np.array_equal(hits["tri"][hit], o_tri[hit].astype(np.uint32))


def test_cornell_any_hit(ctx, cornell_dir):
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 64, 64)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = cornell_rays(100000, 11)
    rays[:, 7] = 0.5
    got = ctx.trace_any(rays)
    want = sc.trace_any(rays, oracle.MODE_BVH_REFERENCE)
    assert np.array_equal(got, want)
    assert 0.05 < want.mean() < 0.95


@pytest.mark.parametrize("n,sigma", [(1000, 0.05), (100000, 0.01)])
def test_soup_closest(ctx, n, sigma):
    mesh = synth.soup(n, sigma, seed=12345)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    nodes, tri_index = ctx.bvh2()
    assert sorted(tri_index.tolist()) == list(range(n))
    rays = synth.incoherent_rays(100000, seed=777)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    # the pruned, ordered walk over the product's own tree with the reference triangle test ...
    o_poly, o_tri, o_t, nv, nt = sc.trace_bvh2(nodes, tri_index, rays)
    check_closest(hits, t64, o_poly, o_tri, o_t, allow_ties=False)
    assert np.array_equal(hits["tri"][o_poly >= 0], o_tri[o_poly >= 0].astype(np.uint32))
    # ... and, on a subset, the reference's own traversal (unpruned, its SAH tree)
    sub = rays[:5000] if n > 1000 else rays[:20000]
    r_poly, r_tri, r_t = sc.trace_closest(sub, oracle.MODE_BVH_REFERENCE)
    check_closest(hits[:len(sub)], t64[:len(sub)], r_poly, r_tri, r_t)


def test_soup_any_hit(ctx):
    mesh = synth.soup(100000, 0.01, seed=12345)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = synth.incoherent_rays(20000, seed=5, t_max=0.5)
    got = ctx.trace_any(rays)
    want = sc.trace_any(rays, oracle.MODE_LIST)
    assert np.array_equal(got, want)


def test_edge_cases(ctx):
    # one triangle, rays straight at it, past it, parallel to it, axis-aligned directions with zero components
    mesh = {
        "v": np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float64), "vt": np.zeros((0, 3)),
        "vn": np.array([[0, 0, 1.0]]), "corners": np.array([[0, -1, 0], [1, -1, 0], [2, -1, 0]], dtype=np.int64),
        "face_begin": np.array([0, 3], dtype=np.uint64), "object_begin": np.array([0, 1], dtype=np.uint64), "names": ["t"],
    }
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    inf = np.inf
    rays = np.array([
        [0.25, 0.25, 1, 1e-3, 0, 0, -1, inf],    # hit at t = 1
        [0.25, 0.25, 1, 1e-3, 0, 0, -1, 0.5],    # t_max too short
        [0.25, 0.25, 1, 1e-3, 0, 0, 1, inf],     # pointing away
        [0.25, 0.25, 1, 1e-3, 1, 0, 0, inf],     # parallel to the plane
        [0.25, 0.25, -1, 1e-3, 0, 0, 1, inf],    # from behind: no back-face culling in the reference
        [2.0, 2.0, 1, 1e-3, 0, 0, -1, inf],      # outside
        [0.0, 0.25, 1, 1e-3, 0, 0, -1, inf],     # exactly on an edge: strict test misses (object.cpp:253)
        [0.25, 0.25, 1, 1.0, 0, 0, -1, inf],     # t == t_min is accepted
        [0.25, 0.25, 1, 1e-3, 0, 0, -1, 1.0],    # t == t_max is accepted
    ], dtype=np.float32)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST)
    check_closest(hits, t64, o_poly, o_tri, o_t)
    assert o_poly.tolist() == [0, -1, -1, -1, 0, -1, -1, 0, 0]
    assert np.array_equal(ctx.trace_any(rays), (o_poly >= 0).astype(np.uint8))
    # zero rays
    assert len(ctx.trace_closest(np.zeros((0, 8), dtype=np.float32))) == 0


def test_errors(ctx):
    bad = {
        "v": np.zeros((3, 3)), "vt": np.zeros((0, 3)), "vn": np.zeros((0, 3)),
        "corners": np.array([[0, -1, 0], [1, -1, 0], [2, -1, 0]], dtype=np.int64),
        "face_begin": np.array([0, 3], dtype=np.uint64), "object_begin": np.array([0, 1], dtype=np.uint64),
    }
    with pytest.raises(capi.BackendError) as e:
        ctx.upload_mesh(bad)  # no vn: the reference dereferences it (object.cpp:115)
    assert e.value.code == capi.ERR_MESH
    with pytest.raises(capi.BackendError):
        ctx.trace_closest(np.zeros((1, 8), dtype=np.float32))  # no mesh after the failed upload
