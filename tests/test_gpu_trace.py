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
    ctx.set_builder(capi.BUILDER_HOST_SAH)  # the binary tree is kept by the host builder only
    ctx.upload_mesh(mesh)
    ctx.set_builder(capi.BUILDER_AUTO)
    nodes, tri_index = ctx.bvh2()
    assert sorted(set(tri_index.tolist())) == list(range(n))  # a triangle whose reference was cut appears once per piece
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


@pytest.mark.parametrize("builder", [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH])
def test_wide_tree_export_walked_by_the_oracle(ctx, cornell_dir, builder):
    """rt_get_wide_bvh hands out the 8-wide tree the kernels walk; the oracle's own walk over it (exact boxes, the reference's triangle
    test) finds the hits the GPU reports, and its visit counts are the per-ray work figures of the benchmark"""
    for mesh, rays in ((oracle.cornell_setup(cornell_dir, 8, 8)[0], None), (synth.soup(50000, 0.01, seed=3), None)):
        ctx.set_builder(builder)
        ctx.upload_mesh(mesh)
        ctx.set_builder(capi.BUILDER_AUTO)
        nodes, tri_index = ctx.wide_bvh()
        n_tris = int(ctx.stats()["n_triangles"])
        assert len(nodes) >= 1 and sorted(tri_index.tolist()) == list(range(n_tris))
        # every triangle sits in exactly one leaf child
        counts = (nodes["meta"] >> 5).astype(np.int64)
        leaf = ((nodes["imask"][:, None] >> np.arange(8)) & 1) == 0
        assert (counts * leaf).sum() == n_tris
        rays = synth.incoherent_rays(60000, seed=777)
        rays[:, 0:3] *= 0.95
        sc = oracle.Scene(mesh)
        hits, t64 = ctx.trace_closest(rays, want_t64=True)
        poly, tri, t, nv, nt = sc.trace_wide(nodes, tri_index, rays)
        check_closest(hits, t64, poly, tri, t, allow_ties=False)
        assert 1.0 <= nv / len(rays) < 60 and nt / len(rays) < 20


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


def _axis_rays(n, seed, scale=1.0, offset=0.0):
    """rays built to stress the conservative box tests: zero direction components (positive and negative zero),
    origins on round coordinates, tiny and huge direction magnitudes"""
    rng = np.random.default_rng(seed)
    rays = synth.incoherent_rays(n, seed=seed)
    k = np.arange(n)
    rays[k % 5 == 0, 4] = 0.0
    rays[k % 7 == 0, 5] = -0.0
    rays[k % 11 == 0, 6] = 0.0
    rays[k % 13 == 0, 0:3] = np.round(rays[k % 13 == 0, 0:3] * 8) / 8
    rays[k % 17 == 0, 4:7] *= 1e-3
    rays[k % 19 == 0, 4:7] *= 1e3
    bad = np.abs(rays[:, 4:7]).sum(1) == 0
    rays[bad, 4] = 1.0
    rays[:, 0:3] = rays[:, 0:3] * scale + offset
    rays[:, 3] = 1e-3
    return rays


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1000.0, 5000.0), (1e-3, 0.25)])
def test_conservative_boxes_against_brute_force(ctx, scale, offset):
    """the quantised, outward-rounded boxes may only over-accept: every hit the reference's linear scan
    (HittableList::Hit, no boxes at all) finds must be found, for awkward rays and coordinate ranges"""
    mesh = synth.soup(4000, 0.08, seed=99)
    mesh["v"] = mesh["v"] * scale + offset
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = _axis_rays(30000, 5, scale, offset)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    check_closest(hits, t64, o_poly, o_tri, o_t, allow_ties=False)
    assert 0.3 < (o_poly >= 0).mean()
    short = rays.copy()
    short[:, 7] = 0.3 * scale
    assert np.array_equal(ctx.trace_any(short), sc.trace_any(short, oracle.MODE_LIST))


def test_axis_aligned_geometry_and_rays(ctx, cornell_dir):
    """Cornell walls are axis-aligned (zero-thickness boxes); rays along the axes graze them"""
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 64, 64)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = _axis_rays(40000, 8, 0.95)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    check_closest(hits, t64, o_poly, o_tri, o_t)


def test_deep_tree_uses_the_scratch_stack(ctx):
    """a long thin chain of triangles gives a tree deeper than the LDS stack holds; the scratch variant must agree"""
    n = 3000
    x = np.cumsum(np.geomspace(1e-6, 1.0, n))  # sizes growing geometrically: SAH splits one triangle off at a time
    v = np.zeros((n, 3, 3))
    v[:, 0] = np.stack([x, np.zeros(n), np.zeros(n)], 1)
    v[:, 1] = np.stack([x + np.geomspace(1e-6, 1.0, n) * 0.9, np.zeros(n), np.zeros(n)], 1)
    v[:, 2] = np.stack([x, np.geomspace(1e-6, 1.0, n), np.zeros(n)], 1)
    idx = np.arange(3 * n, dtype=np.int64)
    mesh = {"v": v.reshape(-1, 3), "vt": np.zeros((0, 3)), "vn": np.tile([[0, 0, 1.0]], (n, 1)),
            "corners": np.stack([idx, np.full(3 * n, -1), idx // 3], 1), "face_begin": np.arange(0, 3 * n + 1, 3, dtype=np.uint64),
            "object_begin": np.array([0, n], dtype=np.uint64), "names": ["chain"]}
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    assert ctx.stats()["bvh_depth"] > 16  # more levels than the LDS stack holds
    rng = np.random.default_rng(3)
    m = 20000
    rays = np.zeros((m, 8), dtype=np.float32)
    t = rng.integers(0, n, m)
    p = v[t, 0] + 0.3 * (v[t, 1] - v[t, 0]) + 0.3 * (v[t, 2] - v[t, 0])
    h = np.geomspace(1e-6, 1.0, n)[t][:, None]
    rays[:, 0:3] = p + np.array([0, 0, 1.0]) * h
    rays[:, 3] = 1e-3
    rays[:, 4:7] = (np.array([0, 0, -1.0]) + rng.normal(size=(m, 3)) * 0.01) * h
    rays[:, 7] = np.inf
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    check_closest(hits, t64, o_poly, o_tri, o_t)
    assert (o_poly >= 0).mean() > 0.5


def test_unsound_rays_miss_without_walking_the_tree(ctx):
    """NaN components and zero directions: the exact test can accept nothing (the oracle agrees), and the kernel must
    not pay a whole-tree walk for them (the conservative box tests accept everything for such rays)"""
    import time
    mesh = synth.soup(200000, 0.01, seed=5)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = synth.incoherent_rays(1 << 18, seed=9)
    t0 = time.time()
    ctx.trace_closest(rays)
    base = time.time() - t0
    bad = rays.copy()
    k = np.arange(len(bad))
    bad[k % 4 == 0, 4] = np.nan
    bad[k % 4 == 1, 1] = np.nan
    bad[k % 4 == 2, 4:7] = 0.0
    bad[k % 4 == 3, 3] = np.nan
    t0 = time.time()
    hits, t64 = ctx.trace_closest(bad, want_t64=True)
    took = time.time() - t0
    assert (hits["prim"] == capi.MISS).all() and np.isinf(t64).all()
    assert took < 20 * base + 0.5, (took, base)  # a whole-tree walk per ray would take minutes
    o_poly, o_tri, o_t = sc.trace_closest(bad[:3000], oracle.MODE_LIST_LOWEST)
    assert (o_poly == -1).all()
    assert not ctx.trace_any(bad[:3000]).any() and not sc.trace_any(bad[:3000], oracle.MODE_LIST).any()


def test_axis_parallel_rays_are_pruned_like_any_other(ctx):
    """rays that do not move along one or two axes (mirror bounces off axis-aligned walls produce them) must still be
    culled by the boxes on those axes: same answers as brute force, and no whole-tree walks"""
    import time
    mesh = synth.soup(200000, 0.01, seed=5)
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    rays = synth.incoherent_rays(1 << 18, seed=9)
    t0 = time.time()
    ctx.trace_closest(rays)
    base = time.time() - t0
    ax = rays.copy()
    k = np.arange(len(ax))
    for a in range(3):
        sel = k % 6 == a
        ax[sel, 4:7] = 0.0
        ax[sel, 4 + a] = np.where(k[sel] % 12 < 6, 1.0, -1.0)          # along one axis
        sel2 = k % 6 == 3 + a
        ax[sel2, 4 + a] = np.where(k[sel2] % 12 < 6, 0.0, -0.0)         # in an axis plane
    ax[k % 50 == 0, 4] = 1e-30                                           # inverse overflows in fp32
    ax[:, 0:3] = np.round(ax[:, 0:3] * 64) / 64                          # origins on round coordinates: box planes are met exactly
    t0 = time.time()
    hits, t64 = ctx.trace_closest(ax, want_t64=True)
    took = time.time() - t0
    assert took < 5 * base + 0.2, (took, base)
    n = 4000
    o_poly, o_tri, o_t = sc.trace_closest(ax[:n], oracle.MODE_LIST_LOWEST)
    check_closest(hits[:n], t64[:n], o_poly, o_tri, o_t, allow_ties=False)
    assert 0.3 < (o_poly >= 0).mean()


@pytest.mark.parametrize("builder", [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH])
def test_rays_ordered_by_cell_give_the_same_answers(ctx, builder, monkeypatch):
    """Large batches in scenes far beyond the caches are traced in the order of the cells their origins lie in and the results are
    put back in the caller's order; MI355RT_RAY_BINS=1 asks for that on any batch.  Same ids and the same fp64 t as without, ray by
    ray, also for origins far outside the scene, NaN origins and a batch that is not a multiple of anything."""
    mesh = synth.soup(50000, 0.02, seed=31)
    ctx.set_builder(builder)
    ctx.upload_mesh(mesh)
    ctx.set_builder(capi.BUILDER_AUTO)
    rays = synth.incoherent_rays(100003, seed=41)
    k = np.arange(len(rays))
    rays[k % 50 == 0, 0:3] *= 1000.0          # far outside the root box
    rays[k % 97 == 0, 1] = np.nan             # unsound rays: answered "miss" wherever they are filed
    rays[k % 101 == 0, 0] = np.inf
    monkeypatch.setenv("MI355RT_RAY_BINS", "0")
    h0, t0 = ctx.trace_closest(rays, want_t64=True)
    monkeypatch.setenv("MI355RT_RAY_BINS", "1")
    h1, t1 = ctx.trace_closest(rays, want_t64=True)
    h2 = ctx.trace_closest(rays[:777])        # a smaller batch through the same buffers, without t64
    assert (h0["prim"] != capi.MISS).sum() > 10000
    assert np.array_equal(h0, h1) and np.array_equal(t0, t1)
    assert np.array_equal(h0[:777], h2)
    a1 = ctx.trace_any(rays)
    monkeypatch.setenv("MI355RT_RAY_BINS", "0")
    a0 = ctx.trace_any(rays)
    assert a0.sum() > 10000 and np.array_equal(a0, a1)


def polygon_mesh(polys):
    """rt_mesh tables of one object from a list of polygons (each an [k, 3] array), one flat normal per polygon"""
    v = np.concatenate(polys).astype(np.float64)
    counts = np.array([len(p) for p in polys])
    begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    nrm = np.array([np.cross(p[1] - p[0], p[2] - p[1]) for p in polys])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    idx = np.arange(len(v), dtype=np.int64)
    corners = np.stack([idx, np.full(len(v), -1, dtype=np.int64), np.repeat(np.arange(len(polys)), counts)], axis=1)
    return {"v": v, "vt": np.zeros((0, 3)), "vn": nrm, "corners": corners, "face_begin": begin,
            "object_begin": np.array([0, len(polys)], dtype=np.uint64), "names": ["warped"]}


def warped_scene(seed):
    """sheets of quads displaced out of their plane by 1e-6 ... 1e-1 of an edge, and pentagons / octagons (fan triangles with obtuse
    corners) whose vertices are moved off their plane: polygons for which the reference's test (every fan triangle on the plane of
    the polygon's first three vertices, object.cpp:244) accepts points outside the fan triangle and outside its vertices' box"""
    rng = np.random.default_rng(seed)
    polys = []
    n = 40
    for level, dev in enumerate([1e-6, 1e-4, 1e-2, 1e-1]):
        u, w = np.meshgrid(np.linspace(-1, 1, n + 1), np.linspace(-1, 1, n + 1), indexing="ij")
        edge = 2.0 / n
        y = -0.75 + 0.5 * level + dev * edge * rng.normal(size=u.shape)
        p = np.stack([u, y, w], -1)
        for i in range(n):
            for j in range(n):
                polys.append(np.array([p[i, j], p[i + 1, j], p[i + 1, j + 1], p[i, j + 1]]))
    for k in range(1500):
        m = int(rng.choice([5, 8]))
        c, r = rng.uniform(-0.9, 0.9, 3), rng.uniform(0.02, 0.12)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        ang = np.sort(rng.uniform(0, 2 * np.pi, m))  # irregular: some fan triangles are slivers
        ring = np.stack([np.cos(ang), np.sin(ang), 10.0 ** rng.uniform(-5, -0.7) * rng.normal(size=m)], -1) * r
        polys.append(c + ring @ q.T)
    return polygon_mesh(polys)


@pytest.mark.parametrize("builder", [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH])
def test_polygons_that_are_not_planar_have_the_brute_force_answer(ctx, builder):
    """For a polygon that is not planar the hit point of a later fan triangle need not lie in that triangle's box.  The builders box
    everything the reference's test can accept (accept_region.h), so whatever the tree and whatever the order in which waves prune,
    the closest hit is the one a loop over all polygons finds (HittableList::Hit, hittablelist.cpp:3-17): ids and fp64 t equal to
    the oracle's brute-force loop, for either builder, twice in a row, and for the any-hit query."""
    mesh = warped_scene(17)
    sc = oracle.Scene(mesh)
    rng = np.random.default_rng(5)
    rays = synth.incoherent_rays(60000, seed=23)
    graze = synth.incoherent_rays(40000, seed=24)  # nearly parallel to the sheets: long walks just above and below them
    graze[:, 5] *= 0.02
    aimed = synth.incoherent_rays(20000, seed=25)  # towards points next to the corners of the many-sided polygons
    v = mesh["v"]
    target = v[rng.integers(4 * 40 * 40 * 4, len(v), len(aimed))]  # corners of the many-sided polygons (they follow the four sheets) + rng.normal(size=(len(aimed), 3)) * 0.01
    aimed[:, 4:7] = (target - aimed[:, 0:3]).astype(np.float32)
    rays = np.concatenate([rays, graze, aimed])
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    assert (o_poly >= 0).mean() > 0.5
    ctx.set_builder(builder)
    runs = []
    for rep in range(2):
        ctx.upload_mesh(mesh)
        hits, t64 = ctx.trace_closest(rays, want_t64=True)
        gp = hits["prim"].astype(np.int64)
        gp[hits["prim"] == capi.MISS] = -1
        assert np.array_equal(t64, o_t), "t differs from the loop over all polygons on %d rays" % (t64 != o_t).sum()
        assert (gp != o_poly).sum() <= 2  # exact ties in t between neighbours
        runs.append((hits.copy(), t64.copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    # the oracle's distance-ordered, pruning walk of the same tree agrees as well: the boxes hold the hits
    wn, wt = ctx.wide_bvh()
    w_poly, _, w_t, _, _ = sc.trace_wide(wn, wt, rays)
    assert np.array_equal(w_t, o_t)
    short = rays.copy()
    short[:, 7] = 0.4
    assert np.array_equal(ctx.trace_any(short), sc.trace_any(short, oracle.MODE_LIST))
    ctx.set_builder(capi.BUILDER_AUTO)
    # and the hits outside the vertices' boxes exist in this scene: the brute-force loop finds hit points that lie outside the box
    # of the fan triangle they are reported for
    hit = o_tri >= 0
    p = rays[hit, 0:3].astype(np.float64) + o_t[hit, None] * rays[hit, 4:7].astype(np.float64)
    tri_v = fan_triangle_vertices(mesh, o_tri[hit])
    lo, hi = tri_v.min(1) - 1e-7, tri_v.max(1) + 1e-7
    assert (np.any((p < lo) | (p > hi), axis=1)).sum() > 0


def long_polygon_scene(seed):
    """small clutter with long thin triangles and long, slightly warped quads lying diagonally through it: the polygons whose
    references the builders cut into pieces (csrc/presplit.h)"""
    rng = np.random.default_rng(seed)
    polys = []
    for k in range(120):
        a = rng.uniform(-1, 1, 3)
        d = rng.normal(size=3)
        d *= rng.uniform(0.5, 2.0) / np.linalg.norm(d)
        w = np.cross(d, rng.normal(size=3))
        w *= rng.uniform(0.002, 0.02) / np.linalg.norm(w)
        if k % 2:
            polys.append(np.stack([a, a + d, a + d + w]))
        else:  # a quad whose last vertices leave the plane of the first three, some corners obtuse
            n = np.cross(d, w)
            n /= np.linalg.norm(n)
            skew = rng.uniform(-0.2, 0.2) * d
            polys.append(np.stack([a, a + d, a + d + w + skew + 10.0 ** rng.uniform(-9, -1.5) * np.linalg.norm(w) * n, a + w + skew]))
    c = rng.uniform(-1, 1, (9000, 1, 3))
    polys += list(c + rng.normal(size=(9000, 3, 3)) * 0.004)
    return polygon_mesh(polys), 120


@pytest.mark.parametrize("builder", [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH])
def test_cut_references_have_the_brute_force_answer(ctx, builder):
    """Long triangles enter the tree as several references, each with the box of a piece.  The closest hit must still be the one a loop
    over all polygons finds (hittablelist.cpp:3-17) -- for rays aimed along and across the long polygons in particular, whose hit
    points lie next to the planes the references were cut at -- and must not depend on whether anything was cut."""
    mesh, n_long = long_polygon_scene(31)
    sc = oracle.Scene(mesh)
    rng = np.random.default_rng(8)
    rays = synth.incoherent_rays(30000, seed=41)
    aimed = synth.incoherent_rays(50000, seed=42)
    tri_v = fan_triangle_vertices(mesh, rng.integers(0, n_long + n_long // 2, len(aimed)))  # the long polygons' fan triangles come first
    u = rng.dirichlet([0.7, 0.7, 0.7], len(aimed))
    u[::5] = np.round(u[::5])                                   # every fifth ray at a vertex
    u[1::5, 2] = 0.0
    u[1::5] /= u[1::5].sum(1, keepdims=True)                    # and every fifth at an edge
    target = np.einsum("ij,ijk->ik", u, tri_v)
    aimed[:, 4:7] = (target - aimed[:, 0:3]).astype(np.float32)
    along = synth.incoherent_rays(20000, seed=43)               # rays that run along a long polygon, just above it
    tv = fan_triangle_vertices(mesh, rng.integers(0, n_long + n_long // 2, len(along)))
    along[:, 0:3] = (tv[:, 0] + 1e-3 * rng.normal(size=(len(along), 3))).astype(np.float32)
    along[:, 4:7] = (tv[:, 1] - tv[:, 0] + 0.02 * rng.normal(size=(len(along), 3))).astype(np.float32)
    rays = np.concatenate([rays, aimed, along])
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    assert (o_poly >= 0).mean() > 0.35 and ((o_poly >= 0) & (o_poly < n_long)).mean() > 0.3
    ctx.set_builder(builder)
    results = []
    for budget in (0.25, 0.0, 4.0):
        ctx.set_reference_cutting(budget)
        ctx.upload_mesh(mesh)
        n_cut, n_extra = ctx.reference_cutting()
        if budget == 0.0:
            assert n_cut == 0 and n_extra == 0
        else:
            assert n_cut >= n_long // 2 and n_extra >= 4 * n_cut
            wn, wt = ctx.wide_bvh()
            assert len(wt) == sc.n_triangles + n_extra and set(wt.tolist()) == set(range(sc.n_triangles))
        hits, t64 = ctx.trace_closest(rays, want_t64=True)
        gp = hits["prim"].astype(np.int64)
        gp[hits["prim"] == capi.MISS] = -1
        assert np.array_equal(t64, o_t), "t differs from the loop over all polygons on %d rays" % (t64 != o_t).sum()
        assert (gp != o_poly).sum() <= 2  # exact ties in t
        results.append((hits.copy(), t64.copy()))
        short = rays.copy()
        short[:, 7] = 0.6
        assert np.array_equal(ctx.trace_any(short), sc.trace_any(short, oracle.MODE_LIST))
    for h, t in results[1:]:
        assert np.array_equal(h, results[0][0]) and np.array_equal(t, results[0][1])
    ctx.set_reference_cutting(0.25)
    ctx.set_builder(capi.BUILDER_AUTO)


@pytest.mark.parametrize("builder", [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH])
def test_odd_triangles_among_ordinary_ones(ctx, builder):
    """a point, a collinear triple, a triangle a million times the scene's size, one a millionth of a millionth of it, and needles
    through the whole scene, among 6000 ordinary triangles: the builders (which cut the long ones into pieces) and the kernel give
    the answers of the loop over all polygons"""
    rng = np.random.default_rng(12)
    c = rng.uniform(-1, 1, (6000, 1, 3))
    polys = list(c + rng.normal(size=(6000, 3, 3)) * 0.01)
    polys += [np.full((3, 3), 0.25),
              np.array([[0.0, 0, 0], [0.1, 0.1, 0.1], [0.2, 0.2, 0.2]]),
              np.array([[-1e6, -1e6, -0.5], [2e6, -1e6, -0.5], [-1e6, 2e6, -0.5]]),
              np.array([[0.5, 0.5, 0.5], [0.5 + 1e-12, 0.5, 0.5], [0.5, 0.5 + 1e-12, 0.5]]),
              np.array([[-50.0, -50, -50], [50, 50, 50], [50, 50.001, 50]]),
              np.array([[-1.0, 0.2, 0.3], [1.0, 0.2, 0.3], [1.0, 0.2000001, 0.3]])]
    with np.errstate(invalid="ignore"):
        mesh = polygon_mesh(polys)
    mesh["vn"] = np.nan_to_num(mesh["vn"])
    mesh["vn"][np.linalg.norm(mesh["vn"], axis=1) == 0] = [0.0, 0.0, 1.0]
    sc = oracle.Scene(mesh)
    rays = synth.incoherent_rays(40000, seed=51)
    aimed = synth.incoherent_rays(10000, seed=52)
    aimed[:, 4:7] = (rng.uniform(-1, 1, (10000, 1)) * np.array([[1.0, 1.0, 1.0]]) + rng.normal(size=(10000, 3)) * 1e-3 - aimed[:, 0:3]).astype(np.float32)  # at the long needle
    rays = np.concatenate([rays, aimed])
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    assert (o_poly >= 0).mean() > 0.3 and (o_poly >= 6000).sum() > 100
    ctx.set_builder(builder)
    ctx.upload_mesh(mesh)
    ctx.set_builder(capi.BUILDER_AUTO)
    assert ctx.reference_cutting()[0] >= 1  # the diagonal needle (the axis-aligned one and the flat giant waste nothing)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    assert np.array_equal(t64, o_t) and (gp != o_poly).sum() <= 2


def fan_triangle_vertices(mesh, tri):
    """[n, 3, 3] vertices of fan triangles given by their index in polygon order"""
    begin = mesh["face_begin"].astype(np.int64)
    counts = np.diff(begin) - 2
    tri_begin = np.concatenate([[0], np.cumsum(counts)])
    face = np.searchsorted(tri_begin, tri, side="right") - 1
    k = tri - tri_begin[face]
    v = mesh["v"]
    return np.stack([v[begin[face]], v[begin[face] + k + 1], v[begin[face] + k + 2]], axis=1)


TWO_STREAMS_WORKER = r"""
import os, sys
import numpy as np
import torch  # before the library: the wheel's HIP runtime has to be the one both use
sys.path.insert(0, sys.argv[1])
from ray_tracer_amd import capi, synth
ctx = capi.Context(0)
ctx.upload_mesh(synth.soup(200000, 0.01, seed=5))
n = 1 << 21
ra, rb = synth.incoherent_rays(n, seed=91), synth.incoherent_rays(n, seed=92)
da, db = torch.from_numpy(ra).cuda(), torch.from_numpy(rb).cuda()
want = []
os.environ["MI355RT_RAY_BINS"] = "0"
for d in (da, db):
    h = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.trace_closest_device(d.data_ptr(), n, h.data_ptr(), t.data_ptr())
    ctx.render_sync()
    want.append((h.cpu().numpy(), t.cpu().numpy()))
assert not np.array_equal(want[0][0], want[1][0])
assert (want[0][0][:, 1] != -1).sum() > n // 4
os.environ["MI355RT_RAY_BINS"] = "1"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(3):
    ha, hb = torch.zeros((n, 4), dtype=torch.int32, device="cuda"), torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    ta, tb = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.trace_closest_device(da.data_ptr(), n, ha.data_ptr(), ta.data_ptr(), stream=s1.cuda_stream)
    ctx.trace_closest_device(db.data_ptr(), n, hb.data_ptr(), tb.data_ptr(), stream=s2.cuda_stream)
    ctx.trace_closest_device(da.data_ptr(), n, ha.data_ptr(), ta.data_ptr(), stream=s1.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(ha.cpu().numpy(), want[0][0]) and np.array_equal(ta.cpu().numpy(), want[0][1]), rep
    assert np.array_equal(hb.cpu().numpy(), want[1][0]) and np.array_equal(tb.cpu().numpy(), want[1][1]), rep
ctx.close()
print("two streams ok")
"""


def test_ordered_batches_on_two_streams_do_not_share_buffers_at_once(tmp_path):
    """Two ordered batches of one context enqueued back to back on two different streams, without a host wait between them: the
    ordering goes through buffers that belong to the context, so the second batch must wait (on the device) until the first one's
    results are out.  Both must equal the unordered answers.  In a process of its own: the streams and device buffers are
    PyTorch's, and a process that uses PyTorch has to import it before the library is loaded."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "two_streams.py"
    worker.write_text(TWO_STREAMS_WORKER)
    out = subprocess.run([sys.executable, str(worker), root], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "two streams ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_host_buffer_batch_in_parts(ctx):
    """rt_trace_closest takes a batch beyond 2^22 rays in parts (copies of one part overlap the kernel of another): the same
    answers as the same rays handed over in batches small enough to go in one part, hits and fp64 t, last short part included"""
    mesh = synth.soup(100000, 0.01, seed=77)
    ctx.upload_mesh(mesh)
    n = (2 << 22) + 4099
    rays = synth.incoherent_rays(n, seed=4242)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    assert (hits["prim"] != capi.MISS).sum() > n // 4
    step = 3000000
    for at in range(0, n, step):
        h, t = ctx.trace_closest(rays[at:at + step], want_t64=True)
        assert np.array_equal(h, hits[at:at + step]) and np.array_equal(t, t64[at:at + step])
    h2 = ctx.trace_closest(rays)  # without t64
    assert np.array_equal(h2, hits)
