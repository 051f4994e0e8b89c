"""The references of long triangles are cut into pieces before the tree is built (csrc/presplit.h).  Every piece carries the box of
its part of the triangle; together the boxes must contain every point the reference's polygon test can accept for the triangle
(object.cpp:244-253), or a ray could slip through the tree past a triangle a loop over all polygons (hittablelist.cpp:3-17) hits.
Host arithmetic only."""
import numpy as np

from ray_tracer_amd import capi
from test_accept_region import accepted, plane_points, region_box, spike_points


def mesh_of(polygons):
    """polygons: list of [k, 3] vertex arrays -> one object of them"""
    v = np.concatenate(polygons)
    starts = np.cumsum([0] + [len(p) for p in polygons])
    with np.errstate(invalid="ignore", over="ignore"):
        vn = np.stack([np.cross(p[1] - p[0], p[2] - p[1]) for p in polygons])
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    face_of = np.repeat(np.arange(len(polygons)), [len(p) for p in polygons])
    corners = np.stack([np.arange(len(v)), np.full(len(v), -1), face_of], -1).astype(np.int64)
    return {"v": v, "vt": np.zeros((0, 3)), "vn": vn, "corners": corners,
            "face_begin": starts.astype(np.uint64), "object_begin": np.array([0, len(polygons)], dtype=np.uint64)}


def fan_triangles(polygons):
    """(a, b, c, face normal) per fan triangle, in the library's order"""
    out = []
    for p in polygons:
        nf = np.cross(p[1] - p[0], p[2] - p[1])
        nf = nf / np.linalg.norm(nf)
        for k in range(len(p) - 2):
            out.append((p[0], p[k + 1], p[k + 2], nf))
    return out


def pieces_of(first, extra, n_tris):
    """per fan triangle the list of its pieces' (lo, hi), empty for a triangle that was not cut"""
    pieces = [[] for _ in range(n_tris)]
    for r in list(first) + list(extra):
        pieces[int(r["tri"])].append((r["lo"].astype(np.float64), r["hi"].astype(np.float64)))
    return pieces


def small_clutter(rng, n, extent):
    """n small triangles spread over the scene: what the budget is measured against"""
    c = rng.uniform(-extent, extent, (n, 1, 3))
    return list(c + rng.normal(size=(n, 3, 3)) * extent * 2e-3)


def covered(pts, boxes, pad):
    inside = np.zeros(len(pts), dtype=bool)
    for lo, hi in boxes:
        inside |= np.all((pts >= lo - pad) & (pts <= hi + pad), axis=1)
    return inside


def test_long_triangles_are_cut_and_their_pieces_cover_them():
    rng = np.random.default_rng(5)
    long_ones = []
    for _ in range(40):  # thin diagonal triangles, 1 to 20 units long in a scene of 20
        a = rng.uniform(-10, 10, 3)
        d = rng.normal(size=3)
        d *= rng.uniform(1, 20) / np.linalg.norm(d)
        w = np.cross(d, rng.normal(size=3))
        w *= rng.uniform(0.005, 0.05) / np.linalg.norm(w)
        long_ones.append(np.stack([a, a + d, a + d + w]))
    polys = long_ones + small_clutter(rng, 6000, 10.0)
    first, extra = capi.cut_references(mesh_of(polys), budget=0.25, min_tris=1000)
    assert len(extra) <= 0.25 * len(polys)
    pieces = pieces_of(first, extra, len(polys))
    n_cut = sum(1 for t in range(40) if len(pieces[t]) >= 2)
    assert n_cut >= 35 and len(first) == sum(1 for p in pieces if p)
    assert sum(len(p) for p in pieces[40:]) < 0.02 * len(polys)  # the clutter is left alone
    for t in range(40):
        if not pieces[t]:
            continue
        a, b, c = polys[t]
        lo0, hi0 = region_box(a, b, c, fan_triangles([polys[t]])[0][3])
        lo0, hi0 = np.float32(lo0).astype(np.float64) - 1e-5, np.float32(hi0).astype(np.float64) + 1e-5
        for lo, hi in pieces[t]:
            assert np.all(lo >= lo0) and np.all(hi <= hi0) and np.all(lo <= hi)  # inside the whole triangle's box
        # the boxes are tight: their volume is a small part of the whole box's
        vol = sum(np.prod(np.maximum(hi - lo, 1e-3)) for lo, hi in pieces[t])
        assert vol < 0.6 * np.prod(np.maximum(hi0 - lo0, 1e-3)) or len(pieces[t]) < 3
        # every point of the triangle lies in a piece
        u = rng.dirichlet([0.6, 0.6, 0.6], 4000)
        pts = u @ np.stack([a, b, c])
        assert covered(pts, pieces[t], 1e-9 * np.abs(pts).max()).all()
        # so do points on its edges and the vertices themselves
        s = rng.uniform(0, 1, (1000, 1))
        edge = np.concatenate([a + s * (b - a), b + s * (c - b), c + s * (a - c), np.stack([a, b, c])])
        assert covered(edge, pieces[t], 1e-9 * np.abs(pts).max()).all()


def test_pieces_of_warped_quads_cover_what_the_reference_test_accepts():
    rng = np.random.default_rng(6)
    quads = []
    for k in range(60):  # long quads whose fourth vertex leaves the plane of the first three, by 1e-9 .. 1e-1 of the width
        a = rng.uniform(-10, 10, 3)
        d = rng.normal(size=3)
        d *= rng.uniform(4, 20) / np.linalg.norm(d)
        w = np.cross(d, rng.normal(size=3))
        w *= rng.uniform(0.05, 0.5) / np.linalg.norm(w)
        n = np.cross(d, w)
        n /= np.linalg.norm(n)
        warp = 10.0 ** rng.uniform(-9, -1) * np.linalg.norm(w)
        skew = rng.uniform(-0.3, 0.3) * d  # some corners obtuse
        quads.append(np.stack([a, a + d, a + d + w + skew + warp * n, a + w + skew]))
    polys = quads + small_clutter(rng, 6000, 10.0)
    first, extra = capi.cut_references(mesh_of(polys), budget=0.25, min_tris=1000)
    tris = fan_triangles(polys)
    pieces = pieces_of(first, extra, len(tris))
    checked = outside_triangle = 0
    for t in range(2 * len(quads)):
        if len(pieces[t]) < 2:
            continue
        a, b, c, nf = tris[t]
        size = np.linalg.norm(b - a) + np.linalg.norm(c - a)
        # points of the polygon's plane above the triangle and a margin around it, and along the spikes beyond its corners
        lab = rng.uniform(-0.05, 1.05, (6000, 2))
        x = lab[:, :1] * a + lab[:, 1:] * b + (1.0 - lab.sum(1, keepdims=True)) * c
        nrm = np.cross(b - a, c - b)
        nrm /= np.linalg.norm(nrm)
        near = x + (-((x - a) @ nf) / (nrm @ nf))[:, None] * nrm
        pts = np.concatenate([near, plane_points(a, nf, rng.uniform(-1.2, 1.2, (1000, 2)), size), spike_points(a, b, c, nf, rng, 600)])
        ok = accepted(a, b, c, pts)
        pad = 1e-9 * (np.abs(pts).max() + size)
        inside = covered(pts, pieces[t], pad)
        assert not (ok & ~inside).any(), (t, pts[ok & ~inside][:3])
        checked += int(ok.sum())
        lo, hi = np.minimum(a, np.minimum(b, c)), np.maximum(a, np.maximum(b, c))
        outside_triangle += int((ok & ~np.all((pts >= lo - pad) & (pts <= hi + pad), axis=1)).sum())
    assert checked > 10000 and outside_triangle > 0  # accepted points beyond the vertices' box were among the samples


def test_small_and_uniform_scenes_are_left_alone():
    rng = np.random.default_rng(7)
    first, extra = capi.cut_references(mesh_of(small_clutter(rng, 3000, 1.0)), budget=0.25)  # below 4096 triangles
    assert len(first) == 0 and len(extra) == 0
    first, extra = capi.cut_references(mesh_of(small_clutter(rng, 20000, 1.0)), budget=0.25)  # nothing stands out
    assert len(extra) < 200
    first, extra = capi.cut_references(mesh_of(small_clutter(rng, 20000, 1.0)), budget=0.0)
    assert len(first) == 0 and len(extra) == 0


def test_pathological_triangles_do_not_upset_the_cutting():
    """degenerate, enormous, tiny and not-finite triangles among ordinary ones: the call returns, pieces are finite boxes inside the
    scene's extent, and triangles that are not finite are left whole"""
    rng = np.random.default_rng(8)
    polys = small_clutter(rng, 5000, 1.0)
    odd = [np.zeros((3, 3)),                                                      # a point
           np.array([[0.0, 0, 0], [1, 1, 1], [2, 2, 2]]),                         # collinear
           np.array([[0.0, 0, 0], [1e30, 0, 0], [0, 1e30, 0]]),                   # enormous
           np.array([[0.0, 0, 0], [1e-30, 0, 0], [0, 1e-30, 0]]),                 # tiny
           np.array([[np.nan, 0, 0], [1.0, 0, 0], [0, 1.0, 0]]),                  # not a number
           np.array([[np.inf, 0, 0], [1.0, 0, 0], [0, 1.0, 0]]),                  # infinite
           np.array([[-50.0, -50, -50], [50, 50, 50], [50, 50.001, 50]]),         # a needle through the whole scene
           np.array([[-1.0, 0.2, 0.3], [1.0, 0.2, 0.3], [1.0, 0.2000001, 0.3]])]  # a needle along an axis
    mesh = mesh_of(odd + polys)
    mesh["vn"] = np.nan_to_num(mesh["vn"], nan=0.0, posinf=0.0, neginf=0.0)
    mesh["vn"][np.linalg.norm(mesh["vn"], axis=1) == 0] = [0.0, 0.0, 1.0]
    for budget in (0.25, 4.0):
        first, extra = capi.cut_references(mesh, budget=budget, min_tris=1000)
        both = np.concatenate([first, extra])
        assert np.isfinite(both["lo"]).all() and np.isfinite(both["hi"]).all() and np.all(both["lo"] <= both["hi"])
        assert not np.isin(both["tri"], [0, 1, 4, 5]).any()  # nothing to cut, or not finite
        assert len(extra) <= budget * len(mesh["face_begin"])
