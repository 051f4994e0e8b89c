"""RT_TRIANGLE_WATERTIGHT: the optional watertight triangle test (no counterpart in the reference; parity is against the oracle's
restatement of the same published algorithm, operation for operation).  What is checked: ids and fp64 t bit-exact against the oracle,
no leaks through shared edges and vertices where the reference's own test leaks, independence of the tree for polygons that are not
planar, and films against the oracle in this mode."""
import numpy as np
import pytest

import oracle
from ray_tracer_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = capi.Context(0)
    c.set_triangle_test(capi.TRIANGLE_WATERTIGHT)
    yield c
    c.close()


def mesh_of(v, faces, normals=None):
    """one object, one polygon per row of `faces` (all rows the same length), a flat normal per polygon unless given per vertex"""
    v = np.asarray(v, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    k = faces.shape[1]
    if normals is None:
        n = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 1]])
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        vn_idx = np.repeat(np.arange(len(faces)), k)
    else:
        n = np.asarray(normals, dtype=np.float64)
        vn_idx = faces.reshape(-1)
    corners = np.stack([faces.reshape(-1), np.full(faces.size, -1), vn_idx], 1)
    return {"v": v, "vt": np.zeros((0, 3)), "vn": n, "corners": corners, "face_begin": np.arange(0, faces.size + 1, k, dtype=np.uint64),
            "object_begin": np.array([0, len(faces)], dtype=np.uint64), "names": ["mesh"]}


def closed_sphere(n_lat, n_lon, radius=1.0):
    th = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    ring = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    v = radius * np.concatenate([[[0, 1.0, 0]], ring, [[0, -1.0, 0]]])
    idx = 1 + np.arange((n_lat - 1) * n_lon).reshape(n_lat - 1, n_lon)
    nxt = np.roll(idx, -1, axis=1)
    tris = [np.stack([np.zeros(n_lon, int), idx[0], nxt[0]], 1)]
    a, b, a2, b2 = idx[:-1], idx[1:], nxt[:-1], nxt[1:]
    tris += [np.stack([a, b, b2], -1).reshape(-1, 3), np.stack([a, b2, a2], -1).reshape(-1, 3)]
    tris.append(np.stack([np.full(n_lon, len(v) - 1), nxt[-1], idx[-1]], 1))
    return v, np.concatenate(tris)


def gpu_vs_oracle(ctx, mesh, rays, brute=2000):
    sc = oracle.Scene(mesh)
    sc.set_triangle_test(True)
    ctx.upload_mesh(mesh)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    nodes, tri_index = ctx.wide_bvh()
    poly, tri, t, _, _ = sc.trace_wide(nodes, tri_index, rays)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    assert np.array_equal(t64, t)
    same = gp == poly
    assert same.mean() > 0.999 and np.array_equal(t64[~same], t[~same])  # exact ties between neighbours may name either
    sub = rays[:brute]
    bp, _, bt = sc.trace_closest(sub, oracle.MODE_LIST_LOWEST)  # every polygon, no tree at all
    assert np.array_equal(t64[:brute], bt)
    got_any = ctx.trace_any(np.concatenate([rays[:, :7], np.full((len(rays), 1), 0.75, np.float32)], 1))
    assert np.array_equal(got_any.astype(bool), t64 <= 0.75)
    return hits, t64


def test_ids_and_t_match_the_oracle(ctx, cornell_dir):
    rays = synth.incoherent_rays(100000, seed=777)
    rays[:, 0:3] *= 0.95
    for obj_name in ("cornell_box_triangulated.obj", "cornell_box.obj"):
        mesh = oracle.cornell_setup(cornell_dir, 8, 8, obj_name=obj_name)[0]
        hits, t64 = gpu_vs_oracle(ctx, mesh, rays)
        assert (hits["prim"] != capi.MISS).mean() > 0.7  # the box is open towards the camera
    gpu_vs_oracle(ctx, synth.soup(100000, 0.01, seed=12345), synth.incoherent_rays(100000, seed=777), brute=300)


def test_shared_edges_and_vertices_do_not_leak(ctx):
    # a unit square of two triangles and a fan of four around a centre vertex: rays aimed at the diagonal and at the shared vertex
    v = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 0], [4, 0, 0], [4, 2, 0], [2, 2, 0], [3, 1, 0]]
    faces = [[0, 1, 2], [0, 2, 3], [4, 5, 8], [5, 6, 8], [6, 7, 8], [7, 4, 8]]
    mesh = mesh_of(v, faces)
    k = np.linspace(0.125, 0.875, 7)
    o = np.concatenate([np.stack([k, k, np.ones(7)], 1), [[3, 1, 1]], np.stack([2 + 2 * k, np.full(7, 1.0) * 2 * k, np.ones(7)], 1)])
    rays = np.zeros((len(o), 8), np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 6], rays[:, 7] = o, 1e-3, -1.0, np.inf
    ctx.upload_mesh(mesh)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    assert (hits["prim"] != capi.MISS).all() and np.array_equal(t64, np.ones(len(o)))
    ctx.set_triangle_test(capi.TRIANGLE_REFERENCE)  # the reference's strict test: the cross products vanish on the edge, the ray goes through
    leaks = ctx.trace_closest(rays)["prim"] == capi.MISS
    assert leaks[:8].all()
    ctx.set_triangle_test(capi.TRIANGLE_WATERTIGHT)
    # a closed sphere seen from inside, rays aimed at its vertices and edge midpoints: nothing may escape
    sv, tris = closed_sphere(48, 96)
    ctx.upload_mesh(mesh_of(sv, tris, normals=sv))
    rng = np.random.default_rng(2)
    targets = np.concatenate([sv, 0.5 * (sv[tris[:, 0]] + sv[tris[:, 1]]), 0.5 * (sv[tris[:, 1]] + sv[tris[:, 2]])])
    org = rng.uniform(-0.3, 0.3, (len(targets), 3))
    rays = np.zeros((len(targets), 8), np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = org, 1e-3, targets - org, np.inf
    assert (ctx.trace_closest(rays)["prim"] != capi.MISS).all()
    assert ctx.trace_any(rays).all()


def test_hits_on_warped_quads_do_not_depend_on_the_tree(ctx):
    """quads that are not planar: the reference's plane distance puts hits outside the polygon's own box (tree-dependent there);
    the watertight test works on each fan triangle's own plane, and both builders' trees answer identically"""
    g = 60
    u, w = np.meshgrid(np.linspace(-1, 1, g), np.linspace(-1, 1, g), indexing="xy")
    v = np.stack([u, 0.15 * np.sin(5 * u) * np.cos(4 * w) + 0.05 * np.sin(31 * u * w), w], -1).reshape(-1, 3)
    i = np.arange(g - 1)[:, None] * g + np.arange(g - 1)[None, :]
    quads = np.stack([i, i + 1, i + g + 1, i + g], -1).reshape(-1, 4)
    mesh = mesh_of(v, quads)
    rays = synth.incoherent_rays(200000, seed=5)
    rays[:, 1] = np.abs(rays[:, 1]) * 0.5 + 0.3
    results = []
    for builder in (capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH):
        ctx.set_builder(builder)
        hits, t64 = gpu_vs_oracle(ctx, mesh, rays, brute=3000)
        results.append((hits["prim"].copy(), hits["tri"].copy(), t64.copy()))
    assert np.array_equal(results[0][2], results[1][2]) and (results[0][1] != results[1][1]).mean() < 1e-4
    assert (results[0][0] != capi.MISS).mean() > 0.15


@pytest.mark.parametrize("alg", [capi.ALG_NEE, capi.ALG_MIS, capi.ALG_NAIVE_BDPT])
def test_film_matches_the_oracle_in_this_mode(ctx, cornell_dir, alg):
    nx, ny, ns = 64, 48, 8
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, nx, ny)
    sc = oracle.Scene(mesh)
    sc.set_triangle_test(True)
    assert sc.set_materials(records) == 1
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    ctx.render_begin(nx, ny, ns, 30, alg, 42)
    ctx.render_pass(0, ns)
    g = ctx.download_film()
    o, _ = sc.render(oc, nx, ny, ns, 30, alg, 42)
    assert abs(g.sum() - o.sum()) <= 1e-3 * o.sum()
    pg, po = g.sum(1), o.sum(1)
    assert (np.abs(pg - po) <= 1e-3 * np.abs(po) + 1e-7).mean() >= 0.99
    assert np.sqrt(np.mean((g - o) ** 2)) / np.sqrt(np.mean(o ** 2)) <= 0.05
