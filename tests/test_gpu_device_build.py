"""BVH built on the device (Morton radix tree collapsed to the 8-wide nodes): the topology differs from the host
builder's and from the reference's (bvh.cpp:134-232), the hits must not."""
import os

import numpy as np
import pytest

import oracle
from ray_tracer_amd import capi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def ctx():
    c = capi.Context(0)
    c.set_builder(capi.BUILDER_DEVICE_LBVH)
    yield c
    c.close()


def compare(ctx, mesh, rays, mode=oracle.MODE_LIST_LOWEST, n_check=None):
    sc = oracle.Scene(mesh)
    ctx.upload_mesh(mesh)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    sub = slice(0, n_check or len(rays))
    o_poly, o_tri, o_t = sc.trace_closest(rays[sub], mode)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    assert np.array_equal(t64[sub], o_t), "t differs from the oracle's"
    tie = gp[sub] != o_poly
    assert tie.sum() <= max(1, len(o_poly) // 50000), "ids differ beyond exact ties"
    short = rays[sub].copy()
    short[:, 7] = 0.3
    assert np.array_equal(ctx.trace_any(short), sc.trace_any(short, oracle.MODE_LIST))
    return hits, t64, (o_poly >= 0).mean()


def test_soup_against_brute_force(ctx):
    mesh = synth.soup(30000, 0.03, seed=4)
    hits, t64, rate = compare(ctx, mesh, synth.incoherent_rays(20000, seed=6))
    assert rate > 0.3
    st = ctx.stats()
    assert st["bvh_depth"] >= 3 and st["bvh_nodes"] > 1000 and st["bvh_build_seconds"] > 0


def test_cornell_against_brute_force(ctx):
    mesh, records, cam, oc = oracle.cornell_setup(os.path.join(HERE, "golden", "cornell"), 64, 64)
    rays = synth.incoherent_rays(30000, seed=8)
    rays[:, 0:3] *= 0.95
    compare(ctx, mesh, rays)


def test_same_hits_as_the_host_builder(ctx):
    """100k-triangle soup, 200k rays: ids and t identical to the tree of the host SAH builder"""
    mesh = synth.soup(100000, 0.01, seed=12)
    rays = synth.incoherent_rays(200000, seed=3)
    ctx.upload_mesh(mesh)
    h_dev, t_dev = ctx.trace_closest(rays, want_t64=True)
    ctx.set_builder(capi.BUILDER_HOST_SAH)
    ctx.upload_mesh(mesh)
    h_host, t_host = ctx.trace_closest(rays, want_t64=True)
    assert np.array_equal(t_dev, t_host)
    assert (h_dev["prim"] != h_host["prim"]).sum() <= 2 and (h_host["prim"] != capi.MISS).mean() > 0.3


def test_degenerate_inputs(ctx):
    """many triangles with the same centre (identical Morton keys), duplicates, two triangles, tiny and huge extents"""
    rng = np.random.default_rng(1)
    n = 3000
    c = np.zeros((n, 3))
    c[n // 2:] = rng.uniform(-1, 1, (n - n // 2, 3)) * np.array([1e3, 1e-3, 1.0])
    v = c[:, None, :] + rng.normal(size=(n, 3, 3)) * 0.05
    v[: n // 4] = v[0]  # exact duplicates
    idx = np.arange(3 * n, dtype=np.int64)
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 1])
    mesh = {"v": v.reshape(-1, 3), "vt": np.zeros((0, 3)), "vn": nrm, "corners": np.stack([idx, np.full(3 * n, -1), idx // 3], 1),
            "face_begin": np.arange(0, 3 * n + 1, 3, dtype=np.uint64), "object_begin": np.array([0, n], dtype=np.uint64), "names": ["x"]}
    rays = synth.incoherent_rays(6000, seed=2)
    rays[:, 0:3] *= 0.2
    compare(ctx, mesh, rays)
    two = {k: (val[:6] if k == "v" else val) for k, val in mesh.items()}
    two.update(vn=nrm[:2], corners=mesh["corners"][:6], face_begin=mesh["face_begin"][:3], object_begin=np.array([0, 2], dtype=np.uint64))
    compare(ctx, two, rays)


def test_render_matches_host_built_scene(ctx):
    """the wavefront integrator over a device-built tree: same film as over the host-built tree (fp32 atomics aside)"""
    from ray_tracer_amd import scenes
    mesh, records, cam = scenes.cornell(48, 32)
    films = []
    for b in (capi.BUILDER_DEVICE_LBVH, capi.BUILDER_HOST_SAH):
        ctx.set_builder(b)
        ctx.upload_mesh(mesh)
        ctx.upload_materials(records)
        ctx.set_camera(cam)
        for alg in (capi.ALG_NEE, capi.ALG_MIS):
            ctx.render_begin(48, 32, 8, 30, alg, 5)
            ctx.render_pass(0, 8)
            ctx.render_sync()
            films.append(ctx.download_film())
    assert films[0].sum() > 0
    assert np.allclose(films[0], films[2], rtol=1e-5, atol=1e-9) and np.allclose(films[1], films[3], rtol=1e-5, atol=1e-9)


def test_bvh2_is_reported_absent(ctx):
    ctx.upload_mesh(synth.soup(100, 0.1, seed=1))
    with pytest.raises(capi.BackendError):
        ctx.bvh2()
