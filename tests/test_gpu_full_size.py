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
    # ids and t against the oracle: its walk of this tree on 64 k rays, and brute force over all ten million triangles on 300 of
    # them (nothing shared with the product: a builder that lost a triangle at this size would show here)
    assert 0.4 < _check_against_oracle(ctx, mesh, rays[: 1 << 16], brute=300) < 0.9
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


def _check_against_oracle(ctx, mesh, rays, brute=300):
    """ids and fp64 t of the GPU on `rays` against the oracle's exact polygon test: walked through the product's own 8-wide tree for
    all rays (topology shared, arithmetic independent; whichever builder made it), and by brute force over every polygon for a
    sample (nothing shared)"""
    sc = oracle.Scene(mesh)
    nodes, tri_index = ctx.wide_bvh()
    poly, tri, t, _, _ = sc.trace_wide(nodes, tri_index, rays, nthreads=16)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    assert np.array_equal(t64, t)
    assert (gp != poly).sum() <= 2  # exact ties in t between neighbouring polygons may resolve differently
    sub = np.random.default_rng(5).choice(len(rays), brute, replace=False)
    bp, _, bt = sc.trace_closest(rays[sub], oracle.MODE_LIST_LOWEST, nthreads=16)
    assert np.array_equal(t64[sub], bt) and (gp[sub] != bp).sum() <= 1
    return float((gp >= 0).mean())


def _film_with_and_without_culling(ctx, nx, ny, ns, seed):
    ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, seed)
    ctx.render_pass(0, ns)
    ctx.render_sync()
    film, st = ctx.download_film(), ctx.stats()
    os.environ["MI355RT_TRACE_ALL_CONNECTIONS"] = "1"
    try:
        ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, seed)
        ctx.render_pass(0, ns)
        ctx.render_sync()
        film_all, st_all = ctx.download_film(), ctx.stats()
    finally:
        del os.environ["MI355RT_TRACE_ALL_CONNECTIONS"]
    assert np.isfinite(film).all() and film.min() >= 0.0 and film.sum() > 0.0
    assert st["camera_samples"] == st_all["camera_samples"] == nx * ny * ns
    assert st_all["closest_rays"] > st["closest_rays"]
    assert np.allclose(film, film_all, rtol=1e-4, atol=1e-8)
    return film, st


def test_scan_sized_ply_object_full_hd(tmp_path):
    """The Cornell box with an 870 k-triangle PLY object (a procedural stand-in for a scanned model: none is available offline),
    written to disk and read back through the PLY reader, once of glass and once of gold, 1920x1080 NEE+MIS"""
    from ray_tracer_amd import scene_io
    v, faces = synth.blob(660, 660, (0.0, -0.45, 0.0), 0.4)
    path = str(tmp_path / "blob.ply")
    scene_io.write_ply(path, v, faces)
    p = scene_io.Ply()
    assert p.Load(path) and len(p.face_begin) - 1 == len(faces) == 869880
    nx, ny = 1920, 1080
    mesh, records, cam = scenes.cornell(nx, ny)
    mesh = p.to_mesh("blob", "auto", into=mesh)
    one = np.ones(30)
    lam = 405.0 + 10.0 * np.arange(30)
    glass = {"type": capi.MAT_DIELECTRIC, "light_flag": 0, "a": 1.5 * one, "b": 0.95 * one}
    gold = {"type": capi.MAT_CONDUCTOR, "light_flag": 0, "a": np.interp(lam, [400, 500, 550, 700], [1.45, 0.95, 0.35, 0.13]),
            "b": np.interp(lam, [400, 500, 550, 700], [1.95, 1.85, 2.60, 3.90])}
    ctx = capi.Context(0)
    ctx.set_builder(capi.BUILDER_HOST_SAH)
    ctx.upload_mesh(mesh)
    st = ctx.stats()
    assert st["n_triangles"] == 869880 + 972 and st["n_objects"] == 8
    ctx.set_camera(cam)
    means = {}
    for name, rec in (("glass", glass), ("gold", gold)):
        ctx.upload_materials(list(records) + [rec])
        film, _ = _film_with_and_without_culling(ctx, nx, ny, 4, 77)
        means[name] = film.sum() / (nx * ny)
    assert 0.03 < means["glass"] < 0.12 and 0.03 < means["gold"] < 0.12
    # the rays of a render of this scene (all bounces, connection rays included), at a tenth of the resolution
    _, _, cam_s = scenes.cornell(nx // 8, ny // 8)
    ctx.set_camera(cam_s)
    ctx.render_begin(nx // 8, ny // 8, 4, 30, capi.ALG_MIS, 1234)
    finish = ctx.capture_rays(1 << 19)
    ctx.render_pass(0, 4)
    ctx.render_sync()
    rays = finish().copy()
    assert len(rays) >= 100_000
    assert _check_against_oracle(ctx, mesh, rays) > 0.5
    ctx.close()


def test_large_obj_project_4k(tmp_path):
    """A multi-object OBJ project of modelled-scene size (about 320 k polygons: quads, 58 k of them not planar, and triangles; 31 objects, 12 light polygons, all five
    BSDF node types; procedural, written to disk) loaded through Renderer.Load + LoadMaterials and rendered at 3840x2160 with 30 bands"""
    from ray_tracer_amd import scene_io
    from ray_tracer_amd.renderer import Renderer, MIS
    paths = synth.atrium(str(tmp_path / "hall"))
    assert paths["polygons"] >= 250_000 and paths["objects"] >= 20
    r = Renderer(0, seed=5)
    r.Load(paths["obj"])
    mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
    assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
    st = r.ctx.stats()
    assert st["n_faces"] == paths["polygons"] and st["n_objects"] == paths["objects"] and st["n_lights"] == 12
    assert st["n_triangles"] > st["n_faces"]  # quads became two fan triangles each
    nx, ny = 3840, 2160
    r.cam.film_height = 0.024
    r.cam.film_width = 0.024 * nx / ny
    r.cam.aperture = 0.05
    r.cam.SetLens(60.0, 0.4)
    r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
    r.algorithm_type = MIS
    r.preview_img_flag = False
    r.samples_per_pass = 2
    r.RenderImage(nx, ny, 2, 30)
    film = r.spectrum_img
    assert film.shape == (nx * ny, 30) and np.isfinite(film).all() and film.min() >= 0.0
    lit = film.sum(1) > 0
    assert lit.mean() > 0.8  # the camera stands inside the hall: nearly every pixel sees a lit surface
    assert r.preview_img.shape == (nx * ny, 3) and 0.0 <= r.preview_img.min() and r.preview_img.max() <= 1.0
    # the same film with nothing dropped, at a size where two renders are cheap; then ids and t on this scene's render rays
    ctx = r.ctx
    film_s, _ = _film_with_and_without_culling(ctx, 480, 270, 4, 11)
    ratio = (film_s.sum() / (480 * 270)) / (film.sum() / (nx * ny))
    assert 0.8 < ratio < 1.25  # same scene and camera at two resolutions: the same mean radiance, up to noise
    ctx.render_begin(480, 270, 2, 30, capi.ALG_MIS, 1234)
    finish = ctx.capture_rays(1 << 19)
    ctx.render_pass(0, 2)
    ctx.render_sync()
    rays = finish().copy()
    assert len(rays) >= 100_000
    assert _check_against_oracle(ctx, r.mesh, rays) > 0.9
    # the tree built on the device answers the same
    hits_host, t_host = ctx.trace_closest(rays, want_t64=True)
    ctx.set_builder(capi.BUILDER_DEVICE_LBVH)
    ctx.upload_mesh(r.mesh)
    hits_dev, t_dev = ctx.trace_closest(rays, want_t64=True)
    assert np.array_equal(t_host, t_dev) and (hits_host["prim"] != hits_dev["prim"]).sum() <= 2
    ctx.close()


def test_hall_with_long_polygons_renders_the_same_whole_or_cut(tmp_path):
    """The hall as a modelled building has it -- strip walls and 96 thin braces crossing it diagonally (synth.atrium(long_polygons=True))
    -- rendered with the references of its long triangles whole and cut into pieces (csrc/presplit.h), under both builders: the
    same ray counts and the same film (the hits are the same, so the paths are; the film differs by the order of its fp32 additions)."""
    from ray_tracer_amd import scene_io
    from ray_tracer_amd.renderer import Renderer
    paths = synth.atrium(str(tmp_path / "hall"), long_polygons=True)
    mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
    nx, ny, ns = 480, 270, 4
    films, rays = [], []
    for builder in (capi.BUILDER_DEVICE_LBVH, capi.BUILDER_HOST_SAH):
        for budget in (0.0, 0.25):
            r = Renderer(0, seed=5)
            r.ctx.set_builder(builder)
            r.ctx.set_reference_cutting(budget)
            r.Load(paths["obj"])
            assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
            n_cut, n_extra = r.ctx.reference_cutting()
            assert (n_cut, n_extra) == (0, 0) if budget == 0.0 else (n_cut >= 150 and n_extra >= 20 * n_cut)
            r.cam.film_height = 0.024
            r.cam.film_width = 0.024 * nx / ny
            r.cam.aperture = 0.05
            r.cam.SetLens(60.0, 0.4)
            r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
            ctx = r.ctx
            ctx.set_camera(r.cam)
            ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, 77)
            ctx.render_pass(0, ns)
            ctx.render_sync()
            films.append(ctx.download_film())
            st = ctx.stats()
            rays.append((st["closest_rays"], st["any_rays"], st["camera_samples"]))
            ctx.close()
    assert films[0].sum() > 0 and np.isfinite(films[0]).all()
    for f, n in zip(films[1:], rays[1:]):
        assert n == rays[0]
        assert np.allclose(f, films[0], rtol=1e-4, atol=1e-8)
