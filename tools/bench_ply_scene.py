"""Auxiliary measurement for the PLY row: the Cornell box with a scanned-model-sized PLY object (about 870 k triangles,
procedural: no model files are available offline) of glass or metal inside it, 1920x1080, NEE+MIS.
Prints one JSON line: load times (both parsers), BVH build, spp/s, closest-hit rate and roofline figures.
usage: python tools/bench_ply_scene.py [--material glass|gold] [--spp 64] [--pass-spp 64] [--grid 660]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--material", default="glass")
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--grid", type=int, default=660)
    ap.add_argument("--pass-spp", type=int, default=128, help="samples per pixel per render pass (one batch of the wavefront queues at 1080p)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    from ray_tracer_amd import capi, scene_io, scenes
    import bench

    path = "/tmp/blob_%d.ply" % args.grid
    from ray_tracer_amd import synth
    v, faces = synth.blob(args.grid, args.grid, (0.0, -0.45, 0.0), 0.4)
    scene_io.write_ply(path, v, faces)
    out = {"ply_file_mb": os.path.getsize(path) / 1e6, "ply_triangles": int(len(faces))}
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli"), "--ply", path, "--dump-ply", "/tmp/blob.bin"], capture_output=True)
    out["parse_s_cpp_host"] = time.time() - t0
    assert r.returncode == 0
    t0 = time.time()
    p = scene_io.Ply()
    assert p.Load(path)
    out["parse_s_python_host"] = time.time() - t0

    nx, ny = args.width, args.height
    mesh, records, cam = scenes.cornell(nx, ny)
    mesh = p.to_mesh("blob", "auto", into=mesh)
    one = np.ones(30)
    if args.material == "glass":
        rec = {"type": capi.MAT_DIELECTRIC, "light_flag": 0, "a": 1.5 * one, "b": 0.95 * one}
    else:  # a gold-like conductor: n falls and k rises with wavelength
        lam = 405.0 + 10.0 * np.arange(30)
        rec = {"type": capi.MAT_CONDUCTOR, "light_flag": 0, "a": np.interp(lam, [400, 500, 550, 700], [1.45, 0.95, 0.35, 0.13]),
               "b": np.interp(lam, [400, 500, 550, 700], [1.95, 1.85, 2.60, 3.90])}
    records = list(records) + [rec]
    ctx = capi.Context(0)
    t0 = time.time()
    ctx.upload_mesh(mesh)
    st = ctx.stats()
    out.update(upload_s=time.time() - t0, bvh_build_s=st["bvh_build_seconds"], triangles=st["n_triangles"], bvh_depth=st["bvh_depth"],
               device_mb=st["device_bytes"] / 1e6)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    spp, step = args.spp, args.pass_spp
    ctx.render_begin(nx, ny, spp + step, 30, capi.ALG_MIS, 20261111)
    ctx.render_pass(0, step)  # warm-up
    ctx.render_sync()
    ctx.profile_start()
    t0 = time.perf_counter()
    for s in range(step, spp + step, step):
        ctx.render_pass(s, min(step, spp + step - s))
    ctx.render_sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_stop()
    film = ctx.download_film()
    k = prof["closest"]
    out.update(material=args.material, spp=spp, spp_per_s=spp / dt, msamples_per_s=spp * nx * ny / dt / 1e6,
               kernels_ms={n: v_["ms"] for n, v_ in prof.items()}, closest_rays=k["items"],
               closest_grays_per_s_in_kernel=k["items"] / (k["ms"] * 1e-3) / 1e9, film_mean=float(film.sum() / (nx * ny)),
               film_finite=bool(np.isfinite(film).all()))
    if not args.no_oracle:
        import oracle
        mesh_s, records_s, cam_s = scenes.cornell(nx // 10, ny // 10)
        ctx.set_camera(cam_s)
        ctx.render_begin(nx // 10, ny // 10, 4, 30, capi.ALG_MIS, 1234)
        finish = ctx.capture_rays(1 << 19)
        ctx.render_pass(0, 4)
        ctx.render_sync()
        rays = finish()
        sc = oracle.Scene(mesh)
        nodes, tri_index = ctx.wide_bvh()
        poly, tri, t, nv, nt = sc.trace_wide(nodes, tri_index, rays, nthreads=16)
        hits, t64 = ctx.trace_closest(rays, want_t64=True)
        gp = hits["prim"].astype(np.int64)
        gp[hits["prim"] == capi.MISS] = -1
        bpr = bench.b_ray_wide(nv / len(rays), nt / len(rays), 128.0)
        out.update(nodes_per_ray=nv / len(rays), tris_per_ray=nt / len(rays), b_ray=bpr, parity_rays=int(len(rays)),
                   parity_prim_mismatch=int((gp != poly).sum()), parity_t_mismatch=int((t64 != t).sum()),
                   roofline_frac=k["items"] * bpr / (k["ms"] * 1e-3) / bench.HBM_PEAK)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
