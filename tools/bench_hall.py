"""Auxiliary measurement for the large-OBJ configuration: the procedural hall (about 320 k polygons, 31 objects, 12 light polygons, all
BSDF types; ray_tracer_amd/synth.py) at 3840x2160, NEE+MIS, in passes of one queue batch.  Prints one JSON line.
usage: python tools/bench_hall.py [--spp 32] [--algorithm mis]"""
import argparse, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer, MIS, NEE, Naive, NaiveBDPTType

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--algorithm", default="mis")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--long-polygons", action="store_true", help="the hall with strips and braces (synth.atrium(long_polygons=True))")
args = ap.parse_args()
d = tempfile.mkdtemp()
t0 = time.time(); paths = synth.atrium(os.path.join(d, "hall"), long_polygons=args.long_polygons); t_synth = time.time() - t0
r = Renderer(0, seed=5)
t0 = time.time(); r.Load(paths["obj"]); t_load = time.time() - t0
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
t0 = time.time(); assert r.LoadMaterials({name: mats[omm[name]] for name in omm}); t_mat = time.time() - t0
st = r.ctx.stats()
nx, ny = args.width, args.height
r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
ctx = r.ctx
alg = {"mis": capi.ALG_MIS, "nee": capi.ALG_NEE, "naive": capi.ALG_NAIVE, "bdpt": capi.ALG_NAIVE_BDPT}[args.algorithm]
per = max(1, (1 << 28) // (nx * ny))  # one batch of the queues where 200 GB of HBM are free
ctx.set_camera(r.cam)
ctx.render_begin(nx, ny, args.spp + per, 30, alg, 7)
ctx.render_pass(0, per); ctx.render_sync()
ctx.profile_start(); t0 = time.perf_counter()
s = per
while s < args.spp + per:
    b = min(per, args.spp + per - s); ctx.render_pass(s, b); s += b
ctx.render_sync(); dt = time.perf_counter() - t0
prof = ctx.profile_stop(); st2 = ctx.stats()
k = prof["closest"]
print(json.dumps({"scene": "hall with long polygons" if args.long_polygons else "hall", "polygons": paths["polygons"], "triangles": st["n_triangles"], "objects": paths["objects"], "lights": st["n_lights"], "cut_triangles": ctx.reference_cutting()[0], "extra_references": ctx.reference_cutting()[1],
                  "obj_load_s": t_load, "upload_and_materials_s": t_mat, "bvh_build_s": st["bvh_build_seconds"], "bvh_depth": st["bvh_depth"],
                  "width": nx, "height": ny, "algorithm": args.algorithm, "spp": args.spp, "spp_per_s": args.spp / dt,
                  "msamples_per_s": args.spp * nx * ny / dt / 1e6, "rays_per_sample": (st2["closest_rays"] + st2["any_rays"]) / max(st2["camera_samples"], 1),
                  "kernels_ms": {n: v["ms"] for n, v in prof.items()}, "closest_grays_per_s_in_kernel": k["items"] / max(k["ms"], 1e-9) / 1e6}))
