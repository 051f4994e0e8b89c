"""Work per ray in the tree the kernels walk: the oracle's ordered walk (exact boxes, entry distances on its stack) of the exported
8-wide tree over a sample of the scene's own render rays, for the hall with or without long polygons, either builder, with the
references of long triangles cut or not.  Prints one line per configuration.
usage: python tools/tree_work.py [--long-polygons] [--rays 100000]"""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer

ap = argparse.ArgumentParser()
ap.add_argument("--long-polygons", action="store_true")
ap.add_argument("--rays", type=int, default=100000)
args = ap.parse_args()
paths = synth.atrium(os.path.join(tempfile.mkdtemp(), "hall"), long_polygons=args.long_polygons)
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
nx, ny = 960, 540
sub = None
for builder, bname in ((capi.BUILDER_DEVICE_LBVH, "device"), (capi.BUILDER_HOST_SAH, "host")):
    for budget in (0.0, 0.25):
        r = Renderer(0, seed=5)
        r.ctx.set_builder(builder); r.ctx.set_reference_cutting(budget)
        r.Load(paths["obj"])
        assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
        r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
        r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
        ctx = r.ctx
        if sub is None:  # the same rays for every tree
            ctx.set_camera(r.cam)
            ctx.render_begin(nx, ny, 2, 30, capi.ALG_MIS, 1234)
            finish = ctx.capture_rays(1 << 22)
            ctx.render_pass(0, 2); ctx.render_sync()
            rays = finish().copy()
            sub = rays[np.random.default_rng(1).choice(len(rays), args.rays, replace=False)]
            sc = oracle.Scene(r.mesh)
        wn, wt = ctx.wide_bvh()
        poly, tri, t, nv, nt = sc.trace_wide(wn, wt, sub)
        hits, t64 = ctx.trace_closest(sub, want_t64=True)
        same = np.array_equal(t64, t) and np.array_equal(hits["prim"].astype(np.int64)[poly >= 0], poly[poly >= 0])
        cut = ctx.reference_cutting()
        print("%s builder, budget %.2f: %d triangles cut, %d references added, %d nodes, depth %d | nodes/ray %.2f triangle tests/ray %.2f | kernel = oracle walk: %s"
              % (bname, budget, cut[0], cut[1], len(wn), ctx.stats()["bvh_depth"], nv / len(sub), nt / len(sub), same), flush=True)
        r.ctx.close()
