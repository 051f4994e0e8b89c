"""diagnostic: tree_work.py's figures for classes of the hall's render rays (camera rays, rays by distance of their origin from the scene's median planes)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer
paths = synth.atrium(os.path.join(tempfile.mkdtemp(), "hall"))
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
nx, ny = 960, 540
sub = None
for budget in (0.0, 0.25):
    r = Renderer(0, seed=5)
    r.ctx.set_builder(capi.BUILDER_DEVICE_LBVH); r.ctx.set_reference_cutting(budget)
    r.Load(paths["obj"])
    assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
    r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
    r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
    ctx = r.ctx
    if sub is None:
        ctx.set_camera(r.cam)
        ctx.render_begin(nx, ny, 2, 30, capi.ALG_MIS, 1234)
        finish = ctx.capture_rays(1 << 22)
        ctx.render_pass(0, 2); ctx.render_sync()
        rays = finish().copy()
        sub = rays[np.random.default_rng(1).choice(len(rays), 100000, replace=False)]
        sc = oracle.Scene(r.mesh)
    wn, wt = ctx.wide_bvh()
    cam = sub[:, 2] > 7.0
    classes = {"camera": cam, "others": ~cam, "origin within 0.1 of x=0": ~cam & (np.abs(sub[:, 0]) < 0.1), "of y=1.25": ~cam & (np.abs(sub[:, 1] - 1.25) < 0.1),
               "of z=0": ~cam & (np.abs(sub[:, 2]) < 0.1), "origin on the floor": ~cam & (sub[:, 1] < 0.05), "on the ceiling": ~cam & (sub[:, 1] > 2.45)}
    for name, m in classes.items():
        rs = sub[m]
        if len(rs) == 0:
            continue
        _, _, _, nv, nt = sc.trace_wide(wn, wt, rs)
        print("budget %.2f %-26s %6d rays: nodes/ray %.2f tests/ray %.2f" % (budget, name, len(rs), nv / len(rs), nt / len(rs)), flush=True)
    r.ctx.close()
