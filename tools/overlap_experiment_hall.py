"""tools/overlap_experiment.py for the hall at 3840 x 2160: one context against two contexts that share the GPU (a group of [0, 0])"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer
paths = synth.atrium(os.path.join(tempfile.mkdtemp(), "hall"))
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
nx, ny = 3840, 2160
r = Renderer(0, seed=5)
r.Load(paths["obj"])
records = [mats[omm[name]].record() for name in r.mesh["names"]]
r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
mesh, cam = r.mesh, r.cam
r.ctx.close()
for members in (1, 2):
    c = capi.Context(0) if members == 1 else capi.Group([0] * members)
    c.upload_mesh(mesh); c.upload_materials(records); c.set_camera(cam)
    per, steps = 32, 3
    c.render_begin(nx, ny, per * (steps + 1), 30, capi.ALG_MIS, 7)
    c.render_pass(0, per); c.render_sync()
    t0 = time.perf_counter()
    for s in range(1, steps + 1):
        c.render_pass(s * per, per)
    c.render_sync()
    dt = time.perf_counter() - t0
    print("%d context(s) on one GPU, %d spp per pass in all: %.1f spp/s" % (members, per, per * steps / dt), flush=True)
    c.close()
