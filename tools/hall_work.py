import os, sys, tempfile, time
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer
d = tempfile.mkdtemp()
paths = synth.atrium(os.path.join(d, "hall"))
r = Renderer(0, seed=5); r.Load(paths["obj"])
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
nx, ny = 480, 270
r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
ctx = r.ctx
ctx.set_camera(r.cam)
ctx.render_begin(nx, ny, 2, 30, capi.ALG_MIS, 1234)
finish = ctx.capture_rays(1 << 19)
ctx.render_pass(0, 2); ctx.render_sync()
rays = finish().copy()
print("rays", len(rays), "stats", {k: ctx.stats()[k] for k in ("n_triangles", "bvh_nodes", "bvh_depth")})
sc = oracle.Scene(r.mesh)
wn, wt = ctx.wide_bvh()
print("wide nodes", len(wn))
sub = rays[:100000]
poly, tri, t, nv, nt = sc.trace_wide(wn, wt, sub)
print("wide: nodes/ray %.2f tris/ray %.2f hit rate %.3f" % (nv / len(sub), nt / len(sub), (poly >= 0).mean()))
tt = np.where(np.isfinite(t), t, 0); print("mean t", tt.mean(), "tmax finite share", np.isfinite(sub[:, 7]).mean())
# child occupancy of the wide nodes
imask = wn["imask"] if "imask" in wn.dtype.names else None
qlo = wn["qlo"]; qhi = wn["qhi"]
valid = (qlo[:, 0, :] <= qhi[:, 0, :])
print("children per node %.2f, interior share %.2f" % (valid.sum(1).mean(), (np.unpackbits(wn["imask"][:, None], axis=1).sum(1)).mean() / valid.sum(1).mean()))
