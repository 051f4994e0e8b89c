"""Would the rays of a render gain from being ordered by the cell of their origin (csrc/ray_bins.hip), as large unordered batches in
scenes beyond 2M triangles do?  Captures the rays of a small render of the hall (all bounces, connection rays included, in the order
the integrator traced them), repeats them to a batch of 2^23, and traces that batch as it is and ordered (MI355RT_RAY_BINS=1, the
ordering and the way back inside the time).  usage: python tools/render_rays_ordering_experiment.py"""
import os, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.renderer import Renderer

paths = synth.atrium(os.path.join(tempfile.mkdtemp(), "hall"))
r = Renderer(0, seed=5); r.Load(paths["obj"])
mats, omm = scene_io.load_materials(paths["materials"]), scene_io.load_objmatmap(paths["objmatmap"])
assert r.LoadMaterials({name: mats[omm[name]] for name in omm})
nx, ny = 960, 540
r.cam.film_height = 0.024; r.cam.film_width = 0.024 * nx / ny; r.cam.aperture = 0.05
r.cam.SetLens(60.0, 0.4); r.cam.set_Camera([0.0, 1.3, 7.5], [0.0, 1.1, 0.0], [0, 1, 0])
ctx = r.ctx
ctx.set_camera(r.cam)
ctx.render_begin(nx, ny, 4, 30, capi.ALG_MIS, 1234)
finish = ctx.capture_rays(1 << 23)
ctx.render_pass(0, 4); ctx.render_sync()
rays = finish().copy()
reps = -(-(1 << 23) // len(rays))
batch = np.concatenate([rays] * reps)[: 1 << 23]
d_rays = torch.from_numpy(batch).cuda()
m = len(batch)
d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
out = {}
for tag, env in (("as traced", "0"), ("ordered by origin cell", "1"), ("shuffled", "0"), ("shuffled, then ordered", "1")):
    if tag == "shuffled":
        d_rays = d_rays[torch.randperm(m, device="cuda")].contiguous()
    os.environ["MI355RT_RAY_BINS"] = env
    for _ in range(2):
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync(); torch.cuda.synchronize()
    ctx.profile_start()
    for _ in range(3):
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync()
    p = ctx.profile_stop()["closest"]
    out[tag] = 3 * m / (p["ms"] * 1e-3) / 1e9
print("hall, %d captured render rays repeated to %d: " % (len(rays), m) + ", ".join("%s %.2f Grays/s" % kv for kv in out.items()))
