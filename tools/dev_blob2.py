"""exploratory: find the slow rays of the Cornell + PLY blob render"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ray_tracer_amd import capi, synth, scenes, scene_io
from bench_ply_scene import blob
ctx = capi.Context(0)
v, faces = blob(660, 660, (0.0, -0.45, 0.0), 0.4)
scene_io.write_ply("/tmp/b.ply", v, faces)
p = scene_io.Ply(); p.Load("/tmp/b.ply")
nx, ny = 1920, 1080
mesh, records, cam = scenes.cornell(nx, ny)
mesh = p.to_mesh("blob", "auto", into=mesh)
one = np.ones(30)
records = list(records) + [{"type": capi.MAT_DIELECTRIC, "light_flag": 0, "a": 1.5 * one, "b": 0.95 * one}]
ctx.upload_mesh(mesh)
ctx.upload_materials(records); ctx.set_camera(cam)
ctx.render_begin(nx, ny, 1, 30, capi.ALG_MIS, 20261111)
finish = ctx.capture_rays(1 << 24)
ctx.render_pass(0, 1); ctx.render_sync()
rays = finish()
print("captured", len(rays), flush=True)
d_all = torch.from_numpy(rays).cuda()
d_hits = torch.empty((len(rays), 4), dtype=torch.int32, device='cuda')
def timed(lo, hi):
    torch.cuda.synchronize(); t0 = time.time()
    ctx.trace_closest_device(d_all.data_ptr() + 32 * lo, hi - lo, d_hits.data_ptr()); ctx.render_sync()
    return time.time() - t0
chunk = 1 << 18
times = [(timed(lo, min(lo + chunk, len(rays))), lo) for lo in range(0, len(rays), chunk)]
for t, lo in times: print("chunk at %9d: %8.3f ms" % (lo, t * 1e3))
t, lo = max(times)
hi = min(lo + chunk, len(rays))
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = timed(lo, mid), timed(mid, hi)
    if a > b: hi = mid
    else: lo = mid
    print(lo, hi, a * 1e3, b * 1e3, flush=True)
np.set_printoptions(precision=9, linewidth=200)
print("slowest ray", lo, rays[lo], rays[lo].view(np.uint32))
print("time of that ray alone %.3f ms" % (timed(lo, lo + 1) * 1e3))
