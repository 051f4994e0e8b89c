"""exploratory: closest-hit rate on the Cornell box + PLY blob scene, render rays in queue order and shuffled"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ray_tracer_amd import capi, synth, scenes, scene_io
from bench_ply_scene import blob
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 660
ctx = capi.Context(0)
v, faces = blob(grid, grid, (0.0, -0.45, 0.0), 0.4)
scene_io.write_ply("/tmp/b.ply", v, faces)
p = scene_io.Ply(); p.Load("/tmp/b.ply")
nx, ny = 960, 540
mesh, records, cam = scenes.cornell(nx, ny)
mesh = p.to_mesh("blob", "auto", into=mesh)
one = np.ones(30)
records = list(records) + [{"type": capi.MAT_DIELECTRIC, "light_flag": 0, "a": 1.5 * one, "b": 0.95 * one}]
ctx.upload_mesh(mesh); print(ctx.stats(), flush=True)
ctx.upload_materials(records); ctx.set_camera(cam)
ctx.render_begin(nx, ny, 4, 30, capi.ALG_MIS, 1234)
finish = ctx.capture_rays(1 << 22)
ctx.render_pass(0, 4); ctx.render_sync()
rays = finish()
print("captured", len(rays), flush=True)
def run(name, r):
    m = len(r)
    d_rays = torch.from_numpy(np.ascontiguousarray(r)).cuda()
    d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr()); ctx.render_sync()
        dt = time.time() - t0
    print('%s: %d rays %.3f ms  %.1f Mrays/s' % (name, m, dt * 1e3, m / dt / 1e6), flush=True)
    return d_hits.cpu().numpy()
h = run("queue order", rays)
perm = np.random.default_rng(1).permutation(len(rays))
run("shuffled", rays[perm])
hit_blob = h[:, 3] == 7
print("fraction hitting blob", hit_blob.mean())
run("only blob hitters", rays[hit_blob])
run("only others", rays[~hit_blob])
run("first 2M (primary)", rays[:nx*ny*4][:2000000])
