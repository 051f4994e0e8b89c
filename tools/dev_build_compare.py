"""exploratory: build time and closest-hit rate, host SAH builder against the device builder"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth
n = int(sys.argv[1]); sigma = float(sys.argv[2]); m = int(sys.argv[3])
ctx = capi.Context(0)
mesh = synth.soup(n, sigma)
rays = synth.incoherent_rays(m)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
ref = None
for name, b in (("device LBVH", capi.BUILDER_DEVICE_LBVH), ("device LBVH again", capi.BUILDER_DEVICE_LBVH), ("host SAH", capi.BUILDER_HOST_SAH)):
    ctx.set_builder(b)
    t0 = time.time(); ctx.upload_mesh(mesh); up = time.time() - t0
    st = ctx.stats()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr()); ctx.render_sync()
        dt = time.time() - t0
    h = d_hits.cpu().numpy()
    same = "" if ref is None else " prim mismatches vs first: %d" % int((h[:, 1] != ref[:, 1]).sum())
    ref = h if ref is None else ref
    print("%-18s upload %.3f s (build %.3f s) nodes %d depth %d | %.1f Mrays/s%s" % (name, up, st["bvh_build_seconds"], st["bvh_nodes"], st["bvh_depth"], m / dt / 1e6, same), flush=True)
