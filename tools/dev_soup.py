"""exploratory GPU timing (not part of the test suite)"""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
ctx = capi.Context(0)
t0 = time.time(); mesh = synth.soup(n, sigma); print('soup gen', time.time() - t0, flush=True)
t0 = time.time(); ctx.upload_mesh(mesh); print('upload+build', time.time() - t0, ctx.stats(), flush=True)
rays = synth.incoherent_rays(m)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr()); ctx.render_sync()
    dt = time.time() - t0
    print('closest: %.3f ms  %.1f Mrays/s' % (dt * 1e3, m / dt / 1e6), flush=True)
h = d_hits.cpu().numpy()
print('hit rate', (h[:, 1] != -1).mean())
