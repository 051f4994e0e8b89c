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

# ---- experiment: the same rays pre-sorted (direction octant, then Morton code of the origin)
def morton3(q):
    q = q.astype(np.uint64)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
if os.environ.get("SORT_EXPERIMENT"):
    bits = int(os.environ.get("SORT_BITS", "9"))
    q = np.clip(((rays[:, 0:3] + 1.0) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    octant = (rays[:, 4] < 0).astype(np.uint64) | ((rays[:, 5] < 0).astype(np.uint64) << 1) | ((rays[:, 6] < 0).astype(np.uint64) << 2)
    for name, key in (("octant+morton", (octant << np.uint64(3 * bits)) | morton3(q)), ("morton only", morton3(q)), ("morton+octant", (morton3(q) << np.uint64(3)) | octant)):
        perm = np.argsort(key, kind="stable")
        d_sorted = torch.from_numpy(rays[perm]).cuda()
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            ctx.trace_closest_device(d_sorted.data_ptr(), m, d_hits.data_ptr()); ctx.render_sync()
            dt = time.time() - t0
        print('sorted (%s): %.3f ms  %.1f Mrays/s' % (name, dt * 1e3, m / dt / 1e6), flush=True)
