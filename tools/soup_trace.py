"""The incoherent closest-hit workload of bench.py's aux section on its own, for profiling under rocprofv3:
soup(n, sigma, seed 12345), m incoherent rays (seed 777), 2 warm-up + `reps` launches of the closest-hit kernel.
usage: python tools/soup_trace.py <n> <sigma> [m = 2^24] [reps = 5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth

n, sigma = int(sys.argv[1]), float(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = capi.Context(0)
ctx.set_builder(capi.BUILDER_HOST_SAH if n < 2_000_000 else capi.BUILDER_AUTO)  # as bench.py's two soup legs
ctx.upload_mesh(synth.soup(n, sigma, seed=12345))
d_rays = torch.from_numpy(synth.incoherent_rays(m, seed=777)).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
for _ in range(2 + reps):
    ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
ctx.render_sync()
ctx.profile_start()
ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
ctx.render_sync()
p = ctx.profile_stop()["closest"]
print("soup %d sigma %g: %d rays, %.3f ms, %.1f Mrays/s, hit rate %.3f" % (n, sigma, m, p["ms"], m / p["ms"] / 1e3, float((d_hits[:, 1] != -1).float().mean().item())))
ctx.close()
