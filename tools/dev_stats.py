"""diagnostic: wave-level loop statistics of the trace kernel (library built with -DTRACE_STATS)"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth
n = int(sys.argv[1]); sigma = float(sys.argv[2]); m = int(sys.argv[3])
ctx = capi.Context(0)
ctx.upload_mesh(synth.soup(n, sigma))
rays = synth.incoherent_rays(m)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
sink = torch.zeros(m, dtype=torch.float64, device='cuda')
ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr(), sink.data_ptr()); ctx.render_sync()
s = sink[:6].view(torch.int64).cpu().numpy()
print("node iters %d lanes/iter %.1f | leaf iters %d lanes/iter %.1f | outer %d lanes %.1f" % (s[0], s[1] / max(s[0], 1), s[2], s[3] / max(s[2], 1), s[4], s[5] / max(s[4], 1)))
print("per ray: node lane-steps %.1f leaf lane-steps %.2f" % (s[1] / m, s[3] / m))
