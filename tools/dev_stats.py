"""diagnostic: wave-level loop statistics of the trace kernel (library built with -DTRACE_STATS)
usage: dev_stats.py soup <n> <sigma> <rays>   |   dev_stats.py cornell"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth, scenes
ctx = capi.Context(0)
if sys.argv[1] == "cornell":
    mesh, records, cam = scenes.cornell(960, 540)
    ctx.upload_mesh(mesh); ctx.upload_materials(records); ctx.set_camera(cam)
    ctx.render_begin(960, 540, 4, 30, capi.ALG_MIS, 1234)
    finish = ctx.capture_rays(1 << 22)
    ctx.render_pass(0, 4); ctx.render_sync()
    rays = finish()
    m = len(rays)
else:
    n = int(sys.argv[2]); sigma = float(sys.argv[3]); m = int(sys.argv[4])
    ctx.upload_mesh(synth.soup(n, sigma))
    rays = synth.incoherent_rays(m)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
sink = torch.zeros(m, dtype=torch.float64, device='cuda')
ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr(), sink.data_ptr()); ctx.render_sync()
s = sink[:16].view(torch.int64).cpu().numpy()
print("rays %d" % m)
print("node iters %d lanes/iter %.1f | leaf iters %d lanes/iter %.1f | outer %d lanes %.1f" % (s[0], s[1] / max(s[0], 1), s[2], s[3] / max(s[2], 1), s[4], s[5] / max(s[4], 1)))
print("per ray: node lane-steps %.2f leaf lane-steps %.2f; per 64 rays: node iters %.1f leaf iters %.1f outer %.1f" % (s[1] / m, s[3] / m, s[0] * 64 / m, s[2] * 64 / m, s[4] * 64 / m))
if s[6]:
    print("inside tests: iters %d lanes/iter %.1f per ray %.2f; per 64 rays %.1f" % (s[6], s[7] / s[6], s[7] / m, s[6] * 64 / m))
clk = s[8:14].astype(float)
if clk.sum() > 0:
    names = ["refill", "node phase", "count+publish", "fetch+test", "merge+readback", "write results"]
    print("wave time by section: " + ", ".join("%s %.1f%%" % (n, 100 * c / clk.sum()) for n, c in zip(names, clk)))
