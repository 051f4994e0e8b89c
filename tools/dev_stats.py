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
elif sys.argv[1] == "hall":  # the procedural hall's render rays of all bounces, with the oracle's ideal walk of the same tree beside them
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle
    from ray_tracer_amd import scene_io
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
    ctx.render_begin(nx, ny, 2, 30, capi.ALG_MIS, 1234)
    finish = ctx.capture_rays(1 << 22)
    ctx.render_pass(0, 2); ctx.render_sync()
    rays = finish().copy()
    m = len(rays)
    wn, wt = ctx.wide_bvh()
    sub = rays[np.random.default_rng(1).choice(m, 100000, replace=False)]
    _, _, _, nv, nt = oracle.Scene(r.mesh).trace_wide(wn, wt, sub)
    print("ideal ordered walk with distances: nodes/ray %.2f tris/ray %.2f" % (nv / len(sub), nt / len(sub)))
else:
    n = int(sys.argv[2]); sigma = float(sys.argv[3]); m = int(sys.argv[4])
    ctx.upload_mesh(synth.soup(n, sigma))
    rays = synth.incoherent_rays(m)
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device='cuda')
sink = torch.zeros(m, dtype=torch.float64, device='cuda')
sink[:20].view(torch.int64)[14] = (1 << 62)  # minimum of the waves' start times
torch.cuda.synchronize()
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
w = sink[:20].view(torch.int64).cpu().numpy()
if w[18] > 0:
    first, last_start, sum_end, last_end, n_w = int(w[14]), int(w[15]), int(w[16]), int(w[17]), int(w[18])
    span = last_end - first
    print("waves %d: from the first wave's start to the last wave's end %.1f us; the last wave starts %.1f us in; the mean wave ends %.1f %% of that span before the last one (idle time)" %
          (n_w, span / 100.0, (last_start - first) / 100.0, 100.0 * (1.0 - (sum_end / n_w - first) / max(span, 1))))
h = sink[:128].view(torch.int64).cpu().numpy()
print("rays by node steps (bucket = highest power of two not above the count; first bucket: none): " + " ".join("%d:%d" % (1 << (k - 1) if k else 0, h[32 + k]) for k in range(0, 24) if h[32 + k]))
print("rays by passes of the wave's main loop they were resident for: " + " ".join("%d:%d" % (1 << (k - 1) if k else 0, h[72 + k]) for k in range(0, 24) if h[72 + k]))

