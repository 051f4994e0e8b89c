"""A/B harness for the traversal kernels: closest-hit rate on the Cornell box's own render rays (cache-resident scene) and on the
1M-triangle soup (memory-bound scene), with a checksum of the results so that variants can be compared for identical answers.
usage: [MI355RT_LIB=path/to/libmi355rt.so] python tools/bench_trace.py [tag] [--soup10m] [--watertight]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, scenes, synth

tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.environ.get("MI355RT_LIB", "default")
ctx = capi.Context(0)
if "--watertight" in sys.argv:
    ctx.set_triangle_test(capi.TRIANGLE_WATERTIGHT)


def rate(rays, reps=5, any_hit=False):
    m = len(rays)
    d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
    d_t64 = torch.empty(m, dtype=torch.float64, device="cuda")
    d_any = torch.empty(m, dtype=torch.uint8, device="cuda")
    if any_hit:
        run = lambda: ctx.trace_any_device(d_rays.data_ptr(), m, d_any.data_ptr())
    else:
        run = lambda: ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr(), d_t64.data_ptr())
    for _ in range(2):
        run()
    ctx.render_sync()
    ctx.profile_start()
    for _ in range(reps):
        run()
    ctx.render_sync()
    p = ctx.profile_stop()["any" if any_hit else "closest"]
    # checksum that does not depend on the order of the rays (the capture's order varies from run to run): a 64-bit mix of each
    # ray with its result, summed with wrap-around
    with np.errstate(over="ignore"):
        key = np.ascontiguousarray(rays).view(np.uint64).reshape(m, 4)
        acc = key[:, 0] * np.uint64(0x9E3779B97F4A7C15) ^ key[:, 1] * np.uint64(0xBF58476D1CE4E5B9) ^ key[:, 2] * np.uint64(0x94D049BB133111EB) ^ key[:, 3]
        if any_hit:
            acc = acc * np.uint64(0xD1B54A32D192ED03) + d_any.cpu().numpy().astype(np.uint64)
        else:
            res = d_hits.cpu().numpy().view(np.uint64).reshape(m, 2)
            acc = (acc * np.uint64(0xD1B54A32D192ED03) + res[:, 0]) * np.uint64(0xFF51AFD7ED558CCD) + res[:, 1]
            acc = acc * np.uint64(0xC4CEB9FE1A85EC53) + d_t64.cpu().numpy().view(np.uint64)
        total = int(acc.sum(dtype=np.uint64))
    return m * reps / (p["ms"] * 1e-3) / 1e9, "%012x" % (total & 0xFFFFFFFFFFFF)


mesh, records, cam = scenes.cornell(960, 540)
ctx.upload_mesh(mesh)
ctx.upload_materials(records)
ctx.set_camera(cam)
os.environ["MI355RT_TRACE_ALL_CONNECTIONS"] = "1"  # the capture below must not depend on what the integrator drops
ctx.render_begin(960, 540, 4, 30, capi.ALG_MIS, 1234)
finish = ctx.capture_rays(1 << 23)
ctx.render_pass(0, 4)
ctx.render_sync()
rays = finish().copy()
rays = np.concatenate([rays, rays[::-1], rays])[: 1 << 24]
g, h = rate(rays)
ga, ha = rate(rays, any_hit=True)
out = "%-28s cornell %d rays: closest %.2f Grays/s [%s] any %.2f [%s]" % (tag, len(rays), g, h, ga, ha)
if "MI355RT_BUILDER" not in os.environ:
    ctx.set_builder(capi.BUILDER_HOST_SAH)  # as bench.py's 1M leg
ctx.upload_mesh(synth.soup(1_000_000, 0.004, seed=12345))
ctx.set_builder(capi.BUILDER_AUTO)
g, h = rate(synth.incoherent_rays(1 << 24, seed=777))
out += " | soup1m closest %.3f [%s]" % (g, h)
if "--soup10m" in sys.argv:
    ctx.upload_mesh(synth.soup(10_000_000, 0.002, seed=12345))
    g, h = rate(synth.incoherent_rays(1 << 24, seed=777), reps=3)
    out += " | soup10m closest %.3f [%s]" % (g, h)
print(out, flush=True)
ctx.close()
