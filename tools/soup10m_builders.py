import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ray_tracer_amd import capi, synth
ctx = capi.Context(0)
mesh = synth.soup(10_000_000, 0.002, seed=12345)
m = 1 << 24
d_rays = torch.from_numpy(synth.incoherent_rays(m, seed=777)).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
for name, b in (("device", capi.BUILDER_AUTO), ("host SAH", capi.BUILDER_HOST_SAH)):
    ctx.set_builder(b)
    t0 = time.time(); ctx.upload_mesh(mesh); up = time.time() - t0
    st = ctx.stats()
    for _ in range(2): ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync(); ctx.profile_start()
    for _ in range(3): ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync(); p = ctx.profile_stop()["closest"]
    print("10M soup, %s tree: upload %.1f s (build %.2f s, %d nodes, depth %d), %.3f Grays/s" % (name, up, st["bvh_build_seconds"], st["bvh_nodes"], st["bvh_depth"], 3 * m / (p["ms"] * 1e-3) / 1e9), flush=True)
