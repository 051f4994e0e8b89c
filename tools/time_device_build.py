"""times the BVH build on the device for soups of growing size (second build of each: allocations warm), with a checksum of the closest hits of 2^20 rays
usage: python tools/time_device_build.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]
for n in sizes:
    sigma = 0.004 * (1e6 / n) ** (1.0 / 3.0)
    mesh = synth.soup(n, sigma, seed=12345)
    ctx = capi.Context(0)
    ctx.set_builder(capi.BUILDER_DEVICE_LBVH)
    secs = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.upload_mesh(mesh)
        secs.append((time.perf_counter() - t0, ctx.stats()["bvh_build_seconds"]))
    m = 1 << 20
    d_rays = torch.from_numpy(synth.incoherent_rays(m, seed=777)).cuda()
    d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
    ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync()
    h = d_hits.cpu().numpy()
    print("n %9d: upload %s s, build %s s, hits %d, checksum %d" % (n, " ".join("%.3f" % a for a, _ in secs), " ".join("%.4f" % b for _, b in secs),
                                                                    int((h[:, 1] != -1).sum()), int(h[:, 0].astype(np.int64).sum() % 1000000007)))
    ctx.close()
