"""What ordering incoherent rays by the cell of their origin is worth to the closest-hit kernel: the soup of tools/soup_trace.py,
the same rays unordered and ordered by a Morton key of `bits` bits per axis (the ordering itself is done by torch here and not timed)
usage: python tools/sorted_rays_experiment.py <n> <sigma> [m = 2^24]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth

n, sigma = int(sys.argv[1]), float(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 24
ctx = capi.Context(0)
ctx.set_builder(capi.BUILDER_HOST_SAH if n < 2_000_000 else capi.BUILDER_AUTO)
ctx.upload_mesh(synth.soup(n, sigma, seed=12345))
rays = torch.from_numpy(synth.incoherent_rays(m, seed=777)).cuda()
hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")


def timed(r):
    for _ in range(2):
        ctx.trace_closest_device(r.data_ptr(), m, hits.data_ptr())
    ctx.render_sync()
    ctx.profile_start()
    for _ in range(3):
        ctx.trace_closest_device(r.data_ptr(), m, hits.data_ptr())
    ctx.render_sync()
    return ctx.profile_stop()["closest"]["ms"] / 3


def spread(v, bits):
    out = torch.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def ordered(key):
    """the rays in the order of the key; torch works on its own stream and the library on the context's, so the copy has to be complete
    before the library is asked to trace it"""
    r = rays[torch.argsort(key)].contiguous()
    torch.cuda.synchronize()
    return r


base = timed(rays)
print("soup %d: unordered %.3f ms, %.1f Mrays/s" % (n, base, m / base / 1e3), flush=True)
o = rays[:, 0:3]
lo, hi = o.min(0).values, o.max(0).values
for bits in (2, 4):
    q = ((o - lo) / (hi - lo) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    key = spread(q[:, 0], bits) | (spread(q[:, 1], bits) << 1) | (spread(q[:, 2], bits) << 2)
    t = timed(ordered(key))
    print("  %2d bits per axis (%d cells): %.3f ms, %.1f Mrays/s (%+.1f %%)" % (bits, 1 << (3 * bits), t, m / t / 1e3, (base / t - 1) * 100), flush=True)
# the same cells, and inside a cell the octant of the direction (the kernel visits a node's children in the order the octant gives)
d = rays[:, 4:7]
octant = (d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2)
for bits in (2, 3):
    q = ((o - lo) / (hi - lo) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    key = spread(q[:, 0], bits) | (spread(q[:, 1], bits) << 1) | (spread(q[:, 2], bits) << 2)
    for name, k2 in (("cell, then octant", key * 8 + octant), ("octant, then cell", octant * (1 << (3 * bits)) + key)):
        t = timed(ordered(k2))
        print("  %d bits per axis, %s: %.3f ms, %.1f Mrays/s (%+.1f %%)" % (bits, name, t, m / t / 1e3, (base / t - 1) * 100), flush=True)
ctx.close()
