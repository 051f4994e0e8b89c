"""rt_trace_closest on host buffers (rays and results cross PCIe): Mrays/s of three calls in a row on the 1M soup (the first one also
allocates the staging buffers, the pinned pieces and the streams)
usage: python tools/time_host_buffer.py [m = 2^24]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_amd import capi, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
ctx = capi.Context(0)
ctx.set_builder(capi.BUILDER_HOST_SAH)
ctx.upload_mesh(synth.soup(1000000, 0.004, seed=12345))
rays = synth.incoherent_rays(m, seed=777)
for want in (False, False, False, True):
    t0 = time.perf_counter()
    ctx.trace_closest(rays, want_t64=want)
    dt = time.perf_counter() - t0
    print("t64 %d: %.1f ms, %.0f Mrays/s" % (want, dt * 1e3, m / dt / 1e6), flush=True)
ctx.close()
