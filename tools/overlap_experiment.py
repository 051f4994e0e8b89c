"""Does the GPU have room next to one context's render?  The bench workload (Cornell box 1920 x 1080, NEE+MIS) on one context and on
groups of two and three contexts that share the one GPU (each member renders its share of every pass on its own stream: where one
member's kernels leave the chip partly empty -- the tail of a traversal launch, a latency-bound shading kernel -- another member's run).
usage: python tools/overlap_experiment.py [hall]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_amd import capi, scenes

nx, ny, spp, steps = 1920, 1080, 64, 8
mesh, records, cam = scenes.cornell(nx, ny)
for members in (1, 2, 3, 4):
    c = capi.Context(0) if members == 1 else capi.Group([0] * members)
    c.upload_mesh(mesh); c.upload_materials(records); c.set_camera(cam)
    per = spp * members  # every member gets a full batch of 64 spp per pass
    c.render_begin(nx, ny, per * (steps + 1), 30, capi.ALG_MIS, 7)
    c.render_pass(0, per); c.render_sync()
    t0 = time.perf_counter()
    for s in range(1, steps + 1):
        c.render_pass(s * per, per)
    c.render_sync()
    dt = time.perf_counter() - t0
    print("%d context(s) on one GPU, %d spp per pass each: %.0f spp/s" % (members, spp, per * steps / dt), flush=True)
    c.close()
