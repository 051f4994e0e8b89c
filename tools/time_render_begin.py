import sys, time, os
sys.path.insert(0, os.getcwd())
from ray_tracer_amd import capi, scenes
ctx = capi.Context(0)
for (nx, ny) in ((1920, 1080), (3840, 2160)):
    mesh, records, cam = scenes.cornell(nx, ny)
    t0 = time.perf_counter(); ctx.upload_mesh(mesh); t1 = time.perf_counter(); ctx.upload_materials(records); ctx.set_camera(cam); t2 = time.perf_counter()
    ctx.render_begin(nx, ny, 64, 30, capi.ALG_MIS, 1); t3 = time.perf_counter()
    ctx.render_pass(0, 8); ctx.render_sync(); t4 = time.perf_counter()
    ctx.render_begin(nx, ny, 64, 30, capi.ALG_MIS, 1); t5 = time.perf_counter()
    ctx.render_pass(0, 8); ctx.render_sync(); t6 = time.perf_counter()
    print("%dx%d upload_mesh %.1f ms, materials+camera %.1f ms, render_begin %.1f ms (again %.1f ms), first pass of 8 spp %.1f ms (again %.1f ms)" % (nx, ny, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t5-t4)*1e3, (t4-t3)*1e3, (t6-t5)*1e3))
    t0 = time.perf_counter(); f = ctx.download_film(); t1 = time.perf_counter(); p = ctx.download_preview(8); t2 = time.perf_counter()
    print("%dx%d download_film %.1f ms (%.0f MB of doubles), download_preview %.1f ms" % (nx, ny, (t1-t0)*1e3, f.nbytes/1e6, (t2-t1)*1e3))
