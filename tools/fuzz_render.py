"""randomised film parity run: the Cornell box (triangles or quads) with random integrator, seed, image size, sample count and number
of spectral bands; GPU film against the oracle's render of the same samples (same random streams), with the tolerances of
tests/test_gpu_render.py: energy within 1e-3, at least 99 % of the pixels within 1e-3 relative, relative RMSE at most 0.05
usage: python tools/fuzz_render.py [iterations = 60] [seed = 1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cornell = os.path.join(ROOT, "tests", "golden", "cornell")
ctx = capi.Context(0)
bad = 0
t0 = time.time()
for it in range(iters):
    alg = [capi.ALG_NAIVE, capi.ALG_NEE, capi.ALG_MIS, capi.ALG_NAIVE_BDPT][int(rng.integers(0, 4))]
    nx, ny = int(rng.integers(8, 72)), int(rng.integers(8, 56))
    ns = int(rng.integers(1, 9))
    S = [30, 30, 10, 7, 1][int(rng.integers(0, 5))]
    seed = int(rng.integers(1, 1 << 40))
    obj_name = ["cornell_box_triangulated.obj", "cornell_box.obj"][int(rng.integers(0, 2))]
    mesh, records, cam, oc = oracle.cornell_setup(cornell, nx, ny, obj_name=obj_name)
    sc = oracle.Scene(mesh)
    assert sc.set_materials(records) == 1
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    ctx.render_begin(nx, ny, ns, S, alg, seed)
    first = 0
    while first < ns:  # random pass sizes: passes must add up
        b = int(rng.integers(1, ns - first + 1))
        ctx.render_pass(first, b)
        first += b
    ctx.render_sync()
    g = ctx.download_film()
    o, st = sc.render(oc, nx, ny, ns, S, alg, seed)
    tot_g, tot_o = g.sum(), o.sum()
    pg, po = g.sum(1), o.sum(1)
    close = (np.abs(pg - po) <= 1e-3 * np.abs(po) + 1e-7).mean()
    rmse = np.sqrt(np.mean((g - o) ** 2)) / max(np.sqrt(np.mean(o ** 2)), 1e-30)
    ok = np.isfinite(g).all() and abs(tot_g - tot_o) <= 1e-3 * tot_o and close >= 0.99 and rmse <= 0.05
    if not ok:
        bad += 1
        print("MISMATCH it %d: alg %d %dx%d ns %d S %d seed %d %s: energy %.3e vs %.3e, pixels %.4f, rmse %.3g" % (it, alg, nx, ny, ns, S, seed, obj_name, tot_g, tot_o, close, rmse), flush=True)
    if it % 10 == 9:
        print("%d iterations, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
ctx.close()
print("done: %d iterations, %d mismatches" % (iters, bad))
sys.exit(1 if bad else 0)
