"""one film parity case in detail: GPU film against the oracle's sample by sample, the pixels that differ and by how much
usage: python tools/film_case.py ALG NX NY NS S SEED OBJ_NAME"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi

alg, nx, ny, ns, S, seed = (int(x) for x in sys.argv[1:7])
obj_name = sys.argv[7]
cornell = os.path.join(ROOT, "tests", "golden", "cornell")
mesh, records, cam, oc = oracle.cornell_setup(cornell, nx, ny, obj_name=obj_name)
sc = oracle.Scene(mesh)
assert sc.set_materials(records) == 1
ctx = capi.Context(0)
ctx.upload_mesh(mesh)
ctx.upload_materials(records)
ctx.set_camera(cam)
for k in range(ns):
    ctx.render_begin(nx, ny, ns, S, alg, seed)
    ctx.render_pass(k, 1)
    ctx.render_sync()
    g = ctx.download_film()
    o, st = sc.render(oc, nx, ny, ns, S, alg, seed, first_sample=k, n_samples=1)
    g, o = g.reshape(o.shape), o
    pg, po = g.sum(1), o.sum(1)
    d = np.abs(pg - po)
    bad = np.argwhere(d > 1e-3 * np.abs(po) + 1e-7)[:, 0]
    print("sample %d: energy %.6e vs %.6e, %d of %d pixels differ by more than 1e-3 relative" % (k, pg.sum(), po.sum(), len(bad), len(pg)))
    for i in bad[np.argsort(-d[bad])][:20]:
        print("    pixel %d (column %d, row %d): gpu %.9e oracle %.9e" % (i, i // ny, i % ny, pg[i], po[i]))
ctx.close()
