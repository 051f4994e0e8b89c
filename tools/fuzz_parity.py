"""randomised closest-hit parity run: soups of random size and triangle size, both tree builders, both triangle tests, rays that
include axis-parallel ones; ids and fp64 t of the GPU against the oracle's walk of the GPU's own tree (exact boxes, same triangle
arithmetic) and, on a subset, against the brute-force list
usage: python tools/fuzz_parity.py [iterations = 40] [seed = 1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from ray_tracer_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0)
bad = 0
t0 = time.time()
for it in range(iters):
    n = int(2 + rng.integers(0, 3) ** 2 * rng.integers(1, 30000) + rng.integers(0, 200))
    sigma = float(10 ** rng.uniform(-3, -0.5))
    builder = [capi.BUILDER_HOST_SAH, capi.BUILDER_DEVICE_LBVH][int(rng.integers(0, 2))]
    watertight = bool(rng.integers(0, 2))
    mesh = synth.soup(n, sigma, seed=int(rng.integers(1, 1 << 30)))
    stretched = 0
    if n >= 4096 and rng.integers(0, 2):
        # some triangles drawn out into long thin ones (30 to 300 times along a random direction): the builders cut their references
        # into pieces (csrc/presplit.h), with a small or a very large budget
        stretched = int(max(1, n * 10 ** rng.uniform(-3.5, -2)))
        pick = rng.choice(n, stretched, replace=False)
        v = mesh["v"].reshape(n, 3, 3)
        c = v[pick].mean(1, keepdims=True)
        d = rng.normal(size=(stretched, 1, 3))
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        rel = v[pick] - c
        v[pick] = c + rel + (rel * d).sum(2, keepdims=True) * d * (10 ** rng.uniform(1.5, 2.5, (stretched, 1, 1)) - 1.0)
        nrm = np.cross(v[pick, 1] - v[pick, 0], v[pick, 2] - v[pick, 1])
        mesh["vn"][pick] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
        ctx.set_reference_cutting(float(rng.choice([0.25, 4.0])))
    ctx.set_builder(builder)
    ctx.set_triangle_test(capi.TRIANGLE_WATERTIGHT if watertight else capi.TRIANGLE_REFERENCE)
    ctx.upload_mesh(mesh)
    m = 40000
    rays = synth.incoherent_rays(m, seed=int(rng.integers(1, 1 << 30)))
    k = m // 8  # some rays along an axis, some with two zero components
    rays[:k, 4 + int(rng.integers(0, 3)) % 3] = 0.0
    rays[k:k + k // 4, 4:7] = 0.0
    rays[k:k + k // 4, 4 + int(rng.integers(0, 3))] = 1.0
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    sc = oracle.Scene(mesh)
    sc.set_triangle_test(watertight)
    nodes, tri_index = ctx.wide_bvh()
    poly, tri, t, nv, nt = sc.trace_wide(nodes, tri_index, rays)
    hit_o = poly >= 0
    hit_g = hits["prim"] != 0xFFFFFFFF
    same = np.array_equal(hit_o, hit_g) and np.array_equal(hits["prim"][hit_o], poly[hit_o].astype(np.uint32)) and np.array_equal(t64[hit_o], t[hit_o])
    sub = rays[:3000]
    l_poly, l_tri, l_t = sc.trace_closest(sub, oracle.MODE_LIST_LOWEST)
    lh = l_poly >= 0
    same_list = np.array_equal(lh, hit_g[:3000]) and np.array_equal(t64[:3000][lh], l_t[lh])
    if not (same and same_list):
        bad += 1
        print("MISMATCH it %d: n %d sigma %.4g builder %d watertight %d stretched %d (tree walk %s, list %s)" % (it, n, sigma, builder, watertight, stretched, same, same_list), flush=True)
    cut_total = globals().get("cut_total", 0) + ctx.reference_cutting()[1]
    if it % 5 == 4:
        print("%d iterations, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
ctx.close()
print("done: %d iterations, %d mismatches, %d references added by cutting in all" % (iters, bad, globals().get("cut_total", 0)))
sys.exit(1 if bad else 0)
