"""How fast the CPU port (oracle/rt_oracle.c) is against the reference itself, so that the benchmark's CPU figure (the port on
the GPU box's cores) can be read as a reference-equivalent.  The reference's renderer was timed once, in a container of this class
(8 vCPU, g++ -O3, OpenMP 8 threads): Cornell box 512 x 512 x 64 spp, NEE+MIS, 13.3 s = 1.26 Msamples/s; its sources do not build in
this image (Boost is absent), so that figure is a constant here.  This script times the port on the same workload with the same
number of threads and prints the ratio; bench.py carries the result as cpu_baseline.port_over_reference.
usage: python tools/port_over_reference.py [threads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402

REFERENCE_MSAMPLES_PER_S = 1.26  # compiled reference, 8 threads, Cornell 512 x 512 x 64 spp MIS: 13.3 s
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx = ny = 512
mesh, records, cam, oc = oracle.cornell_setup(os.path.join(ROOT, "tests", "golden", "cornell"), nx, ny)
sc = oracle.Scene(mesh)
assert sc.set_materials(records) == 1
sc.build_bvh()
best = None
for _ in range(3):
    t0 = time.time()
    film, st = sc.render(oc, nx, ny, 64, 30, oracle.ALG_MIS, seed=1, nthreads=threads)
    dt = time.time() - t0
    best = dt if best is None else min(best, dt)
rate = st["samples"] / best / 1e6
print(json.dumps({"workload": "Cornell box 512 x 512 x 64 spp, NEE+MIS, 30 bands", "threads": threads, "port_seconds": best, "port_msamples_per_s": rate,
                  "reference_msamples_per_s": REFERENCE_MSAMPLES_PER_S, "reference_seconds": 13.3, "port_over_reference": rate / REFERENCE_MSAMPLES_PER_S,
                  "closest_rays_per_sample": st["closest_rays"] / st["samples"], "film_mean_per_pixel": float(film.sum() / (nx * ny))}))
