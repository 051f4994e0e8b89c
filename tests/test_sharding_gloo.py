"""The N>1 path on CPU: two gloo ranks render disjoint sample blocks (with the CPU oracle standing in for
the device), all-reduce their films, and must reproduce the single-process image."""
import os
import subprocess
import sys

import numpy as np

from ray_tracer_amd.parallel import sample_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle
from ray_tracer_amd.parallel import sample_range, reduce_film
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
mesh, records, cam, oc = oracle.cornell_setup(os.path.join(sys.argv[1], "tests", "golden", "cornell"), 16, 12)
sc = oracle.Scene(mesh); sc.set_materials(records)
ns = 5
first, count = sample_range(rank, world, ns)
film, st = sc.render(oc, 16, 12, ns, 30, oracle.ALG_MIS, seed=11, first_sample=first, n_samples=count, nthreads=1)
t = torch.from_numpy(film)
reduce_film(t)
if rank == 0:
    np.save(sys.argv[2], t.numpy())
dist.destroy_process_group()
"""


def test_sample_ranges_tile():
    for ns in (1, 5, 8, 1024):
        for world in (1, 2, 3, 8):
            blocks = [sample_range(r, world, ns) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == ns
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1


def test_two_ranks_sum_to_the_single_process_image(tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "film.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", str(worker), ROOT, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=str(tmp_path))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    mesh, records, cam, oc = oracle.cornell_setup(os.path.join(ROOT, "tests", "golden", "cornell"), 16, 12)
    sc = oracle.Scene(mesh)
    sc.set_materials(records)
    full, _ = sc.render(oc, 16, 12, 5, 30, oracle.ALG_MIS, seed=11)
    got = np.load(out)
    assert np.allclose(got, full, rtol=1e-12, atol=0) and full.sum() > 0
