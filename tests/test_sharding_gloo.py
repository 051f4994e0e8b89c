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


RAY_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle
from ray_tracer_amd import synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m = int(sys.argv[3])
sc = oracle.Scene(synth.soup(3000, 0.05, seed=12345))   # every rank holds a replica of the scene
rays = synth.incoherent_rays(m, seed=777, start=rank * m)  # and its own range of the counter-based ray sequence
poly, tri, t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST, nthreads=1)
# the only exchange is the gathering of the per-rank figures afterwards (bench.py: rates; here: hit counts and the results themselves)
mine = torch.tensor([float((poly >= 0).sum()), float(m)], dtype=torch.float64)
every = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(every, mine)
np.savez(sys.argv[2] + ".%d.npz" % rank, poly=poly, t=t, every=torch.stack(every).numpy())
dist.destroy_process_group()
"""


def test_two_ranks_trace_ray_ranges_of_one_sequence(tmp_path):
    """the traversal benchmark's sharding on CPU (the oracle standing in for the device): rank r traces rays [r M, (r + 1) M) of the ray
    sequence, the halves are the whole, and the gathered per-rank figures add up"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from ray_tracer_amd import synth
    m = 4001
    worker = tmp_path / "ray_worker.py"
    worker.write_text(RAY_WORKER)
    out = tmp_path / "hits"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29575", str(worker), ROOT, str(out), str(m)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=str(tmp_path))
    whole = synth.incoherent_rays(2 * m, seed=777)
    assert np.array_equal(whole[m:], synth.incoherent_rays(m, seed=777, start=m))  # the ranges tile the sequence
    poly, tri, t = oracle.Scene(synth.soup(3000, 0.05, seed=12345)).trace_closest(whole, oracle.MODE_LIST_LOWEST, nthreads=2)
    parts = [np.load(str(out) + ".%d.npz" % r) for r in range(2)]
    assert np.array_equal(np.concatenate([p["poly"] for p in parts]), poly) and np.array_equal(np.concatenate([p["t"] for p in parts]), t)
    assert (poly >= 0).sum() > m // 4
    assert parts[0]["every"][:, 0].sum() == (poly >= 0).sum() and np.array_equal(parts[0]["every"], parts[1]["every"])
