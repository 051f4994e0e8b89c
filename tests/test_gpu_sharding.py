"""The N>1 product path on one GPU: two ranks (both on device 0, gloo standing in for RCCL) render their sample ranges through the C ABI
into caller-owned films (torch tensors handed to rt_render_begin), sum them with parallel.reduce_film, and must reproduce the film a
single context renders for all samples.  This is the hand-off bench.py uses for --gpus N (its BENCH_REHEARSAL=1 mode)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ray_tracer_amd import capi, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ray_tracer_amd import capi, scenes
from ray_tracer_amd.parallel import sample_range, reduce_film
nx, ny, ns, seed = (int(a) for a in sys.argv[3:7])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
ctx = capi.Context(0)
mesh, records, cam = scenes.cornell(nx, ny)
ctx.upload_mesh(mesh); ctx.upload_materials(records); ctx.set_camera(cam)
film = torch.zeros(nx * ny * capi.NBINS, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()  # torch fills on its own stream, the library renders on the context's
ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, seed, film.data_ptr())
first, count = sample_range(rank, world, ns)
# in passes of two, so that paths handed from one pass to the next take part
for s in range(first, first + count, 2):
    ctx.render_pass(s, min(2, first + count - s))
ctx.render_sync()
torch.cuda.synchronize()
own = float(film.sum().item())
reduce_film(film)
torch.cuda.synchronize()
sums = [None] * world
dist.all_gather_object(sums, (first, count, own, ctx.stats()["camera_samples"]))
if rank == 0:
    np.save(sys.argv[2], film.cpu().numpy())
    import json
    json.dump(sums, open(sys.argv[2] + ".json", "w"))
ctx.close()
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_sum_to_the_single_context_film(tmp_path):
    nx, ny, ns, seed = 160, 90, 7, 31
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "film.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", str(worker), ROOT, str(out), str(nx), str(ny), str(ns), str(seed)]
    subprocess.run(cmd, check=True, env=env, timeout=280, cwd=str(tmp_path))
    got = np.load(out).astype(np.float64).reshape(nx * ny, capi.NBINS)
    per_rank = json.load(open(str(out) + ".json"))
    assert [(f, c) for f, c, _, _ in per_rank] == [(0, 4), (4, 3)]          # the ranks' sample blocks tile 0..ns
    assert [n for _, _, _, n in per_rank] == [4 * nx * ny, 3 * nx * ny]      # and each rendered exactly its own
    ctx = capi.Context(0)
    mesh, records, cam = scenes.cornell(nx, ny)
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    ctx.render_begin(nx, ny, ns, 30, capi.ALG_MIS, seed)
    ctx.render_pass(0, ns)
    ctx.render_sync()
    whole = ctx.download_film()
    ctx.close()
    assert whole.sum() > 0 and np.isfinite(got).all()
    assert abs(sum(o for _, _, o, _ in per_rank) - whole.sum()) <= 1e-5 * whole.sum()
    # fp32 atomics add in a different order in every run: equality up to rounding of the sums
    assert np.allclose(got, whole, rtol=1e-4, atol=1e-8)


RAY_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ray_tracer_amd import capi, synth
n_tri, m = int(sys.argv[3]), int(sys.argv[4])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
ctx = capi.Context(0)
ctx.upload_mesh(synth.soup(n_tri, 0.01, seed=12345))       # every rank holds a replica of the scene
rays = synth.incoherent_rays(m, seed=777, start=rank * m)  # and traces its own range of the ray sequence: no exchange step
d_rays = torch.from_numpy(rays).cuda()
d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
d_t64 = torch.empty(m, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr(), d_t64.data_ptr())
ctx.render_sync()
np.savez(sys.argv[2] + ".%d.npz" % rank, hits=d_hits.cpu().numpy(), t64=d_t64.cpu().numpy())
dist.barrier()
ctx.close()
dist.destroy_process_group()
"""


def test_two_ranks_trace_their_ray_ranges_of_one_sequence(tmp_path):
    """the traversal benchmark's sharding (bench.py's soup legs at --gpus N): rank r traces rays [r M, (r + 1) M) of the counter-based
    ray sequence on its own replica of the scene; the two ranks' hits and fp64 t are the matching halves of one context's trace of
    rays [0, 2 M)"""
    n_tri, m = 120000, 300001
    worker = tmp_path / "ray_worker.py"
    worker.write_text(RAY_WORKER)
    out = tmp_path / "hits"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(worker), ROOT, str(out), str(n_tri), str(m)]
    subprocess.run(cmd, check=True, env=env, timeout=280, cwd=str(tmp_path))
    from ray_tracer_amd import synth
    ctx = capi.Context(0)
    ctx.upload_mesh(synth.soup(n_tri, 0.01, seed=12345))
    hits, t64 = ctx.trace_closest(synth.incoherent_rays(2 * m, seed=777), want_t64=True)
    ctx.close()
    assert (hits["prim"] != capi.MISS).sum() > m // 2
    for rank in range(2):
        got = np.load(str(out) + ".%d.npz" % rank)
        part = hits[rank * m:(rank + 1) * m]
        assert np.array_equal(got["hits"].view(np.uint32)[:, 1], part["prim"]) and np.array_equal(got["hits"].view(np.uint32)[:, 2], part["tri"])
        assert np.array_equal(got["t64"], t64[rank * m:(rank + 1) * m])


def test_group_of_contexts_renders_the_single_context_film():
    """rt_group_*: three contexts (all on device 0 here; one per GPU on a node) split every pass's samples; the summed film and its
    tonemap equal a single context's, for the unidirectional and the bidirectional integrator, in one pass or many"""
    nx, ny, ns = 96, 54, 7
    mesh, records, cam = scenes.cornell(nx, ny)
    one = capi.Context(0)
    grp = capi.Group([0, 0, 0])
    assert grp.size() == 3
    for c in (one, grp):
        c.upload_mesh(mesh)
        c.upload_materials(records)
        c.set_camera(cam)
    for alg in (capi.ALG_MIS, capi.ALG_NAIVE_BDPT):
        one.render_begin(nx, ny, ns, 30, alg, 5)
        one.render_pass(0, ns)
        want, want_preview = one.download_film(), one.download_preview(ns)
        grp.render_begin(nx, ny, ns, 30, alg, 5)
        grp.render_pass(0, ns)
        got = grp.download_film()
        assert np.allclose(got, want, rtol=1e-4, atol=1e-8) and want.sum() > 0
        assert np.allclose(grp.download_preview(ns), want_preview, atol=1e-5)
        samples = [grp.member(i).stats()["camera_samples"] for i in range(3)]
        assert samples == [3 * nx * ny, 2 * nx * ny, 2 * nx * ny]  # 7 samples over 3 members
        grp.render_begin(nx, ny, ns, 30, alg, 5)
        for s, n in ((0, 1), (1, 4), (5, 2)):  # passes smaller than the group: some members sit a pass out
            grp.render_pass(s, n)
        assert np.allclose(grp.download_film(), want, rtol=1e-4, atol=1e-8)
    # failures name the member; the group stays usable
    with pytest.raises(capi.BackendError) as e:
        grp.upload_materials([dict(r, light_flag=False) for r in records])
    assert e.value.code == capi.ERR_NO_LIGHTS and "member" in str(e.value)
    with pytest.raises(capi.BackendError):
        capi.Group([0, 99])
    grp.upload_materials(records)
    grp.render_begin(nx, ny, 2, 30, capi.ALG_NEE, 1)
    grp.render_pass(0, 2)
    assert grp.download_film().sum() > 0
    grp.close()
    one.close()


def test_group_traces_ray_ranges_and_sums_films_in_one_pass():
    """rt_group_trace_closest: three members answer contiguous thirds of the rays, the whole equals one context's answers; and the
    films of a 4K render are gathered and summed by one kernel (the time of that is recorded for the log)"""
    from ray_tracer_amd import synth
    mesh = synth.soup(150000, 0.01, seed=3)
    rays = synth.incoherent_rays(400003, seed=9)
    one = capi.Context(0)
    grp = capi.Group([0, 0, 0])
    one.upload_mesh(mesh)
    grp.upload_mesh(mesh)
    h1, t1 = one.trace_closest(rays, want_t64=True)
    hg, tg = grp.trace_closest(rays, want_t64=True)
    assert (h1["prim"] != capi.MISS).sum() > 100000
    assert np.array_equal(h1, hg) and np.array_equal(t1, tg)
    assert np.array_equal(grp.trace_closest(rays[:5]), h1[:5])  # fewer rays than there is to share out evenly
    nx, ny = 3840, 2160
    mesh, records, cam = scenes.cornell(nx, ny)
    for c in (one, grp):
        c.upload_mesh(mesh)
        c.upload_materials(records)
        c.set_camera(cam)
    one.render_begin(nx, ny, 3, 30, capi.ALG_MIS, 5)
    one.render_pass(0, 3)
    want = one.download_film()
    grp.render_begin(nx, ny, 3, 30, capi.ALG_MIS, 5)
    grp.render_pass(0, 3)
    got = grp.download_film()
    assert np.allclose(got, want, rtol=1e-4, atol=1e-8) and want.sum() > 0
    print("gather and sum of three 3840 x 2160 x 30 films on one GPU: %.2f ms" % grp.last_sum_ms())
    assert 0.0 < grp.last_sum_ms() < 100.0
    grp.close()
    one.close()


def test_cpp_host_on_two_contexts_matches_one(tmp_path):
    """the C++ host (Renderer::UseDevices through rt_cli --devices) on two contexts against the same program on one"""
    cornell_dir = os.path.join(ROOT, "tests", "golden", "cornell")
    cli = os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli")
    films = {}
    for tag, extra in (("one", []), ("two", ["--devices", "0,0"])):
        out = tmp_path / (tag + ".f64")
        subprocess.run([cli, "--obj", os.path.join(cornell_dir, "cornell_box_triangulated.obj"), "--materials",
                        os.path.join(cornell_dir, "materials.json"), "--objmatmap", os.path.join(cornell_dir, "objmatmap.json"),
                        "--width", "64", "--height", "36", "--sample", "6", "--algorithm", "mis", "--seed", "9", "--film", str(out)] + extra,
                       check=True, timeout=120)
        films[tag] = np.fromfile(out, dtype=np.float64).reshape(-1, 30)
    assert films["one"].sum() > 0 and np.allclose(films["one"], films["two"], rtol=1e-4, atol=1e-8)
