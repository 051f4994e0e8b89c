#!/usr/bin/env python3
"""Benchmark of the MI355X path-tracing backend.

  python bench.py --gpus N --steps K --warmup W

Workload (N=1): the Cornell box at 1920x1080 through the NEE+MIS wavefront integrator with 30
spectral bands.  One step = one batch of `--spp-per-step` samples per pixel (default 128, i.e. 265 M
camera samples, the batch the wavefront queues are sized for on a GPU with 288 GB); K steps are timed after W warm-up
steps; value = samples per pixel per second over all GPUs.  With N>1 every rank renders its own
sample indices of the same image (weak scaling: K steps per rank) into its own film, and the films
are summed with one RCCL all-reduce inside the timed region.

The JSON line also carries
  roofline      for the dominant kernel (closest-hit traversal).  On this cache-resident scene the kernel is bound by its own vector
                instruction stream: `frac` = the time its instructions take at the chip's sustained issue rate for its mix
                (tools/calib/valu_issue_rate, profiles/r03_valu_issue_rate.txt; instruction counts from rocprofv3 PMC passes over
                this same command, committed under profiles/) over the kernel's time measured here with HIP events on its stream;
                `hbm` / `frac_hbm` = the measured bytes on the memory side of the L2 per step over the kernel's time per step,
                against 8 TB/s; `algorithmic` prices the same launches by work units counted on the CPU (nodes fetched and triangle
                records tested in the 8-wide tree the kernel walks, and the binary-tree formula 32 + 16 + 64 * nodes + 48 * triangles);
  cpu_baseline  the fp64 CPU restatement of the reference renderer (oracle/) timed on this host's cores on a bounded sample of the
                same workload;
  aux           the incoherent closest-hit rate on the 1M- and 10M-triangle random soups, each with its own traffic figures.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # bytes/s
PROFILE_STEPS = 10  # steps of the command the committed PMC profile was taken over (python bench.py --no-cpu --no-soup: 2 warm-up + 8 timed)


# the closest-hit kernel's instantiations (stack in LDS or in scratch; reference triangle test; cache-resident scene or not) as
# rocprofv3 names them; the three-parameter names are those of summaries recorded before the last parameter existed
CLOSEST_KERNELS = ("k_trace<true, false, false, true>", "k_trace<true, false, false, false>", "k_trace<false, false, false, true>",
                   "k_trace<false, false, false, false>", "k_trace<true, false, false>", "k_trace<false, false, false>")


def recorded_traffic(pattern, kernels):
    """HBM-side bytes per launch of the first of `kernels` found in the newest committed PMC summary matching `pattern`
    (tools/profile.sh + tools/traffic_from_profile.py on this same command); None when there is none"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    rec = json.load(open(files[-1]))
    for name in kernels:
        k = rec["kernels"].get(name)
        if k and k.get("dispatches"):
            return k, os.path.relpath(files[-1], ROOT)
    return None, None


def recorded_counters(pattern, kernel):
    """what the committed PMC summary says about `kernel` besides traffic: vector instructions per launch (all, and the fp64 ones), the
    sustained issue rates they are priced at, lane utilisation, the share of wave time spent issuing them and the share spent
    waiting (tools/traffic_from_profile.py); None when there is none"""
    k, f = recorded_traffic(pattern, kernel)
    if not k or "valu_lane_utilization" not in k:
        return None
    out = {name: k.get(name) for name in ("valu_instructions_per_launch", "valu_f64_instructions_per_launch", "valu_issue_rates", "valu_lane_utilization",
                                          "valu_issue_share_of_wave_time", "wait_share_of_wave_time", "wait_inst_share_of_wave_time", "l2_hit_rate")}
    out["valu_issue_frac_in_profile"] = k.get("valu_issue_frac")
    out["source"] = f
    return out


def b_ray_wide(nodes_per_ray, tris_per_ray, tri_record_bytes):
    """algorithmic bytes per closest-hit ray in the tree the kernel walks: ray in (32) + hit out (16) + one 128-byte node per node
    fetched + one triangle record (128 bytes, or 112 in the packed layout of small scenes) per triangle tested"""
    return 32.0 + 16.0 + 128.0 * nodes_per_ray + tri_record_bytes * tris_per_ray


def b_ray(nodes_per_ray, tris_per_ray):
    """algorithmic bytes per closest-hit ray: ray in (32) + hit out (16) + 64 B per interior-node visit
    + 48 B per triangle test of an ordered, t-pruned BVH2 walk"""
    return 32.0 + 16.0 + 64.0 * nodes_per_ray + 48.0 * tris_per_ray


def cpu_leg(args, ctx, capi, scenes, log):
    """CPU side: (1) times the oracle's restatement of the reference renderer on a bounded sample of the
    workload, (2) counts, with the oracle's ordered pruned BVH2 walk over the product's own tree, the
    node visits and triangle tests per ray that the roofline's algorithmic bytes are built from."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), os.cpu_count() or 1, 64))
    out = {}
    # ---- work units per ray on the render's own ray distribution (small render of the same scene/camera)
    cx, cy = args.width // 10, args.height // 10
    mesh, records, cam = scenes.cornell(cx, cy)
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    ctx.render_begin(cx, cy, 4, 30, capi.ALG_MIS, 1234)
    finish = ctx.capture_rays(1 << 20)
    ctx.render_pass(0, 4)
    ctx.render_sync()
    rays = finish()
    sc = oracle.Scene(mesh)
    nodes, tri_index = ctx.bvh2()
    poly, tri, t, nv, nt = sc.trace_bvh2(nodes, tri_index, rays, nthreads=threads)
    out["cornell_nodes_per_ray"] = nv / len(rays)
    out["cornell_tris_per_ray"] = nt / len(rays)
    out["cornell_rays_sampled"] = int(len(rays))
    wnodes, wtri = ctx.wide_bvh()
    _, _, wt, wnv, wnt = sc.trace_wide(wnodes, wtri, rays, nthreads=threads)
    out["cornell_wide_nodes_per_ray"] = wnv / len(rays)
    out["cornell_wide_tris_per_ray"] = wnt / len(rays)
    out["cornell_wide_t_mismatch"] = int((wt != t).sum())
    # GPU results on those rays must be the oracle's (a bench that measures wrong answers is worthless)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    out["cornell_parity_prim_mismatch"] = int((gp != poly).sum())
    out["cornell_parity_t_mismatch"] = int((t64 != t).sum())
    # ---- the reference algorithm on the CPU: same scene, same integrator, bounded sample
    sc.set_materials(records)
    oc = oracle.make_camera(cam)
    oracle.lib().orc_camera_set_lens(ctypes.byref(oc), 60.0, 0.4)
    oracle.lib().orc_camera_set_camera(ctypes.byref(oc), oracle._p(oracle.vec(0, 0, 3)), oracle._p(oracle.vec(0, 0, 2)),
                                        oracle._p(oracle.vec(0, 1, 0)))
    sx, sy, sspp = args.width // 2, args.height // 2, args.cpu_spp
    mesh2, records2, cam2 = scenes.cornell(sx, sy)
    oc.film_width = cam2.film_width
    oracle.lib().orc_camera_set_lens(ctypes.byref(oc), 60.0, 0.4)
    oracle.lib().orc_camera_set_camera(ctypes.byref(oc), oracle._p(oracle.vec(0, 0, 3)), oracle._p(oracle.vec(0, 0, 2)),
                                        oracle._p(oracle.vec(0, 1, 0)))
    sc.build_bvh()
    t0 = time.time()
    film, st = sc.render(oc, sx, sy, sspp, 30, oracle.ALG_MIS, seed=1, nthreads=threads)
    dt = time.time() - t0
    out["cpu"] = {
        "value": st["samples"] / dt / (args.width * args.height), "unit": "spp/s at %dx%d" % (args.width, args.height),
        "cores": threads, "kind": "port",
        "sample": "%dx%d at %d spp (%d camera samples, %.1f s) of the same scene, camera and MIS integrator, "
                  "fp64 restatement of the reference's unpruned BVH traversal and integrator, OpenMP" % (sx, sy, sspp, st["samples"], dt),
        "msamples_per_s": st["samples"] / dt / 1e6, "rays_per_sample": st["closest_rays"] / st["samples"],
    }
    # how the port relates to the reference itself: both timed on the same workload (Cornell 512 x 512 x 64 spp MIS) with 8 threads
    # in the 8-vCPU build container, the reference once (1.26 Msamples/s; it cannot be built in this image), the port by
    # tools/port_over_reference.py (committed result).  value / port_over_reference reads the figure above as the reference's.
    try:
        por = json.load(open(os.path.join(ROOT, "profiles", "r03_port_over_reference.json")))
        out["cpu"]["port_over_reference"] = por["port_over_reference"]
        out["cpu"]["reference_equivalent_value"] = out["cpu"]["value"] / por["port_over_reference"]
        out["cpu"]["port_over_reference_source"] = ("profiles/r03_port_over_reference.json: port %.2f Msamples/s against the reference's %.2f on %d threads, %s"
                                                    % (por["port_msamples_per_s"], por["reference_msamples_per_s"], por["threads"], por["workload"]))
    except (OSError, KeyError, ValueError):
        out["cpu"]["port_over_reference"] = None
    log("cpu leg: %s" % json.dumps(out))
    return out, oracle


def soup_leg(n, sigma, m, tag, ctx, capi, synth, oracle, torch, rank, log, reps=5, host_buffer=False, builder=None, barrier=None):
    """incoherent closest hit on a random soup: Mrays/s, work units per ray, memory-side traffic from the committed PMC summary.
    builder: None leaves the library's choice (the GPU builder above 2^17 triangles); capi.BUILDER_HOST_SAH asks for the binned-SAH
    tree, which takes 50 times longer to build and traces a soup 4-5 % faster"""
    t0 = time.time()
    mesh = synth.soup(n, sigma, seed=12345)
    ctx.set_builder(capi.BUILDER_AUTO if builder is None else builder)
    ctx.upload_mesh(mesh)
    ctx.set_builder(capi.BUILDER_AUTO)
    st = ctx.stats()
    log("soup %s: %d triangles uploaded in %.1f s (BVH build %.2f s, depth %d)" % (tag, n, time.time() - t0, st["bvh_build_seconds"], st["bvh_depth"]))
    rays = synth.incoherent_rays(m, seed=777, start=rank * m)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((m, 4), dtype=torch.int32, device="cuda")
    for _ in range(2):
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync()
    if barrier is not None:
        barrier()
    ctx.profile_start()
    t_wall = time.perf_counter()
    for _ in range(reps):
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
    ctx.render_sync()
    t_wall = time.perf_counter() - t_wall
    prof = ctx.profile_stop()["closest"]
    res = {"triangles": n, "sigma": sigma, "rays_per_launch": m, "launches": reps, "first_ray": rank * m, "wall_ms_per_launch": t_wall * 1e3 / reps,
           "builder": "library's choice (GPU: Morton radix tree, cost-minimising 8-wide cut)" if builder is None else "host binned SAH (rt_set_builder)",
           "kernel_ms_per_launch": prof["ms"] / reps, "mrays_per_s": m * reps / (prof["ms"] * 1e-3) / 1e6,
           "hit_rate": float((d_hits[:, 1] != -1).float().mean().item()), "bvh_build_s": st["bvh_build_seconds"],
           "bvh_nodes": st["bvh_nodes"], "bvh_depth": st["bvh_depth"], "device_mb": st["device_bytes"] / 1e6}
    # the library traces large unordered batches in large scenes in the order of the cells their origins lie in (64 cells; the
    # ordering and the way back are inside the time above); the same batch without, for comparison
    before = os.environ.get("MI355RT_RAY_BINS")
    os.environ["MI355RT_RAY_BINS"] = "0"
    try:
        ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
        ctx.render_sync()
        ctx.profile_start()
        for _ in range(reps):
            ctx.trace_closest_device(d_rays.data_ptr(), m, d_hits.data_ptr())
        ctx.render_sync()
        plain = ctx.profile_stop()["closest"]
    finally:
        if before is None:
            del os.environ["MI355RT_RAY_BINS"]
        else:
            os.environ["MI355RT_RAY_BINS"] = before
    res["unordered_kernel_ms_per_launch"] = plain["ms"] / reps
    res["unordered_mrays_per_s"] = m * reps / (plain["ms"] * 1e-3) / 1e6
    res["ray_ordering"] = ("library's choice: batches of at least 2^22 rays in scenes of at least 2M triangles are traced in the order of their "
                           "origins' cells, ordering and un-ordering kernels included in kernel_ms_per_launch; unordered_* is the same batch with MI355RT_RAY_BINS=0")
    if host_buffer:
        # the host-buffer entry point (rt_trace_closest: rays and hits cross PCIe), for the record only
        ctx.trace_closest(rays)  # the first call allocates the staging buffers, the pinned pieces and the copy streams
        t0 = time.perf_counter()
        ctx.trace_closest(rays)
        res["host_buffer_mrays_per_s"] = m / (time.perf_counter() - t0) / 1e6
    if oracle is not None:
        # work units per ray and parity on a sample: the oracle walks the very tree the kernel walked
        sc = oracle.Scene(mesh)
        sub = rays[:1 << 16]
        wnodes, wtri = ctx.wide_bvh()
        poly, tri, t, nv, nt = sc.trace_wide(wnodes, wtri, sub)
        h = d_hits[:len(sub)].cpu().numpy()
        res["parity_rays"] = int(len(sub))
        res["parity_prim_mismatch"] = int((h[:, 1] != poly).sum())
        res["parity_t_mismatch"] = int((h[:, 0].view(np.float32) != t.astype(np.float32)).sum())
        res["wide_nodes_per_ray"], res["wide_tris_per_ray"] = nv / len(sub), nt / len(sub)
        # and by brute force over every triangle of the soup on a few rays: shares nothing with the product, not even the tree
        nb = 256
        bp, _, bt = sc.trace_closest(sub[:nb], oracle.MODE_LIST_LOWEST)
        res["brute_force_rays"] = nb
        res["brute_force_prim_mismatch"] = int((h[:nb, 1] != bp).sum())
        res["brute_force_t_mismatch"] = int((h[:nb, 0].view(np.float32) != bt.astype(np.float32)).sum())
        res["algorithmic_bytes_per_ray"] = b_ray_wide(res["wide_nodes_per_ray"], res["wide_tris_per_ray"], 128.0)
        res["algorithmic_rate_gbs"] = res["mrays_per_s"] * 1e6 * res["algorithmic_bytes_per_ray"] / 1e9
        res["algorithmic_note"] = ("48 B + 128 B per node fetched + 128 B per triangle record tested, counted by the oracle's walk of the kernel's own tree; "
                                   "part of it is served by the L2, so this rate can exceed what HBM delivers: the traffic figures below are the measured ones")
    tr, tr_file = recorded_traffic("r*_%s_traffic.json" % tag, CLOSEST_KERNELS)
    if tr is not None and tr.get("rays_per_launch", m) == m:
        # what bounds this kernel: the rate at which the memory side of the L2 serves random 128-byte lines
        # (tools/calib/random_lines.hip: 53 G lines/s from HBM, 57 G lines/s from the Infinity Cache)
        res["traffic_bytes_per_launch"] = tr["bytes_per_launch"]
        res["traffic_source"] = tr_file
        res["lines_per_ray"] = tr["read_requests_per_launch"] / m
        res["line_rate_g_per_s"] = tr["read_requests_per_launch"] / (res["kernel_ms_per_launch"] * 1e-3) / 1e9
        res["frac_of_random_line_rate"] = res["line_rate_g_per_s"] / 57.0
        res["traffic_frac_hbm"] = tr["bytes_per_launch"] / (res["kernel_ms_per_launch"] * 1e-3) / HBM_PEAK
    return res


def over_ranks(res, world, dist, torch):
    """adds the whole-job figures to one rank's soup result: every rank traced its own range of the ray sequence on its own replica
    of the scene, between two barriers; the job's rate is the sum of the ranks' rays over the slowest rank's time"""
    res["n_gpus"] = world
    if world == 1:
        res["mrays_per_s_total"] = res["mrays_per_s"]
        return res
    mine = torch.tensor([res["mrays_per_s"], res["kernel_ms_per_launch"], res["unordered_mrays_per_s"], res["hit_rate"], res["wall_ms_per_launch"]],
                        dtype=torch.float64, device="cuda")
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    every = torch.stack(every).cpu().numpy()
    res["per_rank_mrays_per_s"] = [float(x) for x in every[:, 0]]
    res["mrays_per_s_min_rank"], res["mrays_per_s_max_rank"] = float(every[:, 0].min()), float(every[:, 0].max())
    # rays of all ranks over the slowest rank's kernel time per launch (what a job that waits for every rank sees), and the plain sum
    res["mrays_per_s_total"] = float(world * res["rays_per_launch"] / (every[:, 1].max() * 1e-3) / 1e6)
    res["mrays_per_s_sum_of_ranks"] = float(every[:, 0].sum())
    res["unordered_mrays_per_s_total"] = float(every[:, 2].sum())
    res["mrays_per_s_total_wall"] = float(world * res["rays_per_launch"] / (every[:, 4].max() * 1e-3) / 1e6)
    res["hit_rate_per_rank"] = [float(x) for x in every[:, 3]]
    res["sharding"] = "ray ranges: rank r traces rays [r * M, (r + 1) * M) of the sequence on its own replica of the scene, no collective in the data path"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=128)
    ap.add_argument("--cpu-spp", type=int, default=256)
    ap.add_argument("--soup-tris", type=int, default=1000000)
    ap.add_argument("--soup-sigma", type=float, default=0.004)
    ap.add_argument("--soup-rays", type=int, default=1 << 24)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg (roofline then lacks the work-unit counts)")
    ap.add_argument("--no-soup", action="store_true")
    ap.add_argument("--no-soup10m", action="store_true", help="skip the 10M-triangle soup (about 30 s of set-up)")
    ap.add_argument("--algorithm", default="mis", choices=["naive", "nee", "mis", "bdpt"],
                    help="integrator of the timed workload (the headline metric is quoted on mis)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and never
        # touches the GPU itself; the ranks re-enter main() with RANK / WORLD_SIZE set by torch.distributed.run
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the job has %d rank(s): start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    from ray_tracer_amd import capi, parallel, scenes, synth

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU path")
    # rehearsal on a one-GPU box only: BENCH_REHEARSAL=1 puts every rank on device 0 and sums the films with gloo
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = capi.Context(local_rank)

    cpu, oracle = (None, None)
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        cpu, oracle = cpu_leg(args, ctx, capi, scenes, log)

    # ---------------- the timed workload ----------------
    nx, ny, spp = args.width, args.height, args.spp_per_step
    mesh, records, cam = scenes.cornell(nx, ny)
    ctx.upload_mesh(mesh)
    ctx.upload_materials(records)
    ctx.set_camera(cam)
    per_rank = (args.warmup + args.steps) * spp
    ns_total = per_rank * world
    assert parallel.sample_range(rank, world, ns_total) == (rank * per_rank, per_rank)
    film = torch.zeros(nx * ny * capi.NBINS, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # torch fills on its own stream, the library renders on the context's
    alg = {"naive": capi.ALG_NAIVE, "nee": capi.ALG_NEE, "mis": capi.ALG_MIS, "bdpt": capi.ALG_NAIVE_BDPT}[args.algorithm]
    ctx.render_begin(nx, ny, ns_total, 30, alg, 20261111, film.data_ptr())
    first = rank * per_rank
    for s in range(args.warmup):
        ctx.render_pass(first + s * spp, spp)
    ctx.render_sync()
    torch.cuda.synchronize()
    if world > 1:
        # the first collective of a given size sets up rings, buffers and kernels: keep that out of the timed region
        scratch = torch.zeros_like(film)
        dist.all_reduce(scratch)
        del scratch
        torch.cuda.synchronize()
        dist.barrier()
    ctx.profile_start()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        ctx.render_pass(first + s * spp, spp)
    ctx.render_sync()
    parallel.reduce_film(film)  # one RCCL sum of the per-rank films over xGMI (no-op on one GPU)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_stop()
    ranks_in = 1
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # n_gpus is what took part, not what was asked for: every rank that went through the film all-reduce adds one
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        ranks_in = int(round(float(one.item())))
        assert ranks_in == args.gpus, "%d ranks entered the all-reduce, --gpus says %d" % (ranks_in, args.gpus)
    stats = ctx.stats()
    film_mean = float(film.sum().item()) / (nx * ny) * (ns_total / (per_rank * world))
    ok = bool(torch.isfinite(film).all().item())

    # the incoherent closest-hit half of the metric.  With N > 1 every rank holds a replica of the soup and traces its own range of the
    # ray sequence, rays [rank * M, (rank + 1) * M): no exchange step, the per-rank rates are gathered afterwards and summed.
    soup = soup10 = None
    barrier = dist.barrier if world > 1 else None
    if not args.no_soup:
        soup = soup_leg(args.soup_tris, args.soup_sigma, args.soup_rays, "soup1m", ctx, capi, synth, oracle, torch, rank, log, host_buffer=(world == 1),
                        builder=capi.BUILDER_HOST_SAH, barrier=barrier)
        soup = over_ranks(soup, world, dist, torch)
        if not args.no_soup10m:
            soup10 = soup_leg(10_000_000, 0.002, args.soup_rays, "soup10m", ctx, capi, synth, oracle, torch, rank, log, reps=3, barrier=barrier)
            soup10 = over_ranks(soup10, world, dist, torch)

    if rank == 0:
        k = prof["closest"]
        avg_ms = k["ms"] / max(k["launches"], 1)
        # What binds the dominant kernel on this scene is its own instruction stream, not HBM (tree and triangles live in the caches):
        # `frac` is the time the kernel's vector instructions would take at the sustained issue rate of this chip for its own mix
        # (tools/calib/valu_issue_rate: the box-test mix and the fp64 mix at 4 waves per SIMD, the kernel's occupancy) over the time
        # the kernel took in this run.  The HBM figures stay beside it (`hbm`): measured bytes per step over the kernel's time per step.
        launches_per_step = k["launches"] / max(args.steps, 1)
        ms_per_step_in_kernel = k["ms"] / max(args.steps, 1)
        roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None,
                "kernel": "k_trace (closest hit)", "kernel_ms_avg": avg_ms, "launches": k["launches"], "launches_per_step": launches_per_step,
                "kernel_ms_per_step": ms_per_step_in_kernel, "rays": k["items"], "kernel_ms_total": k["ms"],
                "mrays_per_s_in_kernel": k["items"] / max(k["ms"], 1e-9) / 1e3}
        default_workload = (nx, ny, spp, args.algorithm) == (1920, 1080, 128, "mis")
        tr, tr_file = recorded_traffic("r*_bench_traffic.json", CLOSEST_KERNELS)
        if tr is not None and default_workload:
            # per step, so that the number of launches the profile averaged over and the number timed here cannot disagree: the
            # profile's bytes per ray times the rays of a step here, over the kernel's time per step here
            bytes_per_step = tr["bytes_per_launch"] * tr["dispatches"] / PROFILE_STEPS
            roof["traffic"] = tr["bytes_per_launch"]
            roof["hbm"] = {"bound": "hbm", "bytes_per_step": bytes_per_step, "achieved": bytes_per_step / (ms_per_step_in_kernel * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "note": "bytes on the memory side of the L2 (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes over this command, %s), "
                                                   "summed over the profiled run's %d steps and divided by them, over the kernel's time per step in this run: only rays and hits "
                                                   "reach HBM on this scene" % (tr_file, PROFILE_STEPS)}
            roof["hbm"]["frac"] = roof["hbm"]["achieved"] * 1e9 / HBM_PEAK
            roof["frac_hbm"] = roof["hbm"]["frac"]
            lim = recorded_counters("r*_bench_traffic.json", CLOSEST_KERNELS)
            if lim is not None and lim.get("valu_issue_rates"):
                rates = lim["valu_issue_rates"]
                n_all = lim["valu_instructions_per_launch"] * tr["dispatches"] / PROFILE_STEPS  # vector instructions of one step
                n_f64 = lim["valu_f64_instructions_per_launch"] * tr["dispatches"] / PROFILE_STEPS
                issue_seconds = (n_all - n_f64) / rates["node"] + n_f64 / rates["f64"]
                roof["achieved"] = n_all / (ms_per_step_in_kernel * 1e-3) / 1e9
                roof["peak"] = n_all / issue_seconds / 1e9  # the sustained rate for this kernel's mix of fp32 and fp64 instructions
                roof["frac"] = issue_seconds / (ms_per_step_in_kernel * 1e-3)
                roof["limiter"] = dict(lim, what="the kernel's own vector instruction stream: frac = time its instructions take at the sustained issue rate (box-test mix %.0f G/s, "
                                                 "fp64 mix %.0f G/s at 4 waves per SIMD) over the time it took; the rest is latency the four waves per SIMD do not cover "
                                                 "(wait_inst_share_of_wave_time: share of wave time spent waiting for a memory instruction to return)"
                                                 % (rates["node"] / 1e9, rates["f64"] / 1e9))
        counts = None
        if cpu is not None:
            counts = dict(cpu, source="counted in this run")
        elif default_workload:
            # no CPU leg (N > 1 or --no-cpu): the work-unit counts of the same scene, camera and integrator from the committed N = 1 run
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
            if files:
                prev = json.load(open(files[-1])).get("roofline", {}).get("algorithmic")
                if prev:
                    counts = {"cornell_nodes_per_ray": prev["bvh2_survey_formula"]["nodes_per_ray"], "cornell_tris_per_ray": prev["bvh2_survey_formula"]["tris_per_ray"],
                              "cornell_wide_nodes_per_ray": prev["bvh8"]["nodes_per_ray"], "cornell_wide_tris_per_ray": prev["bvh8"]["tris_per_ray"],
                              "source": "from " + os.path.relpath(files[-1], ROOT)}
        if counts is not None:
            rays_per_launch = k["items"] / max(k["launches"], 1)
            def priced(nodes, tris, bytes_per_ray, note):
                return {"nodes_per_ray": nodes, "tris_per_ray": tris, "bytes_per_ray": bytes_per_ray, "bytes_per_launch": rays_per_launch * bytes_per_ray,
                        "rate_gbs": k["items"] * bytes_per_ray / (k["ms"] * 1e-3) / 1e9, "note": note}
            roof["algorithmic"] = {
                "work_unit_counts": counts["source"],
                "bvh8": priced(counts["cornell_wide_nodes_per_ray"], counts["cornell_wide_tris_per_ray"],
                               b_ray_wide(counts["cornell_wide_nodes_per_ray"], counts["cornell_wide_tris_per_ray"], 112.0),
                               "48 B + 128 B per node fetched + 112 B per triangle record tested in the 8-wide tree the kernel walks (oracle's walk of that tree "
                               "on a sample of the rendered rays); served by L1/L2 on this scene, so this is cache bandwidth, not HBM traffic"),
                "bvh2_survey_formula": priced(counts["cornell_nodes_per_ray"], counts["cornell_tris_per_ray"],
                                              b_ray(counts["cornell_nodes_per_ray"], counts["cornell_tris_per_ray"]),
                                              "32 + 16 + 64 B per interior-node visit + 48 B per triangle test of an ordered, pruned binary-tree walk: the planning "
                                              "formula; the kernel walks no binary tree, kept for comparison only"),
            }
        line = {
            "metric": "spp/s at 1080p (Cornell box, NEE+MIS wavefront integrator); the incoherent closest-hit Mrays/s half of the metric is under aux",
            "value": args.steps * spp * world / dt, "unit": "spp/s", "n_gpus": ranks_in, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (f64 triangle test)", "dtype_detail": "f32 conservative box tests, shading and film; the triangle test is the reference's f64 arithmetic", "data": "synthetic",
            "config": {"workload": "Cornell box (example/cornell_box_triangulated.obj + materials.json) %dx%d, %s, 30 spectral bands, "
                                   "%d spp per step, %d spp timed per GPU" % (nx, ny, {"mis": "NEE+MIS", "nee": "NEE", "naive": "naive path tracing",
                                                                                      "bdpt": "NaiveBDPT"}[args.algorithm], spp, args.steps * spp),
                       "parallelism": "samples sharded over %d GPU(s), one %s all-reduce of the film" % (world, "gloo (rehearsal: all ranks on device 0)" if rehearsal else "RCCL")},
            "msamples_per_s": args.steps * spp * world * nx * ny / dt / 1e6,
            "rays_per_sample": (stats["closest_rays"] + stats["any_rays"]) / max(stats["camera_samples"], 1),
            "kernels_ms": {name: v["ms"] for name, v in prof.items()},
            "kernel_items": {name: v["items"] for name, v in prof.items()},
            "film_mean_per_pixel": film_mean, "film_finite": ok,
            "roofline": roof,
            "cpu_baseline": cpu["cpu"] if cpu is not None else None,
            "aux": {"closest_hit_soup": soup, "closest_hit_soup_10m": soup10, "parity_on_bench_rays": None if cpu is None else {
                "prim_mismatch": cpu["cornell_parity_prim_mismatch"], "t_mismatch": cpu["cornell_parity_t_mismatch"],
                "wide_walk_t_mismatch": cpu["cornell_wide_t_mismatch"], "rays": cpu["cornell_rays_sampled"]}},
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
