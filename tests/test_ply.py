"""PLY reader and bridge (reference ply.cpp:13-158; its PlyModel is commented out, object.cpp:153-233).

CPU tests: the oracle restatement (oracle/ply_oracle.py) and the two product parsers -- C++ (ray_tracer_amd/host/ply.cpp through
rt_cli --dump-ply) and Python (scene_io.Ply) -- against tables recorded from the reference's own compiled ply.cpp
(tests/golden/ply_vectors.npz), bit for bit; plus, where oracle/_ref exists, directly against it on generated files.
GPU tests: a PLY scene through the bridge, closest hits against the oracle, C++ host against Python host."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle
from ray_tracer_amd import scene_io, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PLY_DIR = os.path.join(HERE, "golden", "ply")
RT_CLI = os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli")
FIXTURES = ["cube_normals", "tetra_no_normals", "scan_like_crlf", "ragged", "odd_header", "no_end_header", "truncated_no_newline",
            "no_trailing_newline", "header_only_no_newline"]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "ply_vectors.npz"))


def same(a, b):
    """bit-for-bit, NaN payloads aside"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def cpp_load(path, tmp_path):
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([RT_CLI, "--ply", path, "--dump-ply", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    elements = []
    for line in r.stdout.splitlines():
        t = line.split(" ")
        if t[0] == "element":
            elements.append({"name": " ".join(t[1:-2]), "num_of_elements": int(t[-2]), "num_of_property": int(t[-1])})
    raw = open(out, "rb").read()
    nv, nf, ni, has_n = np.frombuffer(raw, dtype=np.uint64, count=4)
    off = 32
    v = np.frombuffer(raw, dtype=np.float64, count=int(nv) * 6, offset=off).reshape(-1, 6)
    off += int(nv) * 48
    fb = np.frombuffer(raw, dtype=np.uint64, count=int(nf) + 1, offset=off)
    off += (int(nf) + 1) * 8
    fi = np.frombuffer(raw, dtype=np.int32, count=int(ni), offset=off)
    return elements, v, fb, fi


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_restatement_against_reference_tables(golden, name):
    elements, vertices, faces = oracle.ply_oracle().load(os.path.join(PLY_DIR, name + ".ply"))
    assert elements == json.loads(bytes(golden[name + "_elements"]).decode())
    assert same(np.asarray(vertices).reshape(-1, 6), golden[name + "_vertices"])
    fb = np.concatenate([[0], np.cumsum([len(f) for f in faces])]).astype(np.uint64)
    assert np.array_equal(fb, golden[name + "_face_begin"])
    assert np.array_equal(np.asarray([i for f in faces for i in f], dtype=np.int32), golden[name + "_face_index"])


@pytest.mark.parametrize("name", FIXTURES)
def test_python_parser_against_reference_tables(golden, name):
    p = scene_io.Ply()
    assert p.Load(os.path.join(PLY_DIR, name + ".ply"))
    assert p.elements == json.loads(bytes(golden[name + "_elements"]).decode())
    assert same(p.vertices, golden[name + "_vertices"])
    assert np.array_equal(p.face_begin, golden[name + "_face_begin"])
    assert np.array_equal(p.face_index, golden[name + "_face_index"])


@pytest.mark.parametrize("name", FIXTURES)
def test_cpp_parser_against_reference_tables(golden, name, tmp_path):
    elements, v, fb, fi = cpp_load(os.path.join(PLY_DIR, name + ".ply"), tmp_path)
    assert elements == json.loads(bytes(golden[name + "_elements"]).decode())
    assert same(v, golden[name + "_vertices"])
    assert np.array_equal(fb, golden[name + "_face_begin"])
    assert np.array_equal(fi, golden[name + "_face_index"])


def random_ply(path, rng, nv, nf, normals, mixed):
    v = rng.normal(size=(nv, 3)) * rng.choice([1e-3, 1.0, 1e3], (nv, 1))
    n = rng.normal(size=(nv, 3)) if normals else None
    if mixed:
        faces = [rng.integers(0, nv, rng.integers(3, 6)).tolist() for _ in range(nf)]
    else:
        faces = rng.integers(0, nv, (nf, 3))
    scene_io.write_ply(path, v, faces, n)
    return v, n, faces


@pytest.mark.parametrize("normals,mixed", [(True, False), (False, False), (True, True)])
def test_parsers_agree_on_generated_files(tmp_path, normals, mixed):
    """larger files (vectorised path of the Python parser, from_chars path of the C++ one) against the oracle and,
    where it was built, against the compiled reference itself"""
    rng = np.random.default_rng(7)
    path = str(tmp_path / "gen.ply")
    v, n, faces = random_ply(path, rng, 3000, 5000, normals, mixed)
    el, ov, of = oracle.ply_oracle().load(path)
    ov = np.asarray(ov).reshape(-1, 6)
    ofi = np.asarray([i for f in of for i in f], dtype=np.int32)
    assert same(ov[:, :3], v)  # %.17g round-trips doubles
    p = scene_io.Ply()
    assert p.Load(path)
    assert same(p.vertices, ov) and np.array_equal(p.face_index, ofi)
    elements, cv, cfb, cfi = cpp_load(path, tmp_path)
    assert same(cv, ov) and np.array_equal(cfi, ofi) and np.array_equal(cfb, p.face_begin)
    L = oracle.ref_ply()
    if L is not None:
        rel, rv, rfb, rfi = oracle.ref_ply_load(L, path)
        assert same(rv, ov) and np.array_equal(rfi, ofi) and np.array_equal(rfb, cfb) and rel == el


def test_binary_little_endian_is_the_same_mesh(tmp_path):
    rng = np.random.default_rng(11)
    v = rng.normal(size=(500, 3))
    n = rng.normal(size=(500, 3))
    faces = rng.integers(0, 500, (800, 3))
    a, b = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    scene_io.write_ply(a, v, faces, n)
    scene_io.write_ply(b, v, faces, n, binary=True)
    pa, pb = scene_io.Ply(), scene_io.Ply()
    assert pa.Load(a) and pb.Load(b) and pb.binary and not pa.binary
    assert same(pa.vertices, pb.vertices) and np.array_equal(pa.face_index, pb.face_index)
    ea, va, fba, fia = cpp_load(a, tmp_path)
    eb, vb, fbb, fib = cpp_load(b, tmp_path)
    assert same(va, vb) and np.array_equal(fia, fib) and np.array_equal(fba, fbb)


def test_errors_are_reported_not_crashes(tmp_path):
    bad = str(tmp_path / "bad.ply")
    open(bad, "wb").write(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nend_header\n0\n0\n0\n")
    p = scene_io.Ply()
    assert not p.Load(bad) and "fewer than two elements" in p.error  # the reference indexes elements[1] and crashes
    r = subprocess.run([RT_CLI, "--ply", bad, "--dump-ply", str(tmp_path / "x.bin")], capture_output=True, text=True)
    assert r.returncode == 7 and "fewer than two elements" in r.stderr
    assert not scene_io.Ply().Load(str(tmp_path / "missing.ply"))
    neg = str(tmp_path / "neg.ply")
    open(neg, "wb").write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nelement face 1\nproperty list uchar int i\nend_header\n0\n-3 1 2 3\n")
    q = scene_io.Ply()
    assert not q.Load(neg) and "negative" in q.error


def test_bridge_tables(golden):
    p = scene_io.Ply()
    assert p.Load(os.path.join(PLY_DIR, "ragged.ply"))
    m = p.to_mesh("thing", normals="flat")
    # faces of ragged.ply: [0 1 2], [0 1 3 3], [] (empty line), [4 5 5], [1 20 -3]: the empty one and the one with indices out of range go
    assert m["dropped_faces"] == 2 and m["face_begin"].tolist() == [0, 3, 7, 10] and m["names"] == ["thing"]
    assert m["corners"][:, 1].tolist() == [-1] * 10 and m["corners"][:3, 2].tolist() == [0, 0, 0]
    q = scene_io.Ply()
    assert q.Load(os.path.join(PLY_DIR, "cube_normals.ply"))
    both = q.to_mesh("cube", into=p.to_mesh("thing", normals="smooth"))
    assert both["names"] == ["thing", "cube"] and both["object_begin"].tolist() == [0, 3, 14]
    assert both["corners"][10:, 0].min() == 6 and len(both["v"]) == 14 and len(both["vn"]) == 14
    sm = q.to_mesh("cube", normals="smooth")
    out = sm["vn"] / np.linalg.norm(sm["vn"], axis=1, keepdims=True)
    assert (np.einsum("ij,ij->i", out, q.vertices[:, :3]) > 0).all()  # outward like the file's own normals


# ---------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------
def bumpy_sphere(n_lat, n_lon, rng, centre=(0.0, -0.4, 0.0), radius=0.45):
    th = np.linspace(0.05, np.pi - 0.05, n_lat)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1 + 0.08 * np.sin(5 * T) * np.cos(4 * P))
    v = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T), r * np.sin(T) * np.sin(P)], -1).reshape(-1, 3) + np.asarray(centre)
    idx = np.arange(n_lat * n_lon).reshape(n_lat, n_lon)
    a, b = idx[:-1, :], idx[1:, :]
    a2, b2 = np.roll(a, -1, axis=1), np.roll(b, -1, axis=1)
    faces = np.concatenate([np.stack([a, b, b2], -1).reshape(-1, 3), np.stack([a, b2, a2], -1).reshape(-1, 3)])
    return v, faces


@pytest.fixture(scope="module")
def ctx():
    from ray_tracer_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("normals", ["smooth", "flat"])
def test_gpu_ply_scene_closest_hits(ctx, tmp_path, normals):
    from ray_tracer_amd import capi
    rng = np.random.default_rng(3)
    v, faces = bumpy_sphere(60, 90, rng)
    path = str(tmp_path / "sphere.ply")
    scene_io.write_ply(path, v, faces)
    p = scene_io.Ply()
    assert p.Load(path) and not p.has_normals()
    mesh = p.to_mesh("blob", normals=normals)
    assert mesh["dropped_faces"] == 0 and len(mesh["face_begin"]) - 1 == len(faces)
    ctx.upload_mesh(mesh)
    sc = oracle.Scene(mesh)
    rays = synth.incoherent_rays(40000, seed=31)
    hits, t64 = ctx.trace_closest(rays, want_t64=True)
    o_poly, o_tri, o_t = sc.trace_closest(rays, oracle.MODE_LIST_LOWEST)
    gp = hits["prim"].astype(np.int64)
    gp[hits["prim"] == capi.MISS] = -1
    assert (o_poly >= 0).mean() > 0.05
    tie = (gp != o_poly) & (t64 == o_t)  # exact ties in t between neighbouring polygons on shared edges
    assert ((gp == o_poly) | tie).all() and tie.sum() <= 5
    assert np.array_equal(t64[o_poly >= 0], o_t[o_poly >= 0])


@pytest.mark.gpu
def test_gpu_ply_object_in_cornell_cpp_host_matches_python_host(tmp_path):
    """the C++ host (Renderer::LoadPly through rt_cli) and the Python host render the same film for OBJ + PLY"""
    from ray_tracer_amd import capi, scenes
    from ray_tracer_amd.renderer import Renderer, MIS
    cdir = os.path.join(HERE, "golden", "cornell")
    rng = np.random.default_rng(3)
    v, faces = bumpy_sphere(24, 36, rng, centre=(0.0, 0.1, 0.0), radius=0.25)
    path = str(tmp_path / "blob.ply")
    scene_io.write_ply(path, v, faces)
    mats = json.load(open(os.path.join(cdir, "materials.json")))
    omm = json.load(open(os.path.join(cdir, "objmatmap.json")))
    first_material = next(iter(omm.values()))
    omm["blob"] = first_material
    mpath = str(tmp_path / "objmatmap.json")
    json.dump(omm, open(mpath, "w"))
    film_path = str(tmp_path / "film.f64")
    w, h, ns, seed = 48, 32, 8, 77
    r = subprocess.run([RT_CLI, "--obj", os.path.join(cdir, "cornell_box_triangulated.obj"), "--materials", os.path.join(cdir, "materials.json"),
                        "--objmatmap", mpath, "--ply", path, "--ply-object", "blob", "--width", str(w), "--height", str(h),
                        "--sample", str(ns), "--seed", str(seed), "--film", film_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cpp = np.fromfile(film_path, dtype=np.float64).reshape(w * h, 30)
    # the same through the Python host
    rr = Renderer(0, seed=seed)
    rr.Load(os.path.join(cdir, "cornell_box_triangulated.obj"))
    assert rr.LoadPly(path, "blob")
    named = scene_io.load_materials(os.path.join(cdir, "materials.json"))
    assert rr.LoadMaterials({obj: named[m] for obj, m in omm.items()})
    rr.cam.film_height = 0.024
    rr.cam.film_width = 0.024 * w / h
    rr.cam.SetLens(60.0, 0.4)
    rr.cam.set_Camera((0, 0, 3), (0, 0, 2), (0, 1, 0))
    rr.algorithm_type = MIS
    rr.preview_img_flag = False
    rr.samples_per_pass = ns
    rr.RenderImage(w, h, ns, 30)
    assert cpp.sum() > 0 and np.allclose(cpp, rr.spectrum_img, rtol=1e-5, atol=1e-9)  # same kernels, same inputs; fp32 atomics may reorder sums
    # and the blob is really there: the film differs from the one without it
    r0 = Renderer(0, seed=seed)
    r0.Load(os.path.join(cdir, "cornell_box_triangulated.obj"))
    del omm["blob"]
    assert r0.LoadMaterials({obj: named[m] for obj, m in omm.items()})
    r0.cam = rr.cam
    r0.algorithm_type = MIS
    r0.preview_img_flag = False
    r0.samples_per_pass = ns
    r0.RenderImage(w, h, ns, 30)
    assert not np.allclose(r0.spectrum_img, rr.spectrum_img, rtol=1e-2, atol=1e-6)
