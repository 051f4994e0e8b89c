"""Host logic that runs without a GPU: file formats, material folding, camera, synthetic inputs,
and the C-ABI library's exported symbols."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ray_tracer_amd import capi, scene_io, synth
from ray_tracer_amd.camera import LensCamera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    lib = C.CDLL(capi.LIB_PATH)  # loads without a GPU; no compute call is made here
    for name in declared:
        assert hasattr(lib, name), name


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.BackendError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE and "no CPU path" in str(e.value)


def test_obj_parser(cornell_dir, tmp_path):
    o = scene_io.Obj()
    assert o.Load(os.path.join(cornell_dir, "cornell_box_triangulated.obj"))
    assert o.object_names() == ["Plane", "Plane.001", "Plane.002", "Plane.003", "Plane.004", "Plane.005", "Sphere_Sphere.001"]
    m = o.to_mesh()
    assert len(m["v"]) == 506 and len(m["face_begin"]) - 1 == 972 and m["corners"].min() >= -1
    assert (np.diff(m["face_begin"].astype(np.int64)) == 3).all()
    q = scene_io.Obj()
    assert q.Load(os.path.join(cornell_dir, "cornell_box.obj"))
    assert np.diff(q.to_mesh()["face_begin"].astype(np.int64)).max() == 4
    # quirks: no `o` line before data -> error (obj.cpp:25-27); missing vn -> refused at flatten time
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\n")
    assert not scene_io.Obj().Load(str(bad))
    novn = tmp_path / "novn.obj"
    novn.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    r = scene_io.Obj()
    assert r.Load(str(novn))
    with pytest.raises(ValueError):
        r.to_mesh()
    trail = tmp_path / "trail.obj"
    trail.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 \n")
    t = scene_io.Obj()
    assert t.Load(str(trail))
    assert len(t.objects[0]["faces"][0]) == 4  # the trailing blank repeats the last corner (obj.cpp:105-106)


def test_material_folding(cornell_dir):
    mats = scene_io.load_materials(os.path.join(cornell_dir, "materials.json"))
    assert set(mats) == {"white", "light", "green", "red"}
    w, l, g, r = (mats[k].record() for k in ("white", "light", "green", "red"))
    assert w["type"] == capi.MAT_LAMBERTIAN and np.all(w["a"] == 1.0) and not w["light_flag"]
    assert l["type"] == capi.MAT_DIFFUSE_LIGHT and l["light_flag"]
    # RGBColor -> RGBtoSpectrum -> albedo: thirds B, G, R (spectrum.cpp:219-232)
    assert np.all(r["a"][:20] == 0.0) and np.all(r["a"][20:] == 1.0)
    assert np.all(g["a"][:10] == 0.0) and np.all(g["a"][10:20] == 1.0) and np.allclose(g["a"][20:], 0.10714244842529297)
    # a graph with nothing wired to the output is a null BSDF
    none = scene_io.NodeMaterial({"name": "x", "nodes": [{"type": "UV"}, {"type": "Output"}], "connections": []})
    assert none.record()["type"] == capi.MAT_NONE
    # an ImageTexture without a file leaves its output untouched (materialnode.cpp:1003-1004): the albedo folds to zero
    tex = {"name": "t", "nodes": [{"type": "UV"}, {"type": "Output"}, {"type": "Lambertian", "albedo": [1] * 30},
                                  {"type": "RGBtoSpectrum"}, {"type": "ImageTexture"}],
           "connections": [{"input_node": 1, "input_slot": "BSDF", "output_node": 2, "output_slot": "BSDF"},
                           {"input_node": 2, "input_slot": "albedo", "output_node": 3, "output_slot": "Spectrum"},
                           {"input_node": 3, "input_slot": "RGB", "output_node": 4, "output_slot": "Out"}]}
    flat = scene_io.NodeMaterial(tex).record()
    assert "graph" not in flat and not flat["a"].any()
    # with a texture and the UV node wired in, the albedo becomes a per-hit program instead of a constant
    tex["nodes"][4]["path"] = "mem"
    tex["connections"].append({"input_node": 4, "input_slot": "UV", "output_node": 0, "output_slot": "Out->"})
    texels = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    rec = scene_io.NodeMaterial(tex, textures={"mem": texels}).record()
    assert rec["graph"][0] == "R2S" and rec["graph"][1][0] == "IMG" and rec["graph"][1][2] == ("UV",)
    g = scene_io.compile_material_graphs([flat, rec])
    assert g["programs"].tolist() == [[1, 0, 3]] and g["code"].tolist() == [[scene_io.OP_UV, 0], [scene_io.OP_IMAGE, 0], [scene_io.OP_RGB2SPEC, 0]]
    assert scene_io.compile_material_graphs([flat]) is None
    # ValueNoise has no vector Compute: wired to an RGB input it leaves the vector untouched, and the albedo folds to zero
    tex["nodes"][4] = {"type": "ValueNoise", "u_size": 4}
    flat = scene_io.NodeMaterial(tex).record()
    assert "graph" not in flat and not flat["a"].any()
    # GGXReflection (undefined behaviour in the reference) becomes the backend's defined variant: n, k carried, alpha from "roughness"
    tex["nodes"][2] = {"type": "GGXReflection", "n": [0.2] * 30, "k": [3.0] * 30}
    ggx = scene_io.NodeMaterial(tex).record()
    assert ggx["type"] == capi.MAT_GGX and ggx["roughness"] == 0.5 and np.all(ggx["a"] == 0.2) and np.all(ggx["b"] == 3.0)
    tex["nodes"][2]["roughness"] = 0.25
    assert scene_io.NodeMaterial(tex).record()["roughness"] == 0.25
    # a node type that is no BSDF at the output is refused, not silently flattened
    tex["nodes"][2] = {"type": "Checkerboard"}
    with pytest.raises(scene_io.UnsupportedMaterialGraph):
        scene_io.NodeMaterial(tex).record()
    # Dielectric: the "surface_color" lookup never matches the slot "surface color" -> own array
    die = {"name": "d", "nodes": [{"type": "UV"}, {"type": "Output"},
                                  {"type": "Dielectric", "n": [1.5] * 30, "surface_color": [0.9] * 30}, {"type": "Spectrum", "data": [0.1] * 30}],
           "connections": [{"input_node": 1, "input_slot": "BSDF", "output_node": 2, "output_slot": "BSDF"},
                           {"input_node": 2, "input_slot": "surface color", "output_node": 3, "output_slot": "Out->"}]}
    d = scene_io.NodeMaterial(die).record()
    assert d["type"] == capi.MAT_DIELECTRIC and np.all(d["a"] == 1.5) and np.all(d["b"] == 0.9)


def test_project_file(cornell_dir):
    pj = scene_io.load_project(os.path.join(cornell_dir, "testproject.json"))
    assert pj["model"].endswith("cornell_box.obj") and os.path.exists(pj["model"])
    assert pj["vfov"] == 60.0 and pj["camera_pos"] == [0, 0.0, 3.0]
    omm = scene_io.load_objmatmap(pj["objmatmap"])
    assert omm["Plane.005"] == "light"


def test_camera_matches_oracle(cornell_dir):
    import oracle
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 1920, 1080)
    for name in ("origin", "film_lower_left_corner", "horizontal", "vertical", "w"):
        assert np.array_equal(np.asarray(getattr(cam, name)), np.array(list(getattr(oc, name))))
    assert cam.d == oc.d and cam.lens_maker_coefficient == oc.lens_maker_coefficient
    assert abs(cam.LensMakersFormula(550.0) - 0.4) < 1e-15


def test_synthetic_inputs_are_reproducible():
    a, b = synth.soup(1000, 0.01, seed=12345), synth.soup(1000, 0.01, seed=12345)
    assert np.array_equal(a["v"], b["v"])
    big = synth.soup(3000, 0.01, seed=12345, chunk=1024)
    assert np.array_equal(big["v"][:3000], a["v"][:3000])  # chunking does not change values
    r = synth.incoherent_rays(1000, seed=777)
    assert r.dtype == np.float32 and r.shape == (1000, 8)
    assert np.allclose(np.linalg.norm(r[:, 4:7], axis=1), 1.0, atol=1e-6)
    r2 = synth.incoherent_rays(500, seed=777, start=500)
    assert np.array_equal(r2, r[500:])
    assert np.abs(a["v"]).max() < 1.2 and np.all(np.isinf(r[:, 7]))


def test_cpp_host_folds_materials_like_the_python_host(cornell_dir, tmp_path):
    """the C++ host mirror (ray_tracer_amd/host) and the Python one produce the same per-object records"""
    import subprocess
    cli = os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli")
    rec = tmp_path / "records.f64"
    p = subprocess.run([cli, "--obj", os.path.join(cornell_dir, "cornell_box_triangulated.obj"), "--materials",
                        os.path.join(cornell_dir, "materials.json"), "--objmatmap", os.path.join(cornell_dir, "objmatmap.json"),
                        "--records", str(rec)], capture_output=True, text=True)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode == 3 and "no CPU path" in p.stderr  # the driver refuses to run without a device
    got = np.fromfile(rec, dtype=np.float64).reshape(-1, 63)
    from ray_tracer_amd import scenes
    mesh, records, cam = scenes.cornell(64, 64)
    assert len(got) == len(records) == 7
    for g, r in zip(got, records):
        assert int(g[0]) == r["type"] and bool(g[1]) == r["light_flag"]
        assert np.array_equal(g[2:32], r["a"]) and np.array_equal(g[32:62], r["b"])
    # the same with a GGXReflection node (the backend's defined variant) on the white material: alpha from "roughness", else 0.5
    import json
    mats = json.load(open(os.path.join(cornell_dir, "materials.json")))
    for rough in (None, 0.2):
        for m in mats:
            if m["name"] == "white":
                m["nodes"][2] = {"type": "GGXReflection", "name": "GGXReflection", "n": [0.2 + 0.01 * i for i in range(30)], "k": [3.0] * 30,
                                 "input_slots": [], "output_slots": [{"name": "BSDF", "type": "SlotBSDF"}], "pos": {"x": 0.0, "y": 0.0}}
                if rough is not None:
                    m["nodes"][2]["roughness"] = rough
        mfile = tmp_path / "ggx_materials.json"
        mfile.write_text(json.dumps(mats))
        subprocess.run([cli, "--obj", os.path.join(cornell_dir, "cornell_box_triangulated.obj"), "--materials", str(mfile),
                        "--objmatmap", os.path.join(cornell_dir, "objmatmap.json"), "--records", str(rec)], capture_output=True, text=True)
        got = np.fromfile(rec, dtype=np.float64).reshape(-1, 63)
        named = scene_io.load_materials(str(mfile))
        omm = scene_io.load_objmatmap(os.path.join(cornell_dir, "objmatmap.json"))
        want = [named[omm[name]].record() for name in mesh["names"]]
        assert sum(r["type"] == capi.MAT_GGX for r in want) >= 3
        for g, r in zip(got, want):
            assert int(g[0]) == r["type"] and np.array_equal(g[2:32], r["a"]) and np.array_equal(g[32:62], r["b"])
            assert g[62] == r.get("roughness", 0.0) and (r["type"] != capi.MAT_GGX or g[62] == (0.5 if rough is None else rough))


def _walk(mat, uv):
    """The reference's Compute() recursion written out directly (materialnode.cpp), independent of the host's compiler:
    returns slot `albedo` of the Lambertian behind the output node at texture coordinate uv."""
    nodes, parent = mat.nodes, mat._parent

    def scal(i):
        n, t = nodes[i], nodes[i]["type"]
        if t == "AccessVec3Component":
            p = parent(i, "vector")
            return 0.0 if p is None else vec(p)[0]
        if t == "ValueNoise":
            p = parent(i, "input")
            if p is None:
                return 0.0
            r = scene_io.NodeMaterial.noise_table(n["u_size"] + 1)
            u = np.fmod(scal(p), 1.0)
            u = u + 1.0 if u < 0.0 else u
            us = 1.0 / (len(r) - 1)
            rem = np.fmod(u, us)
            x0 = int((u - rem) / us)
            t_ = rem / us
            return (1.0 - t_) * r[x0] + t_ * r[x0 + 1]
        if t == "ValueNoise2D":
            pu, pv = parent(i, "u"), parent(i, "v")
            if pu is None or pv is None:
                return 0.0
            nu, nv = n["u_size"], n["v_size"]
            r = scene_io.NodeMaterial.noise_table((nu + 1) * (nv + 1))
            u, v = np.fmod(scal(pu), 1.0), np.fmod(scal(pv), 1.0)
            u = u + 1.0 if u < 0.0 else u
            v = v + 1.0 if v < 0.0 else v
            us, vs = 1.0 / nu, 1.0 / nv
            urem, vrem = np.fmod(u, us), np.fmod(v, vs)
            u0, v0 = int((u - urem) / us), int((v - vrem) / vs)
            ut, vt = urem / us, vrem / vs
            ut, vt = ut * ut * (3.0 - 2.0 * ut), vt * vt * (3.0 - 2.0 * vt)
            a = (1.0 - ut) * r[u0 * (nv + 1) + v0] + ut * r[(u0 + 1) * (nv + 1) + v0]
            b = (1.0 - ut) * r[u0 * (nv + 1) + v0 + 1] + ut * r[(u0 + 1) * (nv + 1) + v0 + 1]
            return (1.0 - vt) * a + vt * b
        if t in ("Addition", "Multiplication"):
            p0, p1 = parent(i, "In0"), parent(i, "In1")
            if p0 is None:
                return 0.0
            a = scal(p0)
            if p1 is None:
                return a if t == "Addition" else 0.0
            return a + scal(p1) if t == "Addition" else a * scal(p1)
        return 0.0

    def vec(i):
        n, t = nodes[i], nodes[i]["type"]
        if t == "UV":
            return np.asarray(uv, dtype=np.float64)
        if t == "RGBColor":
            return np.asarray(n["col"], dtype=np.float64)
        if t == "CombineVec3Component":
            out = np.zeros(3)
            for k, slot in enumerate(("0", "1", "2")):
                p = parent(i, slot)
                if p is None:
                    break
                out[k] = scal(p)
            return out
        if t == "RodriguesRotation":
            p = parent(i, "n")
            nn = np.asarray(n.get("n", [0, 0, 1]), dtype=np.float64) if p is None else vec(p)
            nn = nn / np.sqrt((nn * nn).sum())
            p = parent(i, "theta")
            th = n.get("theta", 0.0) if p is None else scal(p)
            c, s_ = np.cos(th), np.sin(th)
            p = parent(i, "v")
            if p is None:
                return np.zeros(3)
            x, y, z = nn
            rot = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                            [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                            [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]])
            return rot @ vec(p)
        if t in ("Addition", "Multiplication"):
            p0, p1 = parent(i, "In0"), parent(i, "In1")
            if p0 is None:
                return np.zeros(3)
            a = vec(p0)
            if p1 is None:
                return a if t == "Addition" else np.zeros(3)
            return a + vec(p1) if t == "Addition" else a * vec(p1)
        return np.zeros(3)

    def spec(i):
        n, t = nodes[i], nodes[i]["type"]
        if t == "Spectrum":
            return np.asarray(n["data"], dtype=np.float64)
        if t == "RGBtoSpectrum":
            p = parent(i, "RGB")
            return np.zeros(30) if p is None else scene_io.rgb_to_spectrum(vec(p))
        if t in ("Addition", "Multiplication"):
            p0, p1 = parent(i, "In0"), parent(i, "In1")
            if p0 is None:
                return np.zeros(30)
            a = spec(p0)
            if p1 is None:
                return a if t == "Addition" else np.zeros(30)
            return a + spec(p1) if t == "Addition" else a * spec(p1)
        return np.zeros(30)

    bsdf = parent(1, "BSDF")
    return spec(parent(bsdf, "albedo"))


def noise_material(name="noisy"):
    """UV -> rotate about (1, 2, 3) by 0.7 -> components -> 1D noise, 2D noise and a sum -> vector -> RGBtoSpectrum, times a
    spectrum: every scalar and vector node of materialnode.cpp that reads the texture coordinate"""
    nodes = [{"type": "UV"}, {"type": "Output"}, {"type": "Lambertian", "albedo": [1] * 30},
             {"type": "Multiplication"}, {"type": "Spectrum", "data": list(np.linspace(0.5, 1.0, 30))}, {"type": "RGBtoSpectrum"},
             {"type": "CombineVec3Component"},
             {"type": "ValueNoise", "u_size": 7}, {"type": "ValueNoise2D", "u_size": 5, "v_size": 3}, {"type": "Addition"},
             {"type": "AccessVec3Component"}, {"type": "AccessVec3Component"},
             {"type": "RodriguesRotation", "n": [1.0, 2.0, 3.0], "theta": 0.7},
             {"type": "Multiplication"}, {"type": "RGBColor", "col": [3.0, -2.0, 1.0]}]
    def c(out, inn, slot):
        return {"output_node": out, "output_slot": "o", "input_node": inn, "input_slot": slot}
    conns = [c(2, 1, "BSDF"), c(3, 2, "albedo"), c(4, 3, "In0"), c(5, 3, "In1"), c(6, 5, "RGB"),
             c(7, 6, "0"), c(8, 6, "1"), c(9, 6, "2"),
             c(10, 7, "input"),           # noise1(uv.x)
             c(10, 8, "u"), c(11, 8, "v"),  # noise2(uv.x, rotated.x)
             c(7, 9, "In0"), c(11, 9, "In1"),  # noise1 + rotated.x
             c(13, 12, "v"),              # rotate (uv * colour); listed before the rotation's outgoing link into a slot named
                                          # "vector": GetInputParentNode("v") takes the first link whose slot starts with "v"
             c(0, 10, "vector"), c(12, 11, "vector"),
             c(0, 13, "In0"), c(14, 13, "In1")]
    return {"name": name, "nodes": nodes, "connections": conns}


def test_noise_rotation_and_component_nodes_compile_to_what_the_reference_computes(cornell_dir):
    """host compiler + the checker's program evaluation against the direct recursion, at random texture coordinates
    (negative and > 1 included: the noise nodes wrap them)"""
    import oracle
    mat = scene_io.NodeMaterial(noise_material())
    rec = mat.record()
    assert rec["type"] == capi.MAT_LAMBERTIAN and "graph" in rec
    g = scene_io.compile_material_graphs([rec])
    ops = set(g["code"][:, 0].tolist())
    assert {scene_io.OP_ACCESS, scene_io.OP_COMBINE, scene_io.OP_NOISE1, scene_io.OP_NOISE2, scene_io.OP_ROTATE} <= ops
    mesh, records, cam, oc = oracle.cornell_setup(cornell_dir, 8, 8)
    sc = oracle.Scene(mesh)
    recs = [dict(r) for r in records]
    plain = [i for i, r in enumerate(records) if not r["light_flag"]][0]
    recs[plain] = rec
    assert sc.set_materials(recs) == 1
    rng = np.random.default_rng(12)
    for uv in np.concatenate([rng.random((200, 3)), rng.uniform(-3, 3, (100, 3))]):
        got = sc.graph_eval(plain, uv)
        want = _walk(mat, uv)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-14), (uv, got[:3], want[:3])
    assert sc.graph_eval(plain + 1 if plain + 1 < len(recs) else plain - 1, [0.5, 0.5, 0]) is None


def test_cpp_host_compiles_noise_graphs_like_the_python_host(cornell_dir, tmp_path):
    """the same materials file through rt_cli --graphs (host logic only, runs without a device) and through scene_io: identical
    programs and tables, noise tables (srand48 / drand48) included"""
    import json
    import subprocess
    mats = json.load(open(os.path.join(cornell_dir, "materials.json")))
    noisy = noise_material("white")
    for n in noisy["nodes"]:
        n.setdefault("name", n["type"]); n.setdefault("input_slots", []); n.setdefault("output_slots", []); n.setdefault("pos", {"x": 0, "y": 0})
    mats = [noisy if m["name"] == "white" else m for m in mats]
    mpath = tmp_path / "materials.json"
    mpath.write_text(json.dumps(mats))
    cli = os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli")
    out = tmp_path / "graphs.bin"
    subprocess.run([cli, "--obj", os.path.join(cornell_dir, "cornell_box_triangulated.obj"), "--materials", str(mpath), "--objmatmap",
                    os.path.join(cornell_dir, "objmatmap.json"), "--graphs", str(out)], capture_output=True, text=True, timeout=120)
    raw = out.read_bytes()
    head = np.frombuffer(raw, dtype=np.uint64, count=5)
    off = 40
    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=int(count), offset=off)
        off += a.nbytes
        return a
    progs, code = take(np.uint32, head[0] * 3).reshape(-1, 3), take(np.uint32, head[1] * 2).reshape(-1, 2)
    spectra, vectors, scalars = take(np.float64, head[2] * 30).reshape(-1, 30), take(np.float64, head[3] * 3).reshape(-1, 3), take(np.float64, head[4])
    from ray_tracer_amd import scenes
    mesh, _, _ = scenes.cornell(8, 8)
    named = scene_io.load_materials(str(mpath))
    omm = scene_io.load_objmatmap(os.path.join(cornell_dir, "objmatmap.json"))
    g = scene_io.compile_material_graphs([named[omm[n]].record() for n in mesh["names"]])
    assert len(progs) >= 1 and np.array_equal(progs, g["programs"]) and np.array_equal(code, g["code"])
    assert np.array_equal(spectra, g["spectra"]) and np.array_equal(vectors, g["vectors"]) and np.array_equal(scalars, g["scalars"])


def _cpp_obj_dump(path, tmp_path):
    import subprocess
    cli = os.path.join(ROOT, "ray_tracer_amd", "lib", "rt_cli")
    out = tmp_path / "obj_dump.txt"
    p = subprocess.run([cli, "--obj", str(path), "--dump-obj", str(out)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    res = {"ok": None, "mtl": [], "v": [], "vt": [], "vn": [], "objects": []}
    for line in out.read_text().split("\n"):
        tok = line.split(" ")
        if tok[0] == "ok":
            res["ok"] = tok[1] == "1"
        elif tok[0] == "mtl":
            res["mtl"].append(line[5:-1])
        elif tok[0] in ("v", "vt", "vn"):
            res[tok[0]].append([float.fromhex(t) for t in tok[1:4]])
        elif tok[0] == "o":
            name, rest = line[3:].split("] [")
            mat, s = rest.rsplit("] ", 1)
            res["objects"].append({"name": name, "material_name": mat, "s": s == "1", "faces": []})
        elif tok[0] == "f":
            res["objects"][-1]["faces"].append([[None if int(i) == -1 else int(i) for i in c.split("/")] for c in tok[1:]])
    return res


def test_cpp_obj_parser_matches_the_python_mirror(cornell_dir, tmp_path):
    """the C++ host's OBJ parser (a byte-level restatement of the reference's stream code, obj.cpp:6-152) against the Python mirror:
    the reference's two Cornell files, and a file with the formats and quirks the stream code accepts"""
    quirks = tmp_path / "quirks.obj"
    quirks.write_text("# comment\nmtllib scene.mtl\n\no first thing\nv 1 2 3\r\nv +1.5 -.25 5.\nv 1e3 1E-3 -1.25e+2\nv 0.1234567890123456789 123456789012345678 1e-23\n"
                      "vn 0 0 1\nvn 0.6 0.8 0 extra words\nvt 0.25 0.75\nvt 1 0 0.5\n\t \nusemtl red\ns off\nf 1//1 2//2 3//1\nf 1/1/1 2/2/2 3/1/2 4/2/1\n"
                      "f 1//1 2//2 3//1 \nf 1 2 3\nf 1/2 2/1 3/2\no second\ns 1\nf -1//-1 2//2 3//1\t\nv 9 8 7")
    for path in (os.path.join(cornell_dir, "cornell_box_triangulated.obj"), os.path.join(cornell_dir, "cornell_box.obj"), quirks):
        py = scene_io.Obj()
        assert py.Load(str(path))
        cpp = _cpp_obj_dump(path, tmp_path)
        assert cpp["ok"]
        assert cpp["mtl"] == py.mtl_file
        for k in ("v", "vt", "vn"):
            assert cpp[k] == [list(map(float, p)) for p in getattr(py, k)], k  # bit for bit: atof on both sides
        assert [o["name"] for o in cpp["objects"]] == py.object_names()
        for a, b in zip(cpp["objects"], py.objects):
            assert (a["material_name"], a["s"]) == (b["material_name"], b["s"])
            assert a["faces"] == b["faces"]
    assert len(cpp["objects"][0]["faces"][2]) == 4  # the trailing blank repeats the last corner (obj.cpp:105-106)


def test_cpp_obj_parser_keeps_the_stream_quirks(tmp_path):
    """lines the reference's stream code reads in ways of its own: a missing word leaves the previous one in place"""
    f = tmp_path / "odd.obj"
    f.write_text("object named\nv 1 2\nv\nvt 7\nusemtl\ns\nf \nf 1/ /2 //\n")
    d = _cpp_obj_dump(f, tmp_path)
    assert d["ok"] and d["objects"][0]["name"] == "named"        # any line starting with 'o' opens the first object
    assert d["v"] == [[1.0, 2.0, 2.0], [0.0, 0.0, 0.0]]          # the missing third number repeats the second; none at all is atof("")
    assert d["vt"] == [[7.0, 7.0, 0.0]]
    assert d["objects"][0]["material_name"] == "usemtl" and d["objects"][0]["s"] is True
    assert d["objects"][0]["faces"][0] == [[None, None, None]]   # "f " re-reads the word f itself: atoi("f") - 1 = -1
    assert d["objects"][0]["faces"][1] == [[0, None, None], [None, 1, None], [None, None, None]]
    bad = tmp_path / "bad.obj"
    bad.write_text("o a\nv 1 2 3\ng group\n")
    assert _cpp_obj_dump(bad, tmp_path)["ok"] is False           # an unknown first word ends the load (obj.cpp:139-142)
    early = tmp_path / "early.obj"
    early.write_text("v 1 2 3\no a\n")
    assert _cpp_obj_dump(early, tmp_path)["ok"] is False         # data before the first object (obj.cpp:25-27)
    indented = tmp_path / "indented.obj"
    indented.write_text("o a\n  # only a '#' in the first column makes a comment\n")
    assert _cpp_obj_dump(indented, tmp_path)["ok"] is False
