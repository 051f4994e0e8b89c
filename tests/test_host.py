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
    # nodes that are not evaluated on the device are refused, not silently flattened
    tex["nodes"][4] = {"type": "ValueNoise"}
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
    got = np.fromfile(rec, dtype=np.float64).reshape(-1, 62)
    from ray_tracer_amd import scenes
    mesh, records, cam = scenes.cornell(64, 64)
    assert len(got) == len(records) == 7
    for g, r in zip(got, records):
        assert int(g[0]) == r["type"] and bool(g[1]) == r["light_flag"]
        assert np.array_equal(g[2:32], r["a"]) and np.array_equal(g[32:62], r["b"])
