"""ctypes access to the CPU oracle (oracle/librt_oracle.so) and, where it was built, to the partial
reference build under oracle/_ref/.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NBINS = 30
MODE_BVH_REFERENCE, MODE_LIST, MODE_LIST_LOWEST = 0, 1, 2
ALG_NAIVE, ALG_NEE, ALG_MIS, ALG_NAIVE_BDPT, ALG_MIS_BDPT = 0, 1, 2, 3, 4

dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OrcCamera(C.Structure):
    _fields_ = [("film_sensitivity", C.c_double), ("aperture", C.c_double), ("film_width", C.c_double),
                ("film_height", C.c_double), ("n", C.c_double * NBINS), ("d", C.c_double),
                ("lens_maker_coefficient", C.c_double), ("origin", C.c_double * 3),
                ("film_lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("w", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3)]


class OrcMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("light_flag", C.c_int32), ("a", C.c_double * NBINS), ("b", C.c_double * NBINS),
                ("roughness", C.c_double)]


class OrcStats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("samples", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "librt_oracle.so"))
        vp, u64, d = C.c_void_p, C.c_uint64, C.c_double
        L.orc_aabb_hit.argtypes = [vp, vp, vp, vp, d, d]
        L.orc_aabb_surface_area.argtypes = [vp, vp]
        L.orc_aabb_surface_area.restype = d
        L.orc_surrounding_box.argtypes = [vp] * 6
        L.orc_onb_from_w.argtypes = [vp, vp]
        L.orc_onb_local_to_world.argtypes = [vp, vp, vp]
        L.orc_onb_world_to_local.argtypes = [vp, vp, vp]
        L.orc_spectrum_get.argtypes = [vp, d]
        L.orc_spectrum_get.restype = d
        L.orc_spectrum_integrate.argtypes = [vp, d, d]
        L.orc_spectrum_integrate.restype = d
        L.orc_spectrum_add.argtypes = [vp, d, d, d]
        L.orc_xyz.argtypes = [vp, vp]
        L.orc_rgb.argtypes = [vp, vp]
        L.orc_rgb_to_spectrum.argtypes = [vp, vp]
        L.orc_tonemap.argtypes = [vp, d, vp]
        L.orc_rfresnel.argtypes = [d, d]
        L.orc_rfresnel.restype = d
        L.orc_cfresnel.argtypes = [d, d, d]
        L.orc_cfresnel.restype = d
        L.orc_power_heuristic.argtypes = [d, d, d]
        L.orc_power_heuristic.restype = d
        L.orc_cosine_pdf.argtypes = [vp, vp]
        L.orc_cosine_pdf.restype = d
        L.orc_mixture_pick.argtypes = [d, u64]
        L.orc_mixture_pick.restype = u64
        L.orc_mixture_pdf.argtypes = [vp, u64]
        L.orc_mixture_pdf.restype = d
        L.orc_vec3_ops.argtypes = [vp, vp, d, vp]
        L.orc_ray_point.argtypes = [vp, vp, d, vp]
        L.orc_cosine_hemisphere.argtypes = [d, d, vp]
        L.orc_camera_default.argtypes = [C.POINTER(OrcCamera)]
        L.orc_camera_set_lens.argtypes = [C.POINTER(OrcCamera), d, d]
        L.orc_camera_set_camera.argtypes = [C.POINTER(OrcCamera), vp, vp, vp]
        L.orc_camera_get_ray.argtypes = [C.POINTER(OrcCamera), d, d, d, vp, vp, vp, dp, dp, dp]
        L.orc_rng_seed.argtypes = [u64, u64, u64]
        L.orc_rng_seed.restype = u64
        L.orc_rng_next.argtypes = [C.POINTER(u64)]
        L.orc_rng_next.restype = u64
        L.orc_rng_f64.argtypes = [C.POINTER(u64)]
        L.orc_rng_f64.restype = d
        L.orc_scene_create.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp, u64, vp, C.c_uint32]
        L.orc_scene_create.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_n_polygons.argtypes = [vp]
        L.orc_scene_n_polygons.restype = u64
        L.orc_scene_n_triangles.argtypes = [vp]
        L.orc_scene_n_triangles.restype = u64
        L.orc_scene_get_triangles.argtypes = [vp, vp, vp, vp]
        L.orc_scene_polygon_bbox.argtypes = [vp, u64, vp, vp]
        L.orc_polygon_hit.argtypes = [vp, u64, vp, vp, d, d, dp, vp, vp, C.POINTER(C.c_int64)]
        L.orc_scene_build_bvh.argtypes = [vp]
        L.orc_scene_bvh_nodes.argtypes = [vp]
        L.orc_scene_bvh_nodes.restype = u64
        L.orc_scene_bvh_depth.argtypes = [vp]
        L.orc_scene_bvh_depth.restype = C.c_uint32
        L.orc_trace_closest.argtypes = [vp, vp, u64, C.c_int, vp, vp, vp, C.c_int]
        L.orc_trace_closest.restype = u64
        L.orc_trace_any.argtypes = [vp, vp, u64, C.c_int, vp, C.c_int]
        L.orc_trace_any.restype = u64
        L.orc_trace_bvh2.argtypes = [vp, vp, u64, vp, vp, u64, vp, vp, vp, C.POINTER(u64), C.POINTER(u64), C.c_int]
        L.orc_trace_bvh2.restype = u64
        L.orc_trace_wide.argtypes = [vp, vp, u64, vp, vp, u64, vp, vp, vp, C.POINTER(u64), C.POINTER(u64), C.c_int]
        L.orc_trace_wide.restype = u64
        L.orc_scene_set_triangle_test.argtypes = [vp, C.c_int]
        L.orc_scene_set_materials.argtypes = [vp, C.POINTER(OrcMaterial), C.c_uint32]
        L.orc_scene_n_lights.argtypes = [vp]
        L.orc_scene_n_lights.restype = u64
        L.orc_render.argtypes = [vp, C.POINTER(OrcCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64,
                                 vp, C.POINTER(OrcStats), C.c_int]
        L.orc_radiance.argtypes = [vp, C.c_int, vp, vp, d, d, C.POINTER(u64), C.POINTER(OrcStats)]
        L.orc_radiance.restype = d
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def vec(*x):
    return np.array(x, dtype=np.float64)


class Scene:
    """orc_scene over the rt_mesh tables (the dict scene_io.Obj.to_mesh / synth.soup return)."""

    def __init__(self, mesh):
        L = lib()
        self.mesh = mesh
        self._v = np.ascontiguousarray(mesh["v"], dtype=np.float64).reshape(-1, 3)
        self._vt = np.ascontiguousarray(mesh["vt"], dtype=np.float64).reshape(-1, 3)
        self._vn = np.ascontiguousarray(mesh["vn"], dtype=np.float64).reshape(-1, 3)
        self._c = np.ascontiguousarray(mesh["corners"], dtype=np.int64).reshape(-1, 3)
        self._fb = np.ascontiguousarray(mesh["face_begin"], dtype=np.uint64)
        self._ob = np.ascontiguousarray(mesh["object_begin"], dtype=np.uint64)
        self.h = L.orc_scene_create(_p(self._v), len(self._v), _p(self._vt), len(self._vt), _p(self._vn), len(self._vn),
                                    _p(self._c), _p(self._fb), len(self._fb) - 1, _p(self._ob), len(self._ob) - 1)
        if not self.h:
            raise ValueError("orc_scene_create failed")
        self.has_bvh = False

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_free(self.h)
            self.h = None

    @property
    def n_polygons(self):
        return lib().orc_scene_n_polygons(self.h)

    @property
    def n_triangles(self):
        return lib().orc_scene_n_triangles(self.h)

    def build_bvh(self):
        if not self.has_bvh:
            assert lib().orc_scene_build_bvh(self.h) == 1
            self.has_bvh = True
        return lib().orc_scene_bvh_nodes(self.h), lib().orc_scene_bvh_depth(self.h)

    def trace_closest(self, rays, mode=MODE_BVH_REFERENCE, nthreads=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        if mode == MODE_BVH_REFERENCE:
            self.build_bvh()
        n = len(rays)
        poly, tri, t = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        lib().orc_trace_closest(self.h, _p(rays), n, mode, _p(poly), _p(tri), _p(t), nthreads)
        return poly, tri, t

    def trace_any(self, rays, mode=MODE_BVH_REFERENCE, nthreads=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        if mode == MODE_BVH_REFERENCE:
            self.build_bvh()
        out = np.empty(len(rays), np.uint8)
        lib().orc_trace_any(self.h, _p(rays), len(rays), mode, _p(out), nthreads)
        return out

    def trace_bvh2(self, nodes, tri_index, rays, nthreads=0):
        """ordered, pruned traversal over the product's flat BVH2; returns poly, tri, t, node visits, triangle tests"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        nodes = np.ascontiguousarray(nodes)
        tri_index = np.ascontiguousarray(tri_index, dtype=np.uint32)
        n = len(rays)
        poly, tri, t = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        nv, nt = C.c_uint64(), C.c_uint64()
        lib().orc_trace_bvh2(self.h, _p(nodes), len(nodes), _p(tri_index), _p(rays), n, _p(poly), _p(tri), _p(t),
                             C.byref(nv), C.byref(nt), nthreads)
        return poly, tri, t, nv.value, nt.value

    def set_triangle_test(self, watertight):
        """False: the reference's triangle test; True: the optional watertight test (the library's RT_TRIANGLE_WATERTIGHT)"""
        lib().orc_scene_set_triangle_test(self.h, 1 if watertight else 0)

    def trace_wide(self, nodes, tri_index, rays, nthreads=0):
        """ordered, pruned traversal over the product's 8-wide tree; returns poly, tri, t, nodes fetched, triangle records tested"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        nodes = np.ascontiguousarray(nodes)
        assert nodes.dtype.itemsize == 128
        tri_index = np.ascontiguousarray(tri_index, dtype=np.uint32)
        n = len(rays)
        poly, tri, t = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float64)
        nv, nt = C.c_uint64(), C.c_uint64()
        lib().orc_trace_wide(self.h, _p(nodes), len(nodes), _p(tri_index), _p(rays), n, _p(poly), _p(tri), _p(t),
                             C.byref(nv), C.byref(nt), nthreads)
        return poly, tri, t, nv.value, nt.value

    def set_materials(self, records):
        arr = (OrcMaterial * len(records))()
        for i, m in enumerate(records):
            arr[i].type = int(m["type"])
            arr[i].light_flag = int(bool(m["light_flag"]))
            arr[i].roughness = float(m.get("roughness", 0.0))
            for k in range(NBINS):
                arr[i].a[k] = float(m["a"][k])
                arr[i].b[k] = float(m["b"][k])
        rc = lib().orc_scene_set_materials(self.h, arr, len(records))
        self.set_material_graphs(records)
        return rc

    def set_material_graphs(self, records):
        """per-hit graphs of the records that carry one (NodeMaterial.record()["graph"]); clears them otherwise"""
        from ray_tracer_amd import scene_io
        L = lib()
        vp, u32 = C.c_void_p, C.c_uint32
        L.orc_scene_set_material_graphs.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, vp, vp, vp, u32]
        g = scene_io.compile_material_graphs(records)
        if g is None:
            return L.orc_scene_set_material_graphs(self.h, None, 0, None, 0, None, 0, None, 0, None, 0, None, None, None, None, 0)
        tex = [np.ascontiguousarray(t, dtype=np.uint8) for t in g["textures"]]
        n = len(tex)
        ptrs = (C.c_void_p * max(n, 1))(*[t.ctypes.data for t in tex])
        w = (C.c_int * max(n, 1))(*[t.shape[1] for t in tex])
        h = (C.c_int * max(n, 1))(*[t.shape[0] for t in tex])
        bpp = (C.c_int * max(n, 1))(*[t.shape[2] for t in tex])
        return L.orc_scene_set_material_graphs(self.h, _p(g["programs"]), len(g["programs"]), _p(g["code"]), len(g["code"]),
                                               _p(g["spectra"]), len(g["spectra"]), _p(g["vectors"]), len(g["vectors"]),
                                               _p(g["scalars"]), len(g["scalars"]), ptrs, w, h, bpp, n)

    def graph_eval(self, obj, vt):
        """slot a of object `obj` at texture coordinate vt (30 bins), or None when the object has no per-hit program"""
        L = lib()
        L.orc_graph_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        vt = np.ascontiguousarray(vt, dtype=np.float64)
        out = np.zeros(NBINS)
        return out if L.orc_graph_eval(self.h, obj, _p(vt), _p(out)) else None

    def set_environment(self, texels, brightness=0.5):
        L = lib()
        L.orc_scene_set_environment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        if texels is None:
            return L.orc_scene_set_environment(self.h, None, 0, 0, 0, 0.0)
        t = np.ascontiguousarray(texels, dtype=np.uint8)
        return L.orc_scene_set_environment(self.h, _p(t), t.shape[1], t.shape[0], t.shape[2], float(brightness))

    def render(self, cam, nx, ny, ns, spectral_samples, algorithm, seed, first_sample=0, n_samples=None, nthreads=0):
        self.build_bvh()
        film = np.zeros((nx * ny, NBINS), dtype=np.float64)
        st = OrcStats()
        lib().orc_render(self.h, C.byref(cam), nx, ny, ns, first_sample, ns if n_samples is None else n_samples,
                         spectral_samples, algorithm, seed, _p(film), C.byref(st), nthreads)
        return film, {"closest_rays": st.closest_rays, "shadow_rays": st.shadow_rays, "samples": st.samples}


def make_camera(py_cam):
    """OrcCamera from a ray_tracer_amd.camera.LensCamera on which SetLens/set_Camera were called; recomputed
    through the oracle's own SetLens/set_Camera so that both sides are exercised."""
    c = OrcCamera()
    lib().orc_camera_default(C.byref(c))
    c.film_sensitivity, c.aperture = py_cam.film_sensitivity, py_cam.aperture
    c.film_width, c.film_height = py_cam.film_width, py_cam.film_height
    for k in range(NBINS):
        c.n[k] = py_cam.n[k]
    return c


def cornell_setup(cornell_dir, nx, ny, vfov=60.0, focal_length=0.4, obj_name="cornell_box_triangulated.obj"):
    """The benchmark scene: mesh tables, material records per object, python camera, oracle camera."""
    from ray_tracer_amd import scenes
    mesh, records, cam = scenes.cornell(nx, ny, vfov, focal_length, obj_name, cornell_dir)
    oc = make_camera(cam)
    lib().orc_camera_set_lens(C.byref(oc), vfov, focal_length)
    lib().orc_camera_set_camera(C.byref(oc), _p(vec(0, 0, 3)), _p(vec(0, 0, 2)), _p(vec(0, 1, 0)))
    return mesh, records, cam, oc


# ---------------- partial reference build (only where /root/reference was available) ----------------
def ref_core():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_core.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    vp, d = C.c_void_p, C.c_double
    L.ref_aabb_hit.argtypes = [vp, vp, vp, vp, d, d]
    L.ref_aabb_surface_area.argtypes = [vp, vp]
    L.ref_aabb_surface_area.restype = d
    L.ref_surrounding_box.argtypes = [vp] * 6
    L.ref_onb_from_w.argtypes = [vp, vp]
    L.ref_onb_local_to_world.argtypes = [vp, vp, vp]
    L.ref_onb_world_to_local.argtypes = [vp, vp, vp]
    L.ref_spectrum_get.argtypes = [vp, d]
    L.ref_spectrum_get.restype = d
    L.ref_spectrum_integrate.argtypes = [vp, d, d]
    L.ref_spectrum_integrate.restype = d
    L.ref_spectrum_add.argtypes = [vp, d, d, d]
    L.ref_xyz.argtypes = [vp, vp]
    L.ref_rgb.argtypes = [vp, vp]
    L.ref_rgb_to_spectrum.argtypes = [vp, vp]
    L.ref_camera_get_ray.argtypes = [d, d, d, vp, d, d, vp, vp, vp, d, d, d, C.c_long, vp]
    if hasattr(L, "ref_vec3_ops"):  # a library built before these wrappers existed lacks them
        L.ref_vec3_ops.argtypes = [vp, vp, d, vp]
        L.ref_ray_point.argtypes = [vp, vp, d, vp]
    return L


def ref_bvh():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_bvh.so")
    if not os.path.exists(path):
        return None
    lib()  # librt_oracle.so first: libref_bvh links against it
    L = C.CDLL(path)
    vp, u64 = C.c_void_p, C.c_uint64
    L.ref_world_create.argtypes = [vp, vp, C.c_uint32]
    L.ref_world_create.restype = vp
    L.ref_world_free.argtypes = [vp]
    L.ref_world_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_uint32)]
    L.ref_world_trace.argtypes = [vp, vp, u64, C.c_int, vp, vp]
    L.ref_world_trace.restype = u64
    L.ref_world_occluded.argtypes = [vp, vp]
    return L


def ref_ply():
    """the reference's own ply.cpp, compiled unmodified (present only where the reference checkout is)"""
    path = os.path.join(ROOT, "oracle", "_ref", "libref_ply.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    vp, u64 = C.c_void_p, C.c_uint64
    L.ref_ply_load.argtypes = [C.c_char_p]
    L.ref_ply_load.restype = vp
    L.ref_ply_counts.argtypes = [vp] + [C.POINTER(u64)] * 4
    L.ref_ply_tables.argtypes = [vp, vp, vp, vp]
    L.ref_ply_element.argtypes = [vp, u64, C.c_char_p, u64, C.POINTER(u64), C.POINTER(u64)]
    L.ref_ply_free.argtypes = [vp]
    return L


def ref_ply_load(L, path):
    """(elements, vertices [n, 6], face_begin, face_index) as the compiled reference reads the file"""
    h = L.ref_ply_load(path.encode())
    nv, nf, ni, ne = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.ref_ply_counts(h, C.byref(nv), C.byref(nf), C.byref(ni), C.byref(ne))
    vertices = np.zeros((nv.value, 6))
    face_begin = np.zeros(nf.value + 1, dtype=np.uint64)
    face_index = np.zeros(ni.value, dtype=np.int32)
    L.ref_ply_tables(h, _p(vertices), _p(face_begin), _p(face_index))
    elements = []
    for k in range(ne.value):
        name = C.create_string_buffer(256)
        cnt, npr = C.c_uint64(), C.c_uint64()
        L.ref_ply_element(h, k, name, 256, C.byref(cnt), C.byref(npr))
        elements.append({"name": name.value.decode("latin-1"), "num_of_elements": int(cnt.value), "num_of_property": int(npr.value)})
    L.ref_ply_free(h)
    return elements, vertices, face_begin, face_index


def ply_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ply_oracle", os.path.join(ROOT, "oracle", "ply_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_filter():
    """the reference's own filter.cpp, compiled unmodified (present only where the reference checkout is)"""
    path = os.path.join(ROOT, "oracle", "_ref", "libref_filter.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_bilateral.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
    return L


def bilateral(img, window, sigma_d, sigma_r, fn=None):
    """img: [height, width, 4] float64; fn: ref_filter().ref_bilateral, default the oracle's restatement"""
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.zeros_like(img)
    if fn is None:
        fn = lib().orc_bilateral
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
    fn(_p(img), img.shape[1], img.shape[0], int(window), float(sigma_d), float(sigma_r), _p(out))
    return out
