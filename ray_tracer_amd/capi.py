"""ctypes bindings of include/mi355rt.h.  Loading fails loudly when the HIP library is missing."""
import ctypes as C
import os

import numpy as np

NBINS = 30
MISS = 0xFFFFFFFF
OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_NO_LIGHTS, ERR_NO_MATERIAL, ERR_MESH, ERR_HOST = 0, -1, -2, -3, -4, -5, -6, -7
BUILDER_HOST_SAH, BUILDER_DEVICE_LBVH, BUILDER_AUTO = 0, 1, 2
MAT_NONE, MAT_LAMBERTIAN, MAT_DIFFUSE_LIGHT, MAT_DIELECTRIC, MAT_CONDUCTOR, MAT_COLORED_METAL, MAT_GGX = range(7)
ALG_NAIVE, ALG_NEE, ALG_MIS, ALG_NAIVE_BDPT, ALG_MIS_BDPT = 0, 1, 2, 3, 4
STATIC_MAX_RAYS = 1 << 20
TRIANGLE_REFERENCE, TRIANGLE_WATERTIGHT = 0, 1

# MI355RT_LIB names another build of the same library (kernel experiments: tools/bench_trace.py); there is still no fallback
LIB_PATH = os.environ.get("MI355RT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmi355rt.so")

# every symbol include/mi355rt.h declares
EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_upload_mesh", "rt_set_builder", "rt_set_reference_cutting", "rt_get_reference_cutting", "rt_cut_references", "rt_set_triangle_test", "rt_group_set_triangle_test", "rt_set_environment", "rt_upload_materials", "rt_upload_material_graphs", "rt_set_camera", "rt_clear",
    "rt_trace_closest", "rt_trace_closest_static", "rt_trace_any", "rt_trace_closest_device", "rt_trace_any_device", "rt_render_begin",
    "rt_render_pass", "rt_render_sync", "rt_download_film", "rt_download_preview", "rt_film_device_ptr",
    "rt_group_create", "rt_group_destroy", "rt_group_last_error", "rt_group_size", "rt_group_context", "rt_group_upload_mesh",
    "rt_group_upload_materials", "rt_group_upload_material_graphs", "rt_group_set_environment", "rt_group_set_camera", "rt_group_clear",
    "rt_group_render_begin", "rt_group_render_pass", "rt_group_render_sync", "rt_group_download_film", "rt_group_download_preview",
    "rt_get_stats", "rt_get_bvh2", "rt_get_wide_bvh", "rt_filter_bilateral", "rt_filter_bilateral_device", "rt_stream", "rt_profile_start", "rt_profile_stop", "rt_render_capture_rays", "rt_accept_region_box", "rt_group_last_sum_ms", "rt_group_trace_closest",
]

RAY_DTYPE = np.dtype([("ox", "f4"), ("oy", "f4"), ("oz", "f4"), ("tmin", "f4"),
                      ("dx", "f4"), ("dy", "f4"), ("dz", "f4"), ("tmax", "f4")])
HIT_DTYPE = np.dtype([("t", "f4"), ("prim", "u4"), ("tri", "u4"), ("object", "u4")])
BVH2_NODE_DTYPE = np.dtype([("lo0", "f4", 3), ("hi0", "f4", 3), ("lo1", "f4", 3), ("hi1", "f4", 3),
                            ("child0", "i4"), ("child1", "i4"), ("pad", "u4", 2)])
REFERENCE_DTYPE = np.dtype([("tri", "u4"), ("lo", "f4", 3), ("hi", "f4", 3)])
WIDE_NODE_DTYPE = np.dtype([("org", "f4", 3), ("e", "u1", 3), ("imask", "u1"), ("child_base", "u4"), ("tri_base", "u4"), ("meta", "u1", 8),
                            ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8)), ("pad", "u1", 48)])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 16 and BVH2_NODE_DTYPE.itemsize == 64 and WIDE_NODE_DTYPE.itemsize == 128


class Mesh(C.Structure):
    _fields_ = [("v", C.c_void_p), ("n_v", C.c_uint64), ("vt", C.c_void_p), ("n_vt", C.c_uint64),
                ("vn", C.c_void_p), ("n_vn", C.c_uint64), ("corners", C.c_void_p), ("face_begin", C.c_void_p),
                ("n_faces", C.c_uint64), ("object_begin", C.c_void_p), ("n_objects", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("light_flag", C.c_int32), ("a", C.c_double * NBINS), ("b", C.c_double * NBINS),
                ("roughness", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("bpp", C.c_int32)]


class MaterialGraphs(C.Structure):
    _fields_ = [("programs", C.c_void_p), ("n_programs", C.c_uint32), ("code", C.c_void_p), ("n_code", C.c_uint32),
                ("spectra", C.c_void_p), ("n_spectra", C.c_uint32), ("vectors", C.c_void_p), ("n_vectors", C.c_uint32),
                ("scalars", C.c_void_p), ("n_scalars", C.c_uint32), ("textures", C.c_void_p), ("n_textures", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("film_sensitivity", C.c_double), ("aperture", C.c_double), ("film_width", C.c_double),
                ("film_height", C.c_double), ("n", C.c_double * NBINS), ("d", C.c_double),
                ("lens_maker_coefficient", C.c_double), ("origin", C.c_double * 3),
                ("film_lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("w", C.c_double * 3)]


class Stats(C.Structure):
    _fields_ = [("n_faces", C.c_uint64), ("n_triangles", C.c_uint64), ("n_objects", C.c_uint64), ("n_lights", C.c_uint64),
                ("bvh_nodes", C.c_uint64), ("bvh_depth", C.c_uint64), ("bvh_leaves", C.c_uint64),
                ("bvh_build_seconds", C.c_double), ("closest_rays", C.c_uint64), ("any_rays", C.c_uint64),
                ("camera_samples", C.c_uint64), ("device_bytes", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint64), ("items", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("closest", KernelTime), ("any", KernelTime), ("generate", KernelTime), ("shade", KernelTime),
                ("connect", KernelTime)]


class BackendError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mi355rt error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Returns the loaded library; raises when it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C ray_tracer_amd/csrc` (hipcc, gfx950); "
                          "this backend has no CPU path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.rt_create.argtypes = [i32, C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_upload_mesh.argtypes = [vp, C.POINTER(Mesh)]
    lib.rt_set_builder.argtypes = [vp, C.c_int]
    lib.rt_set_reference_cutting.argtypes = [vp, C.c_double]
    lib.rt_get_reference_cutting.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.rt_cut_references.argtypes = [C.POINTER(Mesh), C.c_double, C.c_uint32, vp, C.POINTER(C.c_uint64), vp, C.POINTER(C.c_uint64)]
    lib.rt_set_triangle_test.argtypes = [vp, C.c_int]
    lib.rt_group_set_triangle_test.argtypes = [vp, C.c_int]
    lib.rt_upload_material_graphs.argtypes = [vp, vp]
    lib.rt_filter_bilateral.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    lib.rt_filter_bilateral_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    lib.rt_set_environment.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.rt_upload_materials.argtypes = [vp, C.POINTER(Material), C.c_uint32]
    lib.rt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.rt_clear.argtypes = [vp]
    lib.rt_trace_closest.argtypes = [vp, vp, u64, vp, vp]
    lib.rt_trace_closest_static.argtypes = [vp, vp, u64, vp, vp]
    lib.rt_trace_any.argtypes = [vp, vp, u64, vp]
    lib.rt_trace_closest_device.argtypes = [vp, vp, u64, vp, vp, vp]
    lib.rt_trace_any_device.argtypes = [vp, vp, u64, vp, vp]
    lib.rt_render_begin.argtypes = [vp, i32, i32, i32, i32, i32, u64, vp]
    lib.rt_render_pass.argtypes = [vp, i32, i32]
    lib.rt_render_sync.argtypes = [vp]
    lib.rt_download_film.argtypes = [vp, vp]
    lib.rt_download_preview.argtypes = [vp, i32, vp]
    lib.rt_film_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.rt_get_bvh2.argtypes = [vp, vp, C.POINTER(u64), vp, C.POINTER(u64)]
    lib.rt_get_wide_bvh.argtypes = [vp, vp, C.POINTER(u64), vp, C.POINTER(u64)]
    lib.rt_accept_region_box.argtypes = [vp] * 6
    lib.rt_accept_region_box.restype = None
    lib.rt_stream.argtypes = [vp]
    lib.rt_stream.restype = vp
    lib.rt_profile_start.argtypes = [vp]
    lib.rt_profile_stop.argtypes = [vp, C.POINTER(Profile)]
    lib.rt_render_capture_rays.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.rt_group_create.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
    lib.rt_group_destroy.argtypes = [vp]
    lib.rt_group_destroy.restype = None
    lib.rt_group_last_error.argtypes = [vp]
    lib.rt_group_last_error.restype = C.c_char_p
    lib.rt_group_size.argtypes = [vp]
    lib.rt_group_context.argtypes = [vp, i32]
    lib.rt_group_context.restype = vp
    lib.rt_group_upload_mesh.argtypes = [vp, C.POINTER(Mesh)]
    lib.rt_group_upload_materials.argtypes = [vp, C.POINTER(Material), C.c_uint32]
    lib.rt_group_upload_material_graphs.argtypes = [vp, vp]
    lib.rt_group_set_environment.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.rt_group_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.rt_group_clear.argtypes = [vp]
    lib.rt_group_render_begin.argtypes = [vp, i32, i32, i32, i32, i32, u64]
    lib.rt_group_render_pass.argtypes = [vp, i32, i32]
    lib.rt_group_render_sync.argtypes = [vp]
    lib.rt_group_download_film.argtypes = [vp, vp]
    lib.rt_group_download_preview.argtypes = [vp, i32, vp]
    lib.rt_group_last_sum_ms.argtypes = [vp]
    lib.rt_group_last_sum_ms.restype = C.c_double
    lib.rt_group_trace_closest.argtypes = [vp, vp, u64, vp, vp]
    _lib = lib
    return lib


def cut_references(mesh, budget=0.25, min_tris=4096):
    """what upload_mesh's cutting of long triangles does to `mesh` (host arithmetic only, no device): (first, extra) as
    REFERENCE_DTYPE arrays -- a cut triangle's own reference with the box of its first piece, and its additional references"""
    lib = load()
    v = np.ascontiguousarray(mesh["v"], dtype=np.float64).reshape(-1, 3)
    vt = np.ascontiguousarray(mesh["vt"], dtype=np.float64).reshape(-1, 3)
    vn = np.ascontiguousarray(mesh["vn"], dtype=np.float64).reshape(-1, 3)
    corners = np.ascontiguousarray(mesh["corners"], dtype=np.int64).reshape(-1, 3)
    fb = np.ascontiguousarray(mesh["face_begin"], dtype=np.uint64)
    ob = np.ascontiguousarray(mesh["object_begin"], dtype=np.uint64)
    m = Mesh(_ptr(v), len(v), _ptr(vt), len(vt), _ptr(vn), len(vn), _ptr(corners), _ptr(fb), len(fb) - 1, _ptr(ob), len(ob) - 1)
    nf, ne = C.c_uint64(0), C.c_uint64(0)
    rc = lib.rt_cut_references(C.byref(m), float(budget), int(min_tris), None, C.byref(nf), None, C.byref(ne))
    if rc != 0:
        raise BackendError(rc, "rt_cut_references")
    first, extra = np.zeros(nf.value, dtype=REFERENCE_DTYPE), np.zeros(ne.value, dtype=REFERENCE_DTYPE)
    rc = lib.rt_cut_references(C.byref(m), float(budget), int(min_tris), _ptr(first), C.byref(nf), _ptr(extra), C.byref(ne))
    if rc != 0:
        raise BackendError(rc, "rt_cut_references")
    return first, extra


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One rt_context.  Methods raise BackendError on a non-zero status."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.rt_create(device, C.byref(h))
        if rc != OK:
            raise BackendError(rc, self.lib.rt_last_error(None).decode())
        self.h = h
        self._keep = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise BackendError(rc, self.lib.rt_last_error(self.h).decode())

    def upload_mesh(self, mesh):
        """mesh: dict with v, vt, vn (float64 [n,3]), corners (int64 [nc,3]), face_begin, object_begin (uint64)."""
        v = np.ascontiguousarray(mesh["v"], dtype=np.float64).reshape(-1, 3)
        vt = np.ascontiguousarray(mesh["vt"], dtype=np.float64).reshape(-1, 3)
        vn = np.ascontiguousarray(mesh["vn"], dtype=np.float64).reshape(-1, 3)
        corners = np.ascontiguousarray(mesh["corners"], dtype=np.int64).reshape(-1, 3)
        fb = np.ascontiguousarray(mesh["face_begin"], dtype=np.uint64)
        ob = np.ascontiguousarray(mesh["object_begin"], dtype=np.uint64)
        m = Mesh(_ptr(v), len(v), _ptr(vt), len(vt), _ptr(vn), len(vn), _ptr(corners), _ptr(fb), len(fb) - 1, _ptr(ob), len(ob) - 1)
        self._check(self.lib.rt_upload_mesh(self.h, C.byref(m)))

    def set_reference_cutting(self, budget):
        """additional references the next uploads may give long triangles, as a fraction of the scene's triangles (0: none)"""
        self._check(self.lib.rt_set_reference_cutting(self.h, float(budget)))

    def reference_cutting(self):
        """(triangles cut, references added) by the last upload"""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.rt_get_reference_cutting(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_builder(self, builder):
        """BUILDER_HOST_SAH (default) or BUILDER_DEVICE_LBVH for the next upload_mesh"""
        self._check(self.lib.rt_set_builder(self.h, int(builder)))

    def set_triangle_test(self, mode):
        """TRIANGLE_REFERENCE (default: the reference's own test, not watertight) or TRIANGLE_WATERTIGHT"""
        self._check(self.lib.rt_set_triangle_test(self.h, int(mode)))

    def set_environment(self, texels, brightness=0.5):
        """texels: uint8 [height, width, 3 or 4] (None removes the environment); brightness: env_brightness (scene.cpp:32)"""
        if texels is None:
            self._check(self.lib.rt_set_environment(self.h, None, 0, 0, 0, 0.0))
            return
        t = np.ascontiguousarray(texels, dtype=np.uint8)
        self._check(self.lib.rt_set_environment(self.h, _ptr(t), t.shape[1], t.shape[0], t.shape[2], float(brightness)))

    def filter_bilateral(self, rgba, window, sigma_d, sigma_r):
        """rgba: float64 [height, width, 4]; returns the filtered image (BiliteralFilter::FilterImage, filter.cpp:13-51)"""
        img = np.ascontiguousarray(rgba, dtype=np.float64)
        out = np.empty_like(img)
        self._check(self.lib.rt_filter_bilateral(self.h, _ptr(img), img.shape[1], img.shape[0], int(window), float(sigma_d), float(sigma_r), _ptr(out)))
        return out

    def filter_bilateral_device(self, d_rgba, width, height, window, sigma_d, sigma_r, d_out):
        self._check(self.lib.rt_filter_bilateral_device(self.h, d_rgba, width, height, int(window), float(sigma_d), float(sigma_r), d_out))

    def upload_materials(self, materials):
        """materials: list of dicts {type, light_flag, a, b}, one per obj object."""
        arr = (Material * max(len(materials), 1))()
        for i, m in enumerate(materials):
            arr[i].type = int(m["type"])
            arr[i].light_flag = int(bool(m["light_flag"]))
            arr[i].roughness = float(m.get("roughness", 0.0))
            for k in range(NBINS):
                arr[i].a[k] = float(m["a"][k])
                arr[i].b[k] = float(m["b"][k])
        self._check(self.lib.rt_upload_materials(self.h, arr, len(materials)))
        self.upload_material_graphs(materials)

    def upload_material_graphs(self, materials):
        """attaches the per-hit programs of the records that carry a "graph" (NodeMaterial.record()); none: nothing to do"""
        from . import scene_io
        g = scene_io.compile_material_graphs(materials)
        if g is None:
            return
        tex = [np.ascontiguousarray(t, dtype=np.uint8) for t in g["textures"]]
        tarr = (Texture * max(len(tex), 1))()
        for i, t in enumerate(tex):
            tarr[i].texels = t.ctypes.data
            tarr[i].width, tarr[i].height, tarr[i].bpp = t.shape[1], t.shape[0], t.shape[2]
        progs = np.ascontiguousarray(g["programs"], dtype=np.uint32)
        code = np.ascontiguousarray(g["code"], dtype=np.uint32)
        spectra = np.ascontiguousarray(g["spectra"], dtype=np.float64)
        vectors = np.ascontiguousarray(g["vectors"], dtype=np.float64)
        scalars = np.ascontiguousarray(g["scalars"], dtype=np.float64)
        mg = MaterialGraphs(_ptr(progs), len(progs), _ptr(code), len(code), _ptr(spectra), len(spectra), _ptr(vectors), len(vectors),
                            _ptr(scalars), len(scalars), C.cast(tarr, C.c_void_p), len(tex))
        self._check(self.lib.rt_upload_material_graphs(self.h, C.byref(mg)))

    def set_camera(self, cam):
        c = Camera()
        for name in ("film_sensitivity", "aperture", "film_width", "film_height", "d", "lens_maker_coefficient"):
            setattr(c, name, float(getattr(cam, name)))
        for k in range(NBINS):
            c.n[k] = float(cam.n[k])
        for name in ("origin", "film_lower_left_corner", "horizontal", "vertical", "w"):
            vec = getattr(cam, name)
            for k in range(3):
                getattr(c, name)[k] = float(vec[k])
        self._check(self.lib.rt_set_camera(self.h, C.byref(c)))

    def clear(self):
        self._check(self.lib.rt_clear(self.h))

    def trace_closest(self, rays, want_t64=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        hits = np.zeros(n, dtype=HIT_DTYPE)
        t64 = np.zeros(n, dtype=np.float64) if want_t64 else None
        self._check(self.lib.rt_trace_closest(self.h, _ptr(rays), n, _ptr(hits), _ptr(t64)))
        return (hits, t64) if want_t64 else hits

    def trace_closest_static(self, rays, want_t64=False):
        """rt_trace_closest_static: the same through the library's static staging array (no allocation; at most STATIC_MAX_RAYS rays)"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        hits = np.zeros(n, dtype=HIT_DTYPE)
        t64 = np.zeros(n, dtype=np.float64) if want_t64 else None
        self._check(self.lib.rt_trace_closest_static(self.h, _ptr(rays), n, _ptr(hits), _ptr(t64)))
        return (hits, t64) if want_t64 else hits

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), dtype=np.uint8)
        self._check(self.lib.rt_trace_any(self.h, _ptr(rays), len(rays), _ptr(out)))
        return out

    def trace_closest_device(self, d_rays, n, d_hits, d_t64=None, stream=None):
        self._check(self.lib.rt_trace_closest_device(self.h, d_rays, n, d_hits, d_t64, stream))

    def trace_any_device(self, d_rays, n, d_out, stream=None):
        self._check(self.lib.rt_trace_any_device(self.h, d_rays, n, d_out, stream))

    def render_begin(self, nx, ny, ns, spectral_samples, algorithm, seed=0, d_film=None):
        self._dims = (nx, ny)
        self._check(self.lib.rt_render_begin(self.h, nx, ny, ns, spectral_samples, algorithm, seed, d_film))

    def render_pass(self, first_sample, n_samples):
        self._check(self.lib.rt_render_pass(self.h, first_sample, n_samples))

    def render_sync(self):
        self._check(self.lib.rt_render_sync(self.h))

    def download_film(self):
        nx, ny = self._dims
        out = np.zeros((nx * ny, NBINS), dtype=np.float64)
        self._check(self.lib.rt_download_film(self.h, _ptr(out)))
        return out

    def download_preview(self, samples_done):
        nx, ny = self._dims
        out = np.zeros((nx * ny, 3), dtype=np.float64)
        self._check(self.lib.rt_download_preview(self.h, samples_done, _ptr(out)))
        return out

    def film_device_ptr(self):
        p, b = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rt_film_device_ptr(self.h, C.byref(p), C.byref(b)))
        return p.value, b.value

    def stats(self):
        s = Stats()
        self._check(self.lib.rt_get_stats(self.h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in Stats._fields_}

    def bvh2(self):
        nn, nt = C.c_uint64(), C.c_uint64()
        self._check(self.lib.rt_get_bvh2(self.h, None, C.byref(nn), None, C.byref(nt)))
        nodes = np.zeros(nn.value, dtype=BVH2_NODE_DTYPE)
        tri_index = np.zeros(nt.value, dtype=np.uint32)
        self._check(self.lib.rt_get_bvh2(self.h, _ptr(nodes), C.byref(nn), _ptr(tri_index), C.byref(nt)))
        return nodes, tri_index

    def wide_bvh(self):
        """the 8-wide tree the kernels walk: (nodes as WIDE_NODE_DTYPE records, leaf position -> fan triangle)"""
        nn, nt = C.c_uint64(), C.c_uint64()
        self._check(self.lib.rt_get_wide_bvh(self.h, None, C.byref(nn), None, C.byref(nt)))
        nodes = np.zeros(nn.value, dtype=WIDE_NODE_DTYPE)
        tri_index = np.zeros(nt.value, dtype=np.uint32)
        self._check(self.lib.rt_get_wide_bvh(self.h, _ptr(nodes), C.byref(nn), _ptr(tri_index), C.byref(nt)))
        return nodes, tri_index

    def stream(self):
        return self.lib.rt_stream(self.h)

    def profile_start(self):
        self._check(self.lib.rt_profile_start(self.h))

    def profile_stop(self):
        """{kernel: {ms, launches, items}} from HIP events recorded around every launch since profile_start"""
        p = Profile()
        self._check(self.lib.rt_profile_stop(self.h, C.byref(p)))
        return {k: {"ms": getattr(p, k).ms, "launches": getattr(p, k).launches, "items": getattr(p, k).items}
                for k in ("closest", "any", "generate", "shade", "connect")}

    def capture_rays(self, capacity):
        """starts copying every ray the next render passes trace into a host buffer; returns a function that
        stops the capture and returns the rays"""
        buf = np.zeros((capacity, 8), dtype=np.float32)
        n = C.c_uint64()
        self._check(self.lib.rt_render_capture_rays(self.h, _ptr(buf), capacity, C.byref(n)))

        def finish():
            self._check(self.lib.rt_render_capture_rays(self.h, None, 0, None))
            return buf[:n.value]
        return finish


class _Member(Context):
    """a context owned by a group: usable for per-member calls (stats, profile), never destroyed on its own"""

    def __init__(self, lib, handle):
        self.lib = lib
        self.h = C.c_void_p(handle)
        self._keep = {}

    def close(self):
        self.h = None


class Group:
    """One rt_group: a context per entry of `devices` (ordinals may repeat), the samples of every pass split between them, films summed
    when they are read.  Same method names as Context for what a render needs."""

    def __init__(self, devices):
        self.lib = load()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.rt_group_create(arr, len(devices), C.byref(h))
        if rc != OK:
            raise BackendError(rc, self.lib.rt_group_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise BackendError(rc, self.lib.rt_group_last_error(self.h).decode())

    def size(self):
        return self.lib.rt_group_size(self.h)

    def member(self, i):
        return _Member(self.lib, self.lib.rt_group_context(self.h, i))

    def upload_mesh(self, mesh):
        v = np.ascontiguousarray(mesh["v"], dtype=np.float64).reshape(-1, 3)
        vt = np.ascontiguousarray(mesh["vt"], dtype=np.float64).reshape(-1, 3)
        vn = np.ascontiguousarray(mesh["vn"], dtype=np.float64).reshape(-1, 3)
        corners = np.ascontiguousarray(mesh["corners"], dtype=np.int64).reshape(-1, 3)
        fb = np.ascontiguousarray(mesh["face_begin"], dtype=np.uint64)
        ob = np.ascontiguousarray(mesh["object_begin"], dtype=np.uint64)
        m = Mesh(_ptr(v), len(v), _ptr(vt), len(vt), _ptr(vn), len(vn), _ptr(corners), _ptr(fb), len(fb) - 1, _ptr(ob), len(ob) - 1)
        self._check(self.lib.rt_group_upload_mesh(self.h, C.byref(m)))

    def upload_materials(self, materials):
        arr = (Material * max(len(materials), 1))()
        for i, m in enumerate(materials):
            arr[i].type = int(m["type"])
            arr[i].light_flag = int(bool(m["light_flag"]))
            arr[i].roughness = float(m.get("roughness", 0.0))
            for k in range(NBINS):
                arr[i].a[k] = float(m["a"][k])
                arr[i].b[k] = float(m["b"][k])
        self._check(self.lib.rt_group_upload_materials(self.h, arr, len(materials)))
        for i in range(self.size()):  # per-hit programs go up member by member through the one code path that builds their tables
            self.member(i).upload_material_graphs(materials)

    def set_camera(self, cam):
        c = Camera()
        for name in ("film_sensitivity", "aperture", "film_width", "film_height", "d", "lens_maker_coefficient"):
            setattr(c, name, float(getattr(cam, name)))
        for k in range(NBINS):
            c.n[k] = float(cam.n[k])
        for name in ("origin", "film_lower_left_corner", "horizontal", "vertical", "w"):
            vec = getattr(cam, name)
            for k in range(3):
                getattr(c, name)[k] = float(vec[k])
        self._check(self.lib.rt_group_set_camera(self.h, C.byref(c)))

    def clear(self):
        self._check(self.lib.rt_group_clear(self.h))

    def render_begin(self, nx, ny, ns, spectral_samples, algorithm, seed=0):
        self._dims = (nx, ny)
        self._check(self.lib.rt_group_render_begin(self.h, nx, ny, ns, spectral_samples, algorithm, seed))

    def render_pass(self, first_sample, n_samples):
        self._check(self.lib.rt_group_render_pass(self.h, first_sample, n_samples))

    def render_sync(self):
        self._check(self.lib.rt_group_render_sync(self.h))

    def download_film(self):
        nx, ny = self._dims
        out = np.zeros((nx * ny, NBINS), dtype=np.float64)
        self._check(self.lib.rt_group_download_film(self.h, _ptr(out)))
        return out

    def download_preview(self, samples_done):
        nx, ny = self._dims
        out = np.zeros((nx * ny, 3), dtype=np.float64)
        self._check(self.lib.rt_group_download_preview(self.h, samples_done, _ptr(out)))
        return out

    def last_sum_ms(self):
        """milliseconds the last gather-and-sum of the member films took on the first member's GPU"""
        return float(self.lib.rt_group_last_sum_ms(self.h))

    def trace_closest(self, rays, want_t64=False):
        """member i answers the i-th contiguous range of the rays on its own replica of the scene"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        hits = np.zeros(n, dtype=HIT_DTYPE)
        t64 = np.zeros(n, dtype=np.float64) if want_t64 else None
        self._check(self.lib.rt_group_trace_closest(self.h, _ptr(rays), n, _ptr(hits), _ptr(t64) if want_t64 else None))
        return (hits, t64) if want_t64 else hits

