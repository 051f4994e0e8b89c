"""Host-side mirror of the reference `Renderer` (renderer.h:21-50): same member names, argument
meaning and error behaviour, with the hot path running on the MI355X through the C ABI."""
import numpy as np

from . import capi
from .camera import LensCamera
from .scene_io import Obj, Ply

Naive, NEE, MIS, NaiveBDPTType, MISBDPTType = range(5)  # enum RenderingAlgorithm, renderer.h:12-18


class Renderer:
    def __init__(self, device=0, seed=None):
        self.ctx = capi.Context(device)
        self.obj_loader = Obj()
        self.cam = LensCamera()
        self.algorithm_type = NaiveBDPTType  # renderer.h:33
        self.spectrum_img = None  # [nx*ny, 30] float64, pixel index i*ny+j (renderer.cpp:116)
        self.preview_img = None   # [nx*ny, 3] float64, gamma-encoded in [0, 1]
        self.preview_img_flag = True
        self.img_updated = False
        self.rendering_runnnig = False  # spelling as in the reference (renderer.h:43)
        self.stop_rendering = False
        self.progress = 0.0
        self.n_lights = 0
        # the reference seeds mt19937 from random_device (renderer.cpp:14-15)
        self.seed = int(np.random.SeedSequence().entropy & 0xFFFFFFFFFFFFFFFF) if seed is None else int(seed)
        # samples per pixel handed to the device at a time; 0: as many as fill one batch of its queues (2^27 paths)
        self.samples_per_pass = 0

    def Load(self, objfilename):  # renderer.cpp:18-22 (the parser's bool is ignored there as well)
        self.obj_loader = Obj()
        self.obj_loader.Load(objfilename)
        self.mesh = self.obj_loader.to_mesh()
        self.ctx.upload_mesh(self.mesh)

    def LoadPly(self, plyfilename, object_name, normals="auto"):
        """Addition (the reference parses PLY files, ply.cpp:13-81, but never renders them, object.cpp:153-233): the
        file becomes one more object called `object_name`, appended to what Load or earlier LoadPly calls brought in.
        False on a parse error."""
        self.ply_loader = Ply()
        if not self.ply_loader.Load(plyfilename):
            print(self.ply_loader.error)
            return False
        self.mesh = self.ply_loader.to_mesh(object_name, normals, into=getattr(self, "mesh", None))
        self.ctx.upload_mesh(self.mesh)
        return True

    def SetEnvironment(self, texels, brightness=0.5):
        """The reference keeps the environment in globals that NaivePathTracing reads (env_mapping_texture, width, height,
        bpp, env_brightness: scene.h:20-22, renderer.cpp:175-193).  texels: uint8 [height, width, 3 or 4], None = none."""
        self.ctx.set_environment(texels, brightness)

    def LoadMesh(self, mesh):
        """Same as Load for a mesh already in memory (synthetic scenes)."""
        self.obj_loader = Obj()
        self.obj_loader.objects = [{"name": n, "faces": []} for n in mesh.get("names", [])]
        self.mesh = mesh
        self.ctx.upload_mesh(mesh)

    def LoadMaterials(self, materials):  # renderer.cpp:24-44
        """materials: {obj object name: NodeMaterial}.  False when an object has no entry or no
        object is a light, as the reference returns."""
        records = []
        for name in self.mesh["names"]:  # obj objects in file order (as the reference), then PLY objects
            m = materials.get(name)
            if m is None:
                print("Material is unset for %s" % name)
                return False
            records.append(m.record())
        try:
            self.ctx.upload_materials(records)
        except capi.BackendError as e:
            if e.code in (capi.ERR_NO_LIGHTS, capi.ERR_NO_MATERIAL):
                return False
            raise
        self.n_lights = self.ctx.stats()["n_lights"]
        return True

    def Clear(self):  # renderer.cpp:47-51
        self.obj_loader.Clear()
        self.img_updated = False

    def RenderImage(self, nx, ny, ns, spectral_samples, enable_openmp=True, print_progress=False):  # renderer.cpp:54-160
        if self.algorithm_type not in (Naive, NEE, MIS, NaiveBDPTType, MISBDPTType):
            raise ValueError("unknown algorithm_type")
        self.rendering_runnnig = True
        self.ctx.set_camera(self.cam)
        self.ctx.render_begin(nx, ny, ns, spectral_samples, self.algorithm_type, self.seed)
        s = 0
        while s < ns:
            if self.stop_rendering:  # checked once per pass, renderer.cpp:70
                break
            # samples_per_pass = 0: the library's choice.  Unwatched, a pass is one full device batch (2^28 paths where the
            # GPU's memory allows, the library halves it where not: tens of milliseconds, so stop_rendering is still looked at many times a second).  With a preview or progress output wanted the
            # reference serves them once per sample (renderer.cpp:69-151); here a watched render reports at least a hundred times.
            fill = max(1, (1 << 28) // max(1, nx * ny))
            if self.samples_per_pass > 0:
                want = self.samples_per_pass
            elif self.preview_img_flag or print_progress:
                want = max(1, min(fill, ns // 100))
            else:
                want = fill
            b = min(want, ns - s)
            self.ctx.render_pass(s, b)  # no sync: the pass's last paths ride along with the next pass; the downloads finish them
            s += b
            if self.preview_img_flag or s == ns:
                self.preview_img = self.ctx.download_preview(s)
                self.img_updated = True
            self.progress = float(s) / ns
            if print_progress:
                print("%g%%" % (self.progress * 100.0))
        self.spectrum_img = self.ctx.download_film()
        self.img_updated = True
        if self.stop_rendering:
            self.stop_rendering = False
        self.rendering_runnnig = False
        if print_progress:
            print("Completed")
