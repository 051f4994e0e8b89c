"""The benchmark scenes as (mesh tables, material records, camera)."""
import os

from . import scene_io
from .camera import LensCamera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL_DIR = os.path.join(ROOT, "tests", "golden", "cornell")


def cornell(nx, ny, vfov=60.0, focal_length=0.4, obj_name="cornell_box_triangulated.obj", directory=CORNELL_DIR):
    """Cornell box of the reference's example/ directory with the camera of example/testproject.json
    (pos (0,0,3), front (0,0,-1), up (0,1,0), vfov 60), focal length 0.4 (scene.h:133) and the GUI's
    film rule film_width = film_height * nx / ny (scene.cpp:668)."""
    o = scene_io.Obj()
    if not o.Load(os.path.join(directory, obj_name)):
        raise ValueError("cannot parse %s" % obj_name)
    mesh = o.to_mesh()
    mats = scene_io.load_materials(os.path.join(directory, "materials.json"))
    omm = scene_io.load_objmatmap(os.path.join(directory, "objmatmap.json"))
    records = [mats[omm[name]].record() for name in mesh["names"]]
    cam = LensCamera()
    cam.film_height = 0.024
    cam.film_width = 0.024 * nx / ny
    cam.SetLens(vfov, focal_length)
    cam.set_Camera([0, 0, 3], [0, 0, 2], [0, 1, 0])  # pos, pos + front, up (scene.cpp:867)
    return mesh, records, cam
