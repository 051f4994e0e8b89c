"""The reference's data formats, host side: OBJ subset (obj.cpp), materials JSON (materialnode.cpp),
object->material map and project JSON (scene.cpp).  Output is the flat tables the C ABI takes."""
import json
import os

import numpy as np

from . import capi

NBINS = 30


# ----------------------------------------------------------------------------------------------
# OBJ (obj.cpp:6-152)
# ----------------------------------------------------------------------------------------------
def _atof(s):
    # atof: longest valid prefix, 0.0 when there is none
    try:
        return float(s)
    except ValueError:
        for end in range(len(s), 0, -1):
            try:
                return float(s[:end])
            except ValueError:
                pass
        return 0.0


def _atoi(s):
    s = s.strip()
    end = 0
    if end < len(s) and s[end] in "+-":
        end += 1
    while end < len(s) and s[end].isdigit():
        end += 1
    try:
        return int(s[:end])
    except ValueError:
        return 0


class Obj:
    """Mirror of the reference `obj` container (obj.h:20-36)."""

    def __init__(self):
        self.v, self.vt, self.vn = [], [], []
        self.objects = []  # dicts: name, material_name, s, faces (list of faces; face = list of [v, vt, vn] with None)
        self.mtl_file = []
        self.ok = True

    def Load(self, filename):
        with open(filename, "r") as f:
            lines = f.read().split("\n")
        cur = None
        for line in lines:
            if line.strip() == "" or line.startswith("#"):
                continue
            tok = line.split()
            if cur is None:
                if line.startswith("mtllib"):
                    self.mtl_file.append(tok[1] if len(tok) > 1 else "")
                    continue
                if not line.startswith("o"):
                    self.ok = False  # obj.cpp:25-27: anything before the first `o` line is an error
                    return False
            key = tok[0]
            if key == "o" or (cur is None and line.startswith("o")):
                cur = {"name": tok[1] if len(tok) > 1 else tok[0], "material_name": "", "s": False, "faces": []}
                self.objects.append(cur)
            elif key == "v":
                self.v.append([_atof(tok[1]), _atof(tok[2]), _atof(tok[3])])
            elif key == "vt":
                self.vt.append([_atof(tok[1]), _atof(tok[2]), 0.0])
            elif key == "vn":
                self.vn.append([_atof(tok[1]), _atof(tok[2]), _atof(tok[3])])
            elif key == "f":
                verts = tok[1:]
                if line[-1:].isspace() and verts:
                    verts = verts + [verts[-1]]  # obj.cpp:105-106: a trailing blank re-reads the last token
                face = []
                for s in verts:
                    parts = s.split("/")
                    info = [None, None, None]
                    for k in range(3):
                        if k < len(parts) and parts[k].strip() != "":
                            info[k] = _atoi(parts[k]) - 1
                    face.append(info)
                cur["faces"].append(face)
            elif key == "usemtl":
                cur["material_name"] = tok[1] if len(tok) > 1 else ""
            elif key == "s":
                cur["s"] = not (len(tok) > 1 and tok[1] == "off")
            elif key == "mtllib":
                self.ok = False
                return False
            else:
                self.ok = False  # obj.cpp:139-142
                return False
        return True

    def Clear(self):
        self.__init__()

    def object_names(self):
        return [o["name"] for o in self.objects]

    def to_mesh(self):
        """Flattens to the rt_mesh tables.  Raises ValueError where the reference would dereference
        an absent vn index (object.cpp:115)."""
        corners, face_begin, object_begin = [], [0], [0]
        for o in self.objects:
            for face in o["faces"]:
                for (iv, ivt, ivn) in face:
                    if iv is None or ivn is None:
                        raise ValueError("face corner without v or vn index in object %s" % o["name"])
                    corners.append([iv, -1 if ivt is None else ivt, ivn])
                face_begin.append(len(corners))
            object_begin.append(len(face_begin) - 1)
        return {
            "v": np.asarray(self.v, dtype=np.float64).reshape(-1, 3),
            "vt": np.asarray(self.vt, dtype=np.float64).reshape(-1, 3),
            "vn": np.asarray(self.vn, dtype=np.float64).reshape(-1, 3),
            "corners": np.asarray(corners, dtype=np.int64).reshape(-1, 3),
            "face_begin": np.asarray(face_begin, dtype=np.uint64),
            "object_begin": np.asarray(object_begin, dtype=np.uint64),
            "names": self.object_names(),
        }


# ----------------------------------------------------------------------------------------------
# material node graphs (materialnode.cpp)
# ----------------------------------------------------------------------------------------------
class UnsupportedMaterialGraph(ValueError):
    """The graph depends on the hit's texture coordinate (or uses a node the reference leaves
    undefined); only graphs that fold to constant spectra run on the device."""


def rgb_to_spectrum(rgb):  # spectrum.cpp:219-232
    s = np.zeros(NBINS)
    s[0:10] = rgb[2]
    s[10:20] = rgb[1]
    s[20:30] = rgb[0]
    return s


class NodeMaterial:
    """A material = node list + connections (materialnode.cpp:1611-1638), folded to one BSDF record.

    `record()` returns the flat dict rt_upload_materials takes."""
    OUTPUT_I = 1  # nodes 0/1 are always UV/Output (materialnode.h:449-450)

    def __init__(self, j):
        self.name = j.get("name", "untitled material")
        self.light_flag = bool(j.get("light_flag", False))
        self.nodes = j["nodes"]
        self.conns = [[] for _ in self.nodes]  # per node, in file order, both directions (materialnode.cpp:1635-1636)
        for c in j.get("connections", []):
            self.conns[c["output_node"]].append(c)
            self.conns[c["input_node"]].append(c)

    def _parent(self, node_i, slot_name):
        # MaterialNode::GetInputParentNode (materialnode.cpp:66-74): first connection touching this node
        # whose input slot name starts with slot_name; returns its output node
        for c in self.conns[node_i]:
            if c["input_slot"][:len(slot_name)] == slot_name:
                return c["output_node"]
        return None

    def _spectrum(self, node_i, depth=0):
        """Compute(global_arg, Spectrum&) into a default-constructed (all zero) Spectrum."""
        if depth > 64:
            raise UnsupportedMaterialGraph("cyclic material graph in %s" % self.name)
        node = self.nodes[node_i]
        t = node["type"]
        data = np.zeros(NBINS)
        if t == "Spectrum":  # :878-881
            return np.asarray(node["data"], dtype=np.float64)
        if t == "RGBtoSpectrum":  # :949-957
            p = self._parent(node_i, "RGB")
            return data if p is None else rgb_to_spectrum(self._vec3(p, depth + 1))
        if t == "Addition":  # :1083-1109
            p = self._parent(node_i, "In0")
            if p is None:
                return data
            data = self._spectrum(p, depth + 1)
            p = self._parent(node_i, "In1")
            return data if p is None else data + self._spectrum(p, depth + 1)
        if t == "Multiplication":  # :1179-1203
            p0, p1 = self._parent(node_i, "In0"), self._parent(node_i, "In1")
            if p0 is None:
                return data
            d1 = self._spectrum(p0, depth + 1)
            if p1 is None:
                return data
            return d1 * self._spectrum(p1, depth + 1)
        if t == "ScalarMultiplication":  # :1145-1166: looks up a slot named "scale" that no node has
            p = self._parent(node_i, "scale")
            return data if p is None else data * self._spectrum(p, depth + 1)
        if t == "RandomSampling":  # :1251-1264 consumes a random number and writes nothing
            return data
        if t in ("UV", "ImageTexture", "Checkerboard", "ValueNoise", "ValueNoise2D", "RodriguesRotation",
                 "AccessVec3Component", "CombineVec3Component", "RGBColor"):
            return data  # these nodes implement no Spectrum Compute: base class leaves the zeros (:78-80)
        return data

    def _vec3(self, node_i, depth=0):
        if depth > 64:
            raise UnsupportedMaterialGraph("cyclic material graph in %s" % self.name)
        node = self.nodes[node_i]
        t = node["type"]
        if t == "RGBColor":  # :931-936
            return np.asarray(node["col"], dtype=np.float64)
        if t in ("UV", "ImageTexture", "Checkerboard", "RodriguesRotation", "CombineVec3Component", "Addition",
                 "Multiplication"):
            raise UnsupportedMaterialGraph("node %s of material %s needs per-hit evaluation" % (t, self.name))
        return np.zeros(3)

    def _slot_spectrum(self, node_i, slot, own_key):
        p = self._parent(node_i, slot)
        if p is None:
            own = np.zeros(NBINS)
            vals = self.nodes[node_i].get(own_key, [])
            own[:len(vals)] = vals
            return own
        return self._spectrum(p)

    def record(self):
        zero = np.zeros(NBINS)
        rec = {"type": capi.MAT_NONE, "light_flag": self.light_flag, "a": zero, "b": zero}
        if len(self.nodes) <= self.OUTPUT_I:
            return rec
        b = self._parent(self.OUTPUT_I, "BSDF")  # materialnode.cpp:1760
        if b is None:
            return rec
        node = self.nodes[b]
        t = node["type"]
        own = lambda key: np.asarray((list(node.get(key, [])) + [0.0] * NBINS)[:NBINS], dtype=np.float64)
        if t == "Lambertian":
            rec.update(type=capi.MAT_LAMBERTIAN, a=self._slot_spectrum(b, "albedo", "albedo"))
        elif t == "DiffuseLight":
            rec.update(type=capi.MAT_DIFFUSE_LIGHT, a=self._slot_spectrum(b, "color", "color"))
        elif t == "Dielectric":
            # "surface_color" never matches the slot named "surface color" (:300,359): always the node's own array
            rec.update(type=capi.MAT_DIELECTRIC, a=own("n"), b=self._slot_spectrum(b, "surface_color", "surface_color"))
        elif t == "Conductor":
            rec.update(type=capi.MAT_CONDUCTOR, a=own("n"), b=own("k"))
        elif t == "ColoredMetal":
            rec.update(type=capi.MAT_COLORED_METAL, a=self._slot_spectrum(b, "albedo", "albedo"))
        elif t == "GGXReflection":
            raise UnsupportedMaterialGraph("GGXReflection reads an uninitialised alpha and returns no value from "
                                           "BSDF()/PDF() in the reference (materialnode.cpp:647-695)")
        else:
            raise UnsupportedMaterialGraph("node type %s wired to the BSDF output of %s" % (t, self.name))
        return rec


def load_materials(path):
    """materials JSON -> {name: NodeMaterial} (Scene::LoadMaterial, scene.cpp:249-265)."""
    with open(path) as f:
        arr = json.load(f)
    out = {}
    for j in arr:
        m = NodeMaterial(j)
        out[m.name] = m
    return out


def load_objmatmap(path):  # scene.cpp:282-300
    with open(path) as f:
        return dict(json.load(f))


def load_project(path):
    """Project JSON (scene.cpp:163-204) -> dict with absolute paths and camera settings."""
    base = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        j = json.load(f)
    rel = lambda p: p if os.path.isabs(p) else os.path.join(base, p)
    return {
        "model": rel(j["model"]), "mat": [rel(m) for m in j.get("mat", [])],
        "objmatmap": rel(j["objmatmap"]) if "objmatmap" in j else None,
        "img_width": j.get("img_width", 500), "img_height": j.get("img_height", 500),
        "img_samples": j.get("img_samples", 100),
        "camera_pos": j.get("camera_pos", [0, 0, 3]), "camera_front": j.get("camera_front", [0, 0, -1]),
        "camera_up": j.get("camera_up", [0, 1, 0]), "vfov": j.get("vfov", 60.0),
    }
