"""The reference's data formats, host side: OBJ subset (obj.cpp), materials JSON (materialnode.cpp),
object->material map and project JSON (scene.cpp).  Output is the flat tables the C ABI takes."""
import ctypes
import ctypes.util
import json
import os

import numpy as np

from . import capi

NBINS = 30


# ----------------------------------------------------------------------------------------------
# OBJ (obj.cpp:6-152)
# ----------------------------------------------------------------------------------------------
_libc = ctypes.CDLL(ctypes.util.find_library("c"))
_libc.atof.restype = ctypes.c_double
_libc.atof.argtypes = [ctypes.c_char_p]


def _atof(s):
    # the C library's own atof: longest valid prefix (decimal, hex float, inf, nan), 0.0 when there is none
    return _libc.atof(s.encode("latin-1") if isinstance(s, str) else s)


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
# PLY (ply.cpp:13-158) and its bridge to the mesh tables
# ----------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "<i2", "int16": "<i2", "ushort": "<u2",
              "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4", "float": "<f4", "float32": "<f4",
              "double": "<f8", "float64": "<f8"}


def _tok(line):
    return line.split()  # bytes.split(): the whitespace set of `iss >> s` in the C locale


class Ply:
    """Mirror of the reference `ply` container (ply.h:19-35) with flat tables:
    vertices [n, 6] = position and normal (reference vertices[i][0], vertices[i][1]),
    faces as face_index[face_begin[i]:face_begin[i+1]].  ASCII files are read with the reference's token
    rules (a missing token repeats the previous one of its line, atof/atoi prefixes, elements[0] and
    elements[1] taken as vertices and faces whatever their names); binary little-endian files are an addition."""

    def __init__(self):
        self.elements = []  # dicts: name, num_of_elements, num_of_property
        self.vertices = np.zeros((0, 6))
        self.face_index = np.zeros(0, dtype=np.int32)
        self.face_begin = np.zeros(1, dtype=np.uint64)
        self.binary = False
        self.error = ""

    def has_normals(self):
        return bool(self.elements) and self.elements[0]["num_of_property"] >= 6

    def Load(self, filename):
        self.__init__()
        try:
            with open(filename, "rb") as f:
                data = f.read()
        except OSError:
            self.error = "cannot open %s" % filename
            return False
        if not data:
            self.error = "empty file %s" % filename
            return False
        # ---- header.  The std::ifstream calls of ply.cpp with libstdc++'s state rules, which decide what the reference
        # reads from truncated files: an input function on a stream that is not good() sets failbit and does nothing
        # else (a getline that fails this way leaves the caller's line as it was; one that finds the stream good but
        # at its end erases the line)
        st = {"pos": 0, "eof": False, "fail": False}
        n = len(data)

        def getline(line):
            if st["eof"] or st["fail"]:
                st["fail"] = True
                return False, line
            if st["pos"] >= n:
                st["eof"] = st["fail"] = True
                return False, b""
            nl = data.find(b"\n", st["pos"])
            if nl < 0:
                line, st["pos"], st["eof"] = data[st["pos"]:], n, True
            else:
                line, st["pos"] = data[st["pos"]:nl], nl + 1
            return True, line

        def tellg():
            if st["eof"] or st["fail"]:
                st["fail"] = True
                return -1
            return st["pos"]

        def seekg(p):
            st["eof"] = False
            if not st["fail"]:
                st["pos"] = p

        props = []
        fmt = b""
        line = b""
        oldpos = tellg()
        while True:
            ok, line = getline(line)
            if not ok:
                break
            if line.startswith(b"format"):
                fmt = line
            if line.startswith(b"element"):  # LoadElement, ply.cpp:121-158
                seekg(oldpos)
                old2 = tellg()
                ok2, eline = getline(b"")
                t = _tok(eline)
                name = t[1] if len(t) > 1 else (t[0] if t else b"")
                cnt = t[2] if len(t) > 2 else name
                el = {"name": name.decode("latin-1"), "num_of_elements": max(_atoi(cnt.decode("latin-1")), 0), "num_of_property": 0}
                pl = []
                old2 = tellg()
                while True:
                    ok2, eline = getline(eline)
                    if not ok2:
                        break
                    t = _tok(eline)
                    if not t or t[0] != b"property":
                        break
                    el["num_of_property"] += 1
                    pl.append([x.decode("latin-1") for x in t[1:]])
                    old2 = tellg()
                seekg(old2)
                self.elements.append(el)
                props.append(pl)
            if line.startswith(b"end_header"):
                break
            oldpos = tellg()
        pos = st["pos"]
        if len(self.elements) < 2:
            self.error = "ply header declares fewer than two elements"
            return False
        self.binary = b"binary" in fmt
        if b"binary_big_endian" in fmt:
            self.error = "binary_big_endian ply files are not supported"
            return False
        nv, nf = self.elements[0]["num_of_elements"], self.elements[1]["num_of_elements"]
        if self.binary:
            return self._load_binary(data, pos, props, nv, nf)
        if st["eof"] or st["fail"]:
            body, filler = [], line  # every further getline fails and leaves the last header line in place
        else:
            body = data[pos:].split(b"\n")
            if body[-1] == b"":
                body, filler = body[:-1], b""  # the first getline past the end erases the line, the others leave it empty
            else:
                filler = body[-1]  # no newline after the last line: further getlines fail without touching the line
        if len(body) < nv + nf:
            body = body + [filler] * (nv + nf - len(body))
        self.vertices = self._ascii_vertices(body[:nv])
        ok = self._ascii_faces(body[nv:nv + nf])
        return ok

    @staticmethod
    def _ascii_vertices(lines):
        nv = len(lines)
        out = np.zeros((nv, 6))
        if nv == 0:
            return out
        counts = np.fromiter((len(_tok(l)) for l in lines), dtype=np.int64, count=nv)
        c = int(counts[0])
        if c > 0 and bool((counts == c).all()):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    arr = np.fromstring(b" ".join(lines).decode("latin-1"), dtype=np.float64, sep=" ")
                except (ValueError, DeprecationWarning):
                    arr = np.zeros(0)
            if arr.size == nv * c:
                arr = arr.reshape(nv, c)
                k = min(c, 6)
                out[:, :k] = arr[:, :k]
                out[:, k:] = arr[:, k - 1:k]  # a missing token repeats the last one read
                return out
        for i, l in enumerate(lines):  # exact token rules, line by line
            s = ""
            t = _tok(l)
            for j in range(6):
                if j < len(t):
                    s = t[j].decode("latin-1")
                out[i, j] = _atof(s) if s else 0.0
        return out

    def _ascii_faces(self, lines):
        nf = len(lines)
        if nf == 0:
            return True
        counts = np.fromiter((len(_tok(l)) for l in lines), dtype=np.int64, count=nf)
        c = int(counts[0])
        if c > 1 and bool((counts == c).all()):
            try:
                arr = np.array(b" ".join(lines).split(), dtype=np.int64)
            except (ValueError, OverflowError):
                arr = np.zeros(0, dtype=np.int64)
            if arr.size == nf * c:
                arr = arr.reshape(nf, c)
                if bool((arr[:, 0] == c - 1).all()):
                    self.face_index = arr[:, 1:].astype(np.int32).reshape(-1)
                    self.face_begin = (np.arange(nf + 1, dtype=np.uint64) * np.uint64(c - 1))
                    return True
        idx, begin = [], [0]
        for i, l in enumerate(lines):
            t = _tok(l)
            s = t[0].decode("latin-1") if t else ""
            cnt = _atoi(s) if s else 0
            if cnt < 0:
                self.error = "negative corner count in face %d" % i
                return False
            for j in range(cnt):
                if 1 + j < len(t):
                    s = t[1 + j].decode("latin-1")
                idx.append(_atoi(s) if s else 0)
            begin.append(len(idx))
        self.face_index = np.asarray(idx, dtype=np.int32)
        self.face_begin = np.asarray(begin, dtype=np.uint64)
        return True

    def _load_binary(self, data, pos, props, nv, nf):
        vt = []
        for k, pr in enumerate(props[0]):
            if pr[0] == "list" or pr[0] not in _PLY_TYPES:
                self.error = "unsupported vertex property in binary ply"
                return False
            vt.append(("p%d" % k, _PLY_TYPES[pr[0]]))
        vdt = np.dtype(vt)
        if pos + nv * vdt.itemsize > len(data):
            self.error = "binary ply file ends before its declared elements do"
            return False
        rec = np.frombuffer(data, dtype=vdt, count=nv, offset=pos)
        self.vertices = np.zeros((nv, 6))
        for k in range(min(6, len(vt))):
            self.vertices[:, k] = rec["p%d" % k]
        pos += nv * vdt.itemsize
        fp = props[1]
        if not fp or fp[0][0] != "list" or any(p[0] == "list" for p in fp[1:]) or any(p[0] not in _PLY_TYPES for p in fp[1:]):
            self.error = "unsupported face properties in binary ply"
            return False
        ct, it = np.dtype(_PLY_TYPES[fp[0][1]]), np.dtype(_PLY_TYPES[fp[0][2]])
        tail = sum(np.dtype(_PLY_TYPES[p[0]]).itemsize for p in fp[1:])
        idx, begin = [], [0]
        total = 0
        for _ in range(nf):
            if pos + ct.itemsize > len(data):
                self.error = "binary ply file ends before its declared elements do"
                return False
            c = int(np.frombuffer(data, dtype=ct, count=1, offset=pos)[0])
            pos += ct.itemsize
            if pos + c * it.itemsize + tail > len(data):
                self.error = "binary ply file ends before its declared elements do"
                return False
            idx.append(np.frombuffer(data, dtype=it, count=c, offset=pos).astype(np.int32))
            pos += c * it.itemsize + tail
            total += c
            begin.append(total)
        self.face_index = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int32)
        self.face_begin = np.asarray(begin, dtype=np.uint64)
        return True

    # ---- bridge: one object, one polygon per face (rules of ObjModel, object.cpp:94-140; the reference's own
    # PlyModel is commented out, object.cpp:153-233)
    def to_mesh(self, name="ply", normals="auto", into=None):
        """normals: "file" (columns 3..5), "flat" (one per face: (v1-v0)x(v2-v1)), "smooth" (per vertex, sum of the
        area-weighted face normals), "auto" = "file" when the file has them, else "smooth".  Faces with fewer than
        three corners or indices outside the vertex table are dropped (count in mesh["dropped_faces"]).
        `into`: a mesh to append this object to."""
        if normals == "auto":
            normals = "file" if self.has_normals() else "smooth"
        nv = len(self.vertices)
        fb = self.face_begin.astype(np.int64)
        fi = self.face_index.astype(np.int64)
        sizes = np.diff(fb)
        bad_corner = (fi < 0) | (fi >= nv)
        bad = np.add.reduceat(bad_corner, fb[:-1][sizes > 0]) > 0 if len(fi) else np.zeros(0, dtype=bool)
        valid = sizes >= 3
        if len(fi):
            tmp = np.zeros(len(sizes), dtype=bool)
            tmp[sizes > 0] = bad
            valid &= ~tmp
        keep_corner = np.repeat(valid, sizes)
        kfi = fi[keep_corner]
        ksizes = sizes[valid]
        kbegin = np.concatenate([[0], np.cumsum(ksizes)])
        pos = self.vertices[:, 0:3]
        a, b, c = pos[kfi[kbegin[:-1]]], pos[kfi[kbegin[:-1] + 1]], pos[kfi[kbegin[:-1] + 2]]
        e0, e1 = b - a, c - b
        fn = np.stack([e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1], -(e0[:, 0] * e1[:, 2] - e0[:, 2] * e1[:, 0]),
                       e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]], 1)
        if normals == "file":
            vn, corner_n = self.vertices[:, 3:6].copy(), kfi
        elif normals == "flat":
            vn, corner_n = fn, np.repeat(np.arange(len(ksizes), dtype=np.int64), ksizes)
        elif normals == "smooth":
            w = np.repeat(fn, ksizes, axis=0)
            vn = np.stack([np.bincount(kfi, weights=w[:, k], minlength=nv) for k in range(3)], 1)
            zero = (vn == 0).all(1)
            vn[zero] = [0.0, 0.0, 1.0]
            corner_n = kfi
        else:
            raise ValueError("normals must be auto, file, flat or smooth")
        v0 = 0 if into is None else len(into["v"])
        n0 = 0 if into is None else len(into["vn"])
        corners = np.stack([kfi + v0, np.full(len(kfi), -1, dtype=np.int64), corner_n + n0], 1)
        mesh = {"v": pos.copy(), "vt": np.zeros((0, 3)), "vn": vn, "corners": corners, "face_begin": kbegin.astype(np.uint64),
                "object_begin": np.array([0, len(ksizes)], dtype=np.uint64), "names": [name], "dropped_faces": int((~valid).sum())}
        if into is None:
            return mesh
        nf0, nc0 = len(into["face_begin"]) - 1, len(into["corners"])
        return {"v": np.concatenate([into["v"], mesh["v"]]), "vt": into["vt"], "vn": np.concatenate([into["vn"], vn]),
                "corners": np.concatenate([into["corners"], corners]),
                "face_begin": np.concatenate([into["face_begin"], (kbegin[1:] + nc0).astype(np.uint64)]),
                "object_begin": np.concatenate([into["object_begin"], np.array([nf0 + len(ksizes)], dtype=np.uint64)]),
                "names": list(into["names"]) + [name], "dropped_faces": into.get("dropped_faces", 0) + mesh["dropped_faces"]}


def write_ply(path, vertices, faces, normals=None, binary=False):
    """Writes an ASCII (or binary little-endian) PLY: vertices [n, 3], optional normals [n, 3], faces = list of index lists
    or an [m, k] array.  Used for synthetic test and benchmark scenes."""
    vertices = np.asarray(vertices, dtype=np.float64)
    uniform = isinstance(faces, np.ndarray) and faces.ndim == 2
    nfaces = len(faces)
    head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "element vertex %d" % len(vertices)]
    head += ["property %s %s" % ("double" if binary else "float", c) for c in ("x", "y", "z")]
    if normals is not None:
        head += ["property %s %s" % ("double" if binary else "float", c) for c in ("nx", "ny", "nz")]
    head += ["element face %d" % nfaces, "property list uchar int vertex_indices", "end_header"]
    cols = vertices if normals is None else np.concatenate([vertices, np.asarray(normals, dtype=np.float64)], 1)
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if binary:
            f.write(cols.astype("<f8").tobytes())
            if uniform:
                rec = np.zeros(nfaces, dtype=[("c", "u1"), ("i", "<i4", (faces.shape[1],))])
                rec["c"] = faces.shape[1]
                rec["i"] = faces
                f.write(rec.tobytes())
            else:
                for face in faces:
                    f.write(bytes([len(face)]) + np.asarray(face, dtype="<i4").tobytes())
        else:
            np.savetxt(f, cols, fmt="%.17g")
            if uniform:
                np.savetxt(f, np.concatenate([np.full((nfaces, 1), faces.shape[1], dtype=np.int64), faces], 1), fmt="%d")
            else:
                for face in faces:
                    f.write((" ".join([str(len(face))] + [str(int(i)) for i in face]) + "\n").encode())


# ----------------------------------------------------------------------------------------------
# material node graphs (materialnode.cpp)
# ----------------------------------------------------------------------------------------------
_ggx_warned = False


def _warn_ggx_once(name):
    """a scene that wires GGXReflection renders the backend's defined variant, not numbers of the reference (which has none for it:
    its node is undefined behaviour): say so once per process instead of silently"""
    global _ggx_warned
    if not _ggx_warned:
        _ggx_warned = True
        import sys
        print("mi355rt: material %s uses GGXReflection, which is undefined in the reference (uninitialised alpha, BSDF()/PDF() return nothing); "
              "rendering the backend's defined variant RT_MAT_GGX with alpha from the node's \"roughness\" key (0.5 without one)" % name, file=sys.stderr)


class UnsupportedMaterialGraph(ValueError):
    """The graph cannot be turned into a material record: a node the reference leaves undefined (GGX), a cycle, a texture
    file in a format that is not decoded here."""


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

    def __init__(self, j, textures=None):
        self.textures = dict(textures or {})  # path -> uint8 [h, w, bpp] for ImageTexture nodes
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

    # ---- per-hit graphs: the same walks, but a sub-graph that reads the hit's texture coordinate becomes an expression
    # ("S", spectrum) | ("V", vec3) constants; ("UV",); ("IMG", texels, uv); ("CHK", size, uv); ("R2S", vec);
    # ("ADDS" | "MULS", a, b) on spectra; ("ADDV" | "MULV", a, b) on vectors; scalars: ("D", x) constants, ("ACC", i, vec),
    # ("N1", size, x), ("N2", u_size, v_size, u, v), ("ADDD" | "MULD", a, b); vectors from scalars: ("CMB", a, b, c), ("ROT", v, n, theta)
    def _texture(self, path):
        """texels of an ImageTexture node as uint8 [h, w, bpp], or None when there is no readable file (the reference then leaves
        its output untouched, materialnode.cpp:1003-1004).  The reference decodes with stb_image; here: a dict handed to the
        constructor, .npy files and binary PPM (P6)."""
        if path in self.textures:
            return np.ascontiguousarray(self.textures[path], dtype=np.uint8)
        if not path or not os.path.exists(path):
            return None
        if path.endswith(".npy"):
            return np.ascontiguousarray(np.load(path), dtype=np.uint8)
        with open(path, "rb") as f:
            data = f.read()
        if data[:2] != b"P6":
            raise UnsupportedMaterialGraph("texture %s: only .npy and binary PPM files can be decoded here" % path)
        tok, pos = [], 2
        while len(tok) < 3:
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            tok.append(int(data[pos:end]))
            pos = end
        w, h, _ = tok
        return np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos + 1).reshape(h, w, 3).copy()

    def _vec_expr(self, node_i, depth=0):
        if depth > 64:
            raise UnsupportedMaterialGraph("cyclic material graph in %s" % self.name)
        node = self.nodes[node_i]
        t = node["type"]
        zero = ("V", np.zeros(3))
        if t == "RGBColor":  # :931-936
            return ("V", np.asarray(node["col"], dtype=np.float64))
        if t == "UV":  # :893-896
            return ("UV",)
        if t == "ImageTexture":  # :1001-1018
            tex = self._texture(node.get("path", ""))
            p = self._parent(node_i, "UV")
            if tex is None or p is None:
                return zero
            return ("IMG", tex, self._vec_expr(p, depth + 1))
        if t == "Checkerboard":  # :1046-1063
            p = self._parent(node_i, "UV")
            if p is None:
                return zero
            return ("CHK", float(node["size"]), self._vec_expr(p, depth + 1))
        if t in ("Addition", "Multiplication"):  # the dvec3 overloads, :1098-1102 and :1195-1198
            p0, p1 = self._parent(node_i, "In0"), self._parent(node_i, "In1")
            if p0 is None:
                return zero
            a = self._vec_expr(p0, depth + 1)
            if p1 is None:
                return a if t == "Addition" else zero
            b = self._vec_expr(p1, depth + 1)
            if a[0] == "V" and b[0] == "V":
                return ("V", a[1] + b[1] if t == "Addition" else a[1] * b[1])
            return ("ADDV" if t == "Addition" else "MULV", a, b)
        if t == "CombineVec3Component":  # :1333-1347: stops at the first unconnected input, the rest stays as it was (zero)
            parts = []
            for slot in ("0", "1", "2"):
                p = self._parent(node_i, slot)
                if p is None:
                    break
                parts.append(self._scalar_expr(p, depth + 1))
            if not parts:
                return zero
            parts += [("D", 0.0)] * (3 - len(parts))
            if all(e[0] == "D" for e in parts):
                return ("V", np.array([e[1] for e in parts]))
            return ("CMB",) + tuple(parts)
        if t == "RodriguesRotation":  # :1536-1573
            p = self._parent(node_i, "n")
            n = ("V", np.asarray(node.get("n", [0.0, 0.0, 1.0]), dtype=np.float64)) if p is None else self._vec_expr(p, depth + 1)
            p = self._parent(node_i, "theta")
            theta = ("D", float(node.get("theta", 0.0))) if p is None else self._scalar_expr(p, depth + 1)
            p = self._parent(node_i, "v")
            if p is None:
                return zero  # returns before writing
            return ("ROT", self._vec_expr(p, depth + 1), n, theta)
        return zero  # no dvec3 Compute: the base class leaves the caller's vector untouched (:81-83)

    @staticmethod
    def noise_table(count, seed=323048):
        """ValueNoiseNode::GenerateRand / ValueNoise2DNode::GenerateRand (:1412-1419, :1500-1508): srand48(seed), then drand48()
        per entry (the default seed of materialnode.h:395,408; nothing in a materials file changes it)"""
        import ctypes
        libc = ctypes.CDLL(None)
        libc.drand48.restype = ctypes.c_double
        libc.srand48(ctypes.c_long(seed))
        return np.array([libc.drand48() for _ in range(count)])

    def _scalar_expr(self, node_i, depth=0):
        """Compute(global_arg, double&).  A node without that overload leaves the caller's variable untouched, which the
        reference's callers never initialise (undefined there); here it reads as 0."""
        if depth > 64:
            raise UnsupportedMaterialGraph("cyclic material graph in %s" % self.name)
        node = self.nodes[node_i]
        t = node["type"]
        zero = ("D", 0.0)
        if t == "AccessVec3Component":  # :1293-1301; the index is not stored in the file and stays 0 (materialnode.h:375)
            p = self._parent(node_i, "vector")
            if p is None:
                return zero
            v = self._vec_expr(p, depth + 1)
            return ("D", float(v[1][0])) if v[0] == "V" else ("ACC", 0, v)
        if t == "ValueNoise":  # :1366-1384
            p = self._parent(node_i, "input")
            if p is None:
                return zero
            return ("N1", int(node["u_size"]), self._scalar_expr(p, depth + 1))
        if t == "ValueNoise2D":  # :1447-1478
            pu, pv = self._parent(node_i, "u"), self._parent(node_i, "v")
            if pu is None or pv is None:
                return zero
            return ("N2", int(node["u_size"]), int(node["v_size"]), self._scalar_expr(pu, depth + 1), self._scalar_expr(pv, depth + 1))
        if t in ("Addition", "Multiplication"):  # the double overloads, :1095-1099 and :1192-1195
            p0, p1 = self._parent(node_i, "In0"), self._parent(node_i, "In1")
            if p0 is None:
                return zero
            a = self._scalar_expr(p0, depth + 1)
            if p1 is None:
                return a if t == "Addition" else zero
            b = self._scalar_expr(p1, depth + 1)
            if a[0] == "D" and b[0] == "D":
                return ("D", a[1] + b[1] if t == "Addition" else a[1] * b[1])
            return ("ADDD" if t == "Addition" else "MULD", a, b)
        return zero  # ScalarMultiplication looks up a slot no node has (:1145-1158); RandomSampling writes nothing (:1223-1236)

    def _spec_expr(self, node_i, depth=0):
        if depth > 64:
            raise UnsupportedMaterialGraph("cyclic material graph in %s" % self.name)
        node = self.nodes[node_i]
        t = node["type"]
        zero = ("S", np.zeros(NBINS))
        if t == "RGBtoSpectrum":  # :949-957
            p = self._parent(node_i, "RGB")
            if p is None:
                return zero
            v = self._vec_expr(p, depth + 1)
            return ("S", rgb_to_spectrum(v[1])) if v[0] == "V" else ("R2S", v)
        if t in ("Addition", "Multiplication"):
            p0, p1 = self._parent(node_i, "In0"), self._parent(node_i, "In1")
            if p0 is None:
                return zero
            a = self._spec_expr(p0, depth + 1)
            if p1 is None:
                return a if t == "Addition" else zero
            b = self._spec_expr(p1, depth + 1)
            if a[0] == "S" and b[0] == "S":
                return ("S", a[1] + b[1] if t == "Addition" else a[1] * b[1])
            return ("ADDS" if t == "Addition" else "MULS", a, b)
        return ("S", self._spectrum(node_i, depth))  # every other node: the constant walk

    def _slot_expr(self, node_i, slot, own_key):
        p = self._parent(node_i, slot)
        if p is None:
            return ("S", self._slot_spectrum(node_i, slot, own_key))
        return self._spec_expr(p)

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
        def slot_a(slot):
            e = self._slot_expr(b, slot, slot)
            if e[0] == "S":
                return {"a": e[1]}
            return {"a": zero, "graph": e}  # evaluated per hit on the device (rt_upload_material_graphs)
        if t == "Lambertian":
            rec.update(type=capi.MAT_LAMBERTIAN, **slot_a("albedo"))
        elif t == "DiffuseLight":
            rec.update(type=capi.MAT_DIFFUSE_LIGHT, **slot_a("color"))
        elif t == "Dielectric":
            # "surface_color" never matches the slot named "surface color" (:300,359): always the node's own array
            rec.update(type=capi.MAT_DIELECTRIC, a=own("n"), b=self._slot_spectrum(b, "surface_color", "surface_color"))
        elif t == "Conductor":
            rec.update(type=capi.MAT_CONDUCTOR, a=own("n"), b=own("k"))
        elif t == "ColoredMetal":
            rec.update(type=capi.MAT_COLORED_METAL, **slot_a("albedo"))
        elif t == "GGXReflection":
            # the reference reads an uninitialised alpha and returns no value from BSDF()/PDF() (materialnode.cpp:647-695): the
            # backend runs a well-defined variant (RT_MAT_GGX) with alpha from the node's "roughness" key, 0.5 without one
            rec.update(type=capi.MAT_GGX, a=own("n"), b=own("k"), roughness=float(node.get("roughness", 0.5)))
            _warn_ggx_once(self.name)
        else:
            raise UnsupportedMaterialGraph("node type %s wired to the BSDF output of %s" % (t, self.name))
        return rec


(OP_SPEC, OP_VEC, OP_UV, OP_ZERO_S, OP_ZERO_V, OP_IMAGE, OP_CHECKER, OP_RGB2SPEC, OP_ADD_S, OP_MUL_S, OP_ADD_V, OP_MUL_V,
 OP_ACCESS, OP_COMBINE, OP_NOISE1, OP_NOISE2, OP_ROTATE, OP_SCALAR) = range(18)


def compile_material_graphs(records):
    """records (one per object, as NodeMaterial.record() returns them) -> the tables of rt_upload_material_graphs, or None when
    no record carries a per-hit graph: programs [n, 3] (object, first pair, pairs), code [m, 2] (op, arg), spectra [k, 30],
    vectors [k, 3], scalars [k], textures (list of uint8 [h, w, bpp])."""
    code, spectra, vectors, scalars, textures, programs = [], [], [], [], [], []

    def emit(e):
        k = e[0]
        if k == "S":
            spectra.append(np.asarray(e[1], dtype=np.float64)); code.append((OP_SPEC, len(spectra) - 1))
        elif k == "V":
            vectors.append(np.asarray(e[1], dtype=np.float64)); code.append((OP_VEC, len(vectors) - 1))
        elif k == "UV":
            code.append((OP_UV, 0))
        elif k == "IMG":
            emit(e[2])
            for i, t in enumerate(textures):
                if t is e[1]:
                    break
            else:
                textures.append(e[1]); i = len(textures) - 1
            code.append((OP_IMAGE, i))
        elif k == "CHK":
            emit(e[2]); scalars.append(e[1]); code.append((OP_CHECKER, len(scalars) - 1))
        elif k == "R2S":
            emit(e[1]); code.append((OP_RGB2SPEC, 0))
        elif k == "D":
            scalars.append(float(e[1])); code.append((OP_SCALAR, len(scalars) - 1))
        elif k == "ACC":
            emit(e[2]); code.append((OP_ACCESS, e[1]))
        elif k == "CMB":
            emit(e[1]); emit(e[2]); emit(e[3]); code.append((OP_COMBINE, 0))
        elif k == "N1":
            emit(e[2])
            code.append((OP_NOISE1, len(scalars)))
            scalars.extend([float(e[1])] + list(NodeMaterial.noise_table(e[1] + 1)))
        elif k == "N2":
            emit(e[3]); emit(e[4])
            code.append((OP_NOISE2, len(scalars)))
            scalars.extend([float(e[1]), float(e[2])] + list(NodeMaterial.noise_table((e[1] + 1) * (e[2] + 1))))
        elif k == "ROT":
            emit(e[1]); emit(e[2]); emit(e[3]); code.append((OP_ROTATE, 0))
        else:
            emit(e[1]); emit(e[2])
            code.append(({"ADDS": OP_ADD_S, "MULS": OP_MUL_S, "ADDV": OP_ADD_V, "MULV": OP_MUL_V, "ADDD": OP_ADD_S, "MULD": OP_MUL_S}[k], 0))

    for obj, r in enumerate(records):
        if r.get("graph") is None:
            continue
        begin = len(code)
        emit(r["graph"])
        programs.append((obj, begin, len(code) - begin))
    if not programs:
        return None
    return {"programs": np.asarray(programs, dtype=np.uint32).reshape(-1, 3), "code": np.asarray(code, dtype=np.uint32).reshape(-1, 2),
            "spectra": np.asarray(spectra, dtype=np.float64).reshape(-1, NBINS), "vectors": np.asarray(vectors, dtype=np.float64).reshape(-1, 3),
            "scalars": np.asarray(scalars, dtype=np.float64).reshape(-1), "textures": textures}


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
