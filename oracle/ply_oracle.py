"""TEST INFRASTRUCTURE ONLY -- plain restatement of the reference's PLY reader, one statement per statement of
ply.cpp, used to check the product's parsers (ray_tracer_amd/host/ply.cpp and ray_tracer_amd/scene_io.Ply).
Pinned: tests/test_ply.py compares it bit for bit with tables recorded from the reference's own ply.cpp compiled
unmodified (oracle/_ref/libref_ply.so, fixtures in tests/golden/ply_vectors.npz).  Nothing in the product imports this."""
import ctypes
import ctypes.util

_libc = ctypes.CDLL(ctypes.util.find_library("c"))
_libc.atof.restype = ctypes.c_double
_libc.atof.argtypes = [ctypes.c_char_p]
_libc.atoi.restype = ctypes.c_int
_libc.atoi.argtypes = [ctypes.c_char_p]

_SPACE = b" \t\n\v\f\r"  # std::isspace in the C locale: what `iss >> s` skips and stops at


class _Stream:
    """the std::ifstream calls ply.cpp makes (getline, tellg, seekg) with the state rules of libstdc++: an input
    function on a stream that is not good() sets failbit and does nothing else -- in particular a getline that fails
    this way leaves the caller's string as it was, while one that finds the stream good but empty erases it"""

    def __init__(self, data):
        self.data, self.pos, self.eof, self.fail = data, 0, False, False

    def getline(self, line):
        if self.eof or self.fail:
            self.fail = True
            return False, line
        if self.pos >= len(self.data):
            self.eof = self.fail = True
            return False, b""
        nl = self.data.find(b"\n", self.pos)
        if nl < 0:
            line, self.pos, self.eof = self.data[self.pos:], len(self.data), True
        else:
            line, self.pos = self.data[self.pos:nl], nl + 1
        return True, line

    def tellg(self):
        if self.eof or self.fail:
            self.fail = True
            return -1
        return self.pos

    def seekg(self, pos):
        self.eof = False
        if not self.fail:
            self.pos = pos


class _Tokens:
    """std::istringstream iss(line); iss >> s : a failed extraction leaves s unchanged"""

    def __init__(self, line):
        self.line, self.i = line, 0

    def read(self, s):
        n = len(self.line)
        while self.i < n and self.line[self.i] in _SPACE:
            self.i += 1
        if self.i >= n:
            return s
        j = self.i
        while j < n and self.line[j] not in _SPACE:
            j += 1
        s, self.i = self.line[self.i:j], j
        return s


def _load_element(f):  # ply.cpp:121-158
    oldpos = f.tellg()
    ok, line = f.getline(b"")
    iss = _Tokens(line)
    s = iss.read(b"")
    s = iss.read(s)
    name = s
    s = iss.read(s)
    count = _libc.atoi(s)
    property_size = 0
    oldpos = f.tellg()
    while True:
        ok, line = f.getline(line)
        if not ok:
            break
        s = _Tokens(line).read(b"")
        if s != b"property":
            break
        property_size += 1
        oldpos = f.tellg()
    f.seekg(oldpos)
    return {"name": name.decode("latin-1"), "num_of_elements": count, "num_of_property": property_size}


def load(filename):  # ply.cpp:13-81
    """returns (elements, vertices [n][6] as a flat list of floats, faces as a list of lists)"""
    with open(filename, "rb") as fh:
        f = _Stream(fh.read())
    elements = []
    line = b""
    oldpos = f.tellg()
    while True:  # :21-35
        ok, line = f.getline(line)
        if not ok:
            break
        if line.find(b"element") == 0:
            f.seekg(oldpos)
            elements.append(_load_element(f))
        if line.find(b"end_header") == 0:
            break
        oldpos = f.tellg()
    vertices = []
    for _ in range(elements[0]["num_of_elements"]):  # :42-59
        ok, line = f.getline(line)
        iss = _Tokens(line)
        s = b""
        for _j in range(6):
            s = iss.read(s)
            vertices.append(_libc.atof(s))
    faces = []
    for _ in range(elements[1]["num_of_elements"]):  # :64-79
        ok, line = f.getline(line)
        iss = _Tokens(line)
        s = iss.read(b"")
        size = _libc.atoi(s)
        face = []
        for _j in range(size):
            s = iss.read(s)
            face.append(_libc.atoi(s))
        faces.append(face)
    return elements, vertices, faces
