"""Writes the PLY fixtures tests/golden/ply/*.ply (hand-made here, not taken from anywhere) and records in
tests/golden/ply_vectors.npz what the reference's own ply.cpp (compiled unmodified into oracle/_ref/libref_ply.so)
reads from each of them.  Run from the repository root, where the reference checkout is:  python tests/golden/make_golden_ply.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402

FIXTURES = {
    # the ordinary case: positions and normals, triangles and one quad
    "cube_normals.ply": None,
    # no normals: the reference fills columns 3..5 from the repeated last token (ply.cpp:52-55)
    "tetra_no_normals.ply": b"""ply
format ascii 1.0
comment made by hand
element vertex 4
property float x
property float y
property float z
element face 4
property list uchar int vertex_indices
end_header
0 0 0
1 0 0
0 1 0
0 0 1
3 0 2 1
3 0 1 3
3 0 3 2
3 1 2 3
""",
    # five columns (confidence, intensity as in scanned models), CRLF line ends, tabs, exponents, signs, junk suffixes
    "scan_like_crlf.ply": b"ply\r\nformat ascii 1.0\r\nelement vertex 5\r\nproperty float x\r\nproperty float y\r\nproperty float z\r\n"
                          b"property float confidence\r\nproperty float intensity\r\nelement face 3\r\nproperty list uchar int vertex_indices\r\n"
                          b"end_header\r\n-1.5e-1\t2.5E+0 +3 0.25 0.5\r\n1 2 3 4 5\r\n.5 -.5 5. 1e0 1e-3\r\n0x10 1.0f 2abc nan 7\r\n"
                          b"1e400 -1e400 1e-400 inf -inf\r\n3 0 1 2\r\n3 2 1 4 \r\n3 3 2 1 9 9\r\n",
    # missing tokens, an empty line inside each block, a short face line, a face with a float-looking index, fewer lines than declared
    "ragged.ply": b"""ply
format ascii 1.0
element vertex 6
property float x
property float y
property float z
property float nx
property float ny
property float nz
element face 5
property list uchar int vertex_indices
end_header
0 0 0 0 0 1
1 0 0
0 1

1 1 0 0 0 1 99 98
2 2
3 0 1 2
4 0 1 3

3 4 5
3 1.9 2e1 -3
""",
    # comment between the properties of an element (ends its property list), a third element, "element" without a count,
    # properties after the comment are skipped by the header loop
    "odd_header.ply": b"""ply
format ascii 1.0
element vertex 3
property float x
comment interrupts the list
property float y
property float z
element face 1
property list uchar int vertex_indices
element edge
property int a
end_header
0 0 0 0 0 1
1 0 0 0 0 1
0 1 0 0 0 1
3 0 1 2
""",
    # no end_header at all: the header loop runs to the end of the file and every vertex and face line is empty
    "no_end_header.ply": b"""ply
format ascii 1.0
element vertex 2
property float x
element face 1
property list uchar int vertex_indices
""",
    # declares more vertices and faces than it has lines, and the last line has no newline: every getline past the end
    # fails without touching the line, so the reference keeps re-reading the last line
    "truncated_no_newline.ply": b"ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                                b"element face 3\nproperty list uchar int vertex_indices\nend_header\n0.5 1.5 2.5\n3 0 1 1",
    # complete file whose last line has no newline
    "no_trailing_newline.ply": b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                               b"element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2",
    # the header ends the file without a newline and without end_header
    "header_only_no_newline.ply": b"ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nelement face 2\nproperty list uchar int 7",
}


def cube_with_normals():
    rng = np.random.default_rng(5)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * 0.5 + rng.normal(size=(8, 3)) * 1e-3
    n = v / np.linalg.norm(v, axis=1, keepdims=True)
    faces = [[0, 1, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
    lines = ["ply", "format ascii 1.0", "element vertex 8"] + ["property float %s" % c for c in ("x", "y", "z", "nx", "ny", "nz")]
    lines += ["element face %d" % len(faces), "property list uchar int vertex_indices", "end_header"]
    lines += [" ".join(repr(float(x)) for x in np.concatenate([v[i], n[i]])) for i in range(8)]
    lines += [" ".join([str(len(f))] + [str(i) for i in f]) for f in faces]
    return ("\n".join(lines) + "\n").encode()


def main():
    L = oracle.ref_ply()
    assert L is not None, "build oracle/_ref first (make -C oracle)"
    FIXTURES["cube_normals.ply"] = cube_with_normals()
    os.makedirs(os.path.join(HERE, "ply"), exist_ok=True)
    out = {}
    for name, data in FIXTURES.items():
        path = os.path.join(HERE, "ply", name)
        with open(path, "wb") as f:
            f.write(data)
        elements, vertices, face_begin, face_index = oracle.ref_ply_load(L, path)
        key = name[:-4]
        out[key + "_vertices"] = vertices
        out[key + "_face_begin"] = face_begin
        out[key + "_face_index"] = face_index
        out[key + "_elements"] = np.frombuffer(json.dumps(elements).encode(), dtype=np.uint8)
        print(name, elements, vertices.shape, face_begin.tolist())
    np.savez_compressed(os.path.join(HERE, "ply_vectors.npz"), **out)


if __name__ == "__main__":
    main()
