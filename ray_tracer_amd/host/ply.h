// The reference's `ply` container and reader (ply.h:8-35, ply.cpp:13-158), host side, with flat tables instead of
// one std::vector per vertex and per face (a 1M-face file would otherwise cost 2M small allocations).
// The ASCII reader reproduces the reference's token rules exactly; binary little-endian files, which the reference
// cannot read at all, are an addition.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mi355host {

class element { // ply.h:8-17
public:
	std::string name;
	size_t num_of_elements = 0;
	size_t num_of_property = 0;
};

class ply {
public:
	bool Load(const char *filename); // ply.cpp:13-81; false (with `error` set) where the reference would crash
	void Clear(void);

	std::vector<element> elements;
	// vertices[i][0] and vertices[i][1] of the reference = position and normal of vertex i
	std::vector<double> vertex_data; // [n][6]: x y z nx ny nz
	// faces[i] of the reference = face_index[face_begin[i] .. face_begin[i + 1])
	std::vector<int32_t> face_index;
	std::vector<uint64_t> face_begin{0};

	size_t n_vertices(void) const { return vertex_data.size() / 6; }
	size_t n_faces(void) const { return face_begin.size() - 1; }
	const double *vertex(size_t i) const { return &vertex_data[6 * i]; }
	const double *normal(size_t i) const { return &vertex_data[6 * i + 3]; }
	// true when the vertex element declares at least six properties, i.e. when columns 3..5 exist in the file
	// (the reference fills the normal from repeated tokens otherwise, ply.cpp:52-55)
	bool has_normals(void) const { return !elements.empty() && elements[0].num_of_property >= 6; }

	bool binary = false;
	std::string error;
};

// How a ply becomes the mesh tables of rt_upload_mesh: one object, one polygon per face.  The reference never got
// this far (its PlyModel is commented out, object.cpp:153-233); the rules follow ObjModel (object.cpp:94-140).
enum class PlyNormals {
	Auto,   // File when the file has normals, else Smooth
	File,   // per-vertex normals from the file
	Flat,   // one normal per face: unit((v1 - v0) x (v2 - v1)), the face normal ObjModel computes
	Smooth, // per-vertex sum of the (area-weighted) face normals
};

struct MeshTables {
	std::vector<double> v, vt, vn;
	std::vector<int64_t> corners;
	std::vector<uint64_t> face_begin{0}, object_begin{0};
	std::vector<std::string> object_names;
};

// appends the ply as one more object; faces with fewer than 3 corners or out-of-range indices are dropped and counted
size_t AppendPly(const ply &p, const std::string &object_name, PlyNormals mode, MeshTables &out);

} // namespace mi355host
