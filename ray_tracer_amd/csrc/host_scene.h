// Host-side scene preparation: what ObjModel::ObjModel does to an `obj` (reference object.cpp:94-140),
// kept in flat arrays instead of a tree of shared_ptr<Hittable>.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "mi355rt.h"

namespace mi355rt {

struct HostScene {
	uint64_t n_faces = 0, n_corners = 0, n_tris = 0;
	uint32_t n_objects = 0;
	std::vector<uint64_t> face_begin;   // n_faces+1 -> corner arrays
	std::vector<uint64_t> tri_begin;    // n_faces+1 -> fan triangles
	std::vector<uint64_t> object_begin; // n_objects+1 -> faces
	std::vector<double> pv, pvt, pn;    // per corner xyz: position, uv0, unit vertex normal
	std::vector<double> face_normal;    // per face xyz
	std::vector<double> polygon_area;   // per face
	std::vector<double> tri_cum;        // per fan triangle: normalised cumulative area inside its face
	std::vector<uint32_t> tri_face;     // per fan triangle
	std::vector<uint32_t> face_object;  // per face

	// fills everything from the flattened obj container; false + message on a malformed mesh
	bool build(const rt_mesh &m, std::string &err);
	// vertices of fan triangle t (triangle k of face f is (v[0], v[k+1], v[k+2]))
	void tri_vertices(uint64_t t, const double *&v0, const double *&v1, const double *&v2) const
	{
		uint32_t f = tri_face[t];
		uint64_t b = face_begin[f], k = t - tri_begin[f];
		v0 = &pv[3 * b];
		v1 = &pv[3 * (b + k + 1)];
		v2 = &pv[3 * (b + k + 2)];
	}
};

} // namespace mi355rt
