// Device-resident scene: pointers into HBM, passed to kernels by value.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "mi355rt.h"

namespace mi355rt {

// Geometry the traversal kernels read.
//   nodes     WideNode (bvh_builder.h), 128 bytes each, read as 5 x uint4 (80 bytes are used)
//   tri_a     per leaf-ordered triangle, 6 doubles: v0.xyz, face_normal.xyz   (the plane; enough to compute t)
//   tri_b     per leaf-ordered triangle, 6 doubles: v1.xyz, v2.xyz            (read only when t is in range)
//   leaf_tri  leaf order -> fan triangle id
struct DeviceGeom {
	const float4 *nodes;
	const double2 *tri_a;
	const double2 *tri_b;
	const uint32_t *leaf_tri;
	const uint32_t *tri_face;    // fan triangle -> face (polygon)
	const uint32_t *face_object; // face -> obj object
	const uint2 *tri_fo;         // fan triangle -> (face, object): what a hit record needs, in one read
	uint32_t n_tris;
	uint32_t depth;
};

} // namespace mi355rt
