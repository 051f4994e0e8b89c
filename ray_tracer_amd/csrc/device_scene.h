// Device-resident scene: pointers into HBM, passed to kernels by value.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "mi355rt.h"

namespace mi355rt {

// Geometry the traversal kernels read.
//   nodes     WideNode (bvh_builder.h), 128 bytes each, read as 5 x uint4 (80 bytes are used)
//   tri       per leaf-ordered triangle one record of tri_stride double2 (8 = one 128-byte line; 7 for scenes small enough to live
//             in the caches, where density counts and alignment does not):
//               [0..2] v0.xyz, face_normal.xyz   (the plane; enough to compute t)
//               [3..5] v1.xyz, v2.xyz
//               [6]    as uint4: fan triangle, face (polygon), obj object, 0     (what a hit record needs)
//               [7]    unused
//             One triangle test touches exactly one line; two 48-byte records in separate arrays straddled lines 37 % of the
//             time and cost 2.75 lines per test, the largest item after the nodes on scenes that wait for memory.
//   leaf_tri  leaf order -> fan triangle id
struct TraceWork; // trace.h

struct DeviceGeom {
	const float4 *nodes;
	const double2 *tri;
	uint32_t tri_stride;         // double2 per triangle record: 7 or 8
	const uint32_t *leaf_tri;
	const uint32_t *tri_face;    // fan triangle -> face (polygon)
	const uint32_t *face_object; // face -> obj object
	uint32_t n_tris;
	uint32_t depth;
	uint32_t watertight;         // 0: the reference's triangle test; 1: the watertight test (rt_set_triangle_test)
	uint32_t surface_rays;       // 1: the rays are an integrator's (set by the render passes on their copy): they end on surfaces, so most
	                             // node visits find a leaf and the kernel variant that always fetches a node's whole header pays
	uint32_t ordered_rays;       // 1: the rays of this launch have been ordered by the cell of their origin (ray_bins.h): every XCD takes a contiguous eighth
	TraceWork *work;             // host side only (kernels ignore it): the context's work counters for the traversal launches, or null
};

} // namespace mi355rt
