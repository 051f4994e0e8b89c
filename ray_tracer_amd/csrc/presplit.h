// References of long triangles cut into pieces before the tree is built.
//
// A bounding volume hierarchy stores every triangle once, under the box of the whole triangle.  A long thin triangle that lies
// diagonally in space (a brace, a cable, a roof edge of a modelled building) has a box that covers far more space than the triangle
// does: every ray through that space finds the box, descends to the leaf and pays an exact fp64 test for nothing.  A few hundred such
// triangles among half a million small ones cost the hall scene of the tests 60 % of its ray rate.  The remedy is the one of
// Karras and Aila, "Fast Parallel Construction of High-Quality Bounding Volume Hierarchies" (HPG 2013), section 4: the tree is built
// over references (triangle, box) instead of triangles, and the reference of a triangle whose box wastes much space is cut, before
// the build, along the planes a spatial-median hierarchy over the scene would use, into several references with the boxes of the
// pieces.  Both builders take the result (the host builder's binned SAH and the device builder's Morton tree); the traversal kernel
// needs no change: a triangle record carries its fan triangle, polygon and object, and the same triangle found through two leaves
// gives the same t and the same ids.  The reference's own tree (bvh.cpp:134-232) has no counterpart; its results do not change.
//
// A piece's box must contain every point of its part of space that the reference's polygon test can accept for the triangle
// (accept_region.h): the triangle of the lifted corners is clipped against the piece's part of space, together with the
// triangle itself and the corners' triangle in its plane, and the box of what is left is padded against the rounding of the
// clipping and rounded outward to float.  Neighbouring pieces overlap by two float steps at the plane between them, so that a hit
// point within rounding of the plane lies inside both.  A triangle whose region is only known as a box (accept_region_corners
// returns -1) is not cut.
#pragma once
#include <cstdint>
#include <vector>
#include "host_scene.h"

namespace mi355rt {

struct SplitRef {
	uint32_t tri;      // fan triangle
	float lo[3], hi[3];
};

struct PresplitOptions {
	float budget = 0.25f;      // at most this many additional references per triangle of the scene, over the whole scene
	uint32_t min_tris = 4096;  // smaller scenes are left alone (they live in the caches; their trees are a few levels deep)
	uint32_t max_pieces = 256; // per triangle
	int threads = 0;           // 0 = hardware concurrency (capped at 32)
};

struct Presplit {
	// A triangle that is cut keeps its own reference with the box of its first piece (`first`, ordered by triangle) and gets
	// additional references for the other pieces (`extra`, grouped by triangle in the same order).  Empty: nothing was cut.
	std::vector<SplitRef> first, extra;
	double seconds = 0.0;
};

void presplit(const HostScene &scene, const PresplitOptions &opt, Presplit &out);

} // namespace mi355rt
