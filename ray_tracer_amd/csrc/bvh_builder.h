// Host BVH2 builder: binned SAH over fan triangles, flattened to 64-byte nodes that carry both
// child boxes (float, rounded outward from the fp64 vertices).  Replaces the pointer tree that
// SAHBVHNode::SAHBVHNode builds (reference bvh.cpp:134-232); topology is free to differ because a
// closest hit does not depend on it (ties in t aside).
#pragma once
#include <cstdint>
#include <vector>
#include "host_scene.h"
#include "mi355rt.h"

namespace mi355rt {

struct Bvh2 {
	std::vector<rt_bvh2_node> nodes;   // node 0 = root, preorder
	std::vector<uint32_t> tri_index;   // leaf order -> fan triangle id
	uint32_t depth = 0;                // interior levels on the longest root-to-leaf path
	uint64_t n_leaves = 0;
	double build_seconds = 0.0;
};

struct BvhBuildOptions {
	int max_leaf = 4;       // triangles per leaf, at most 16
	int bins = 32;
	float cost_node = 1.0f; // SAH: one interior step
	float cost_tri = 1.5f;  // SAH: one triangle test
	int sah_depth_limit = 48; // below this depth: median splits, which bound the total depth
	int threads = 0;        // 0 = hardware concurrency (capped at 32)
};

void build_bvh2(const HostScene &scene, const BvhBuildOptions &opt, Bvh2 &out);

inline int32_t make_leaf_link(uint32_t first, uint32_t count) { return ~(int32_t)(((count - 1u) << 27) | first); }

} // namespace mi355rt
