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
	std::vector<uint32_t> tri_index;   // leaf order -> fan triangle id (a triangle whose reference was cut appears once per piece, presplit.h)
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
	// the collapse to 8-wide nodes: 1 = the cut of the binary tree that minimises the surface-area cost of the wide tree (dynamic
	// programme over the number of roots a subtree is cut into), 0 = open the child with the largest area until there are eight
	int optimal_cut = 1;
	int slot_assignment = 1; // 1: the assignment of children to slots that maximises the sum of (centre offset) . (the slot's octant direction), 2: the same with
	                         // offsets in units of the node's extents, 0: children take the free slot nearest their octant, far children first (1-4 % slower to trace)
	float wide_cost_node = 1.0f; // one 8-wide node fetched and tested
	float wide_cost_tri = 1.5f;  // one triangle record fetched and tested (swept 0.3 .. 3 on the Cornell box, the 1M soup and a 490 k-triangle
	                             // hall: 1.5 is the best or within noise of it on all three; 0.5 costs the Cornell box 17 %)
};

// The tree the traversal kernel walks: the BVH2 collapsed to 8 children per node, child boxes quantised to
// 8 bits per coordinate inside the node's own box (always rounded outward), 128 bytes = one L2 line per node.
//   bytes  0..11  org[3]      node box minimum (float)
//         12..14  e[3]        biased exponents: cell size on axis a is 2^(e[a]-127)
//         15      imask       bit i: child i is an interior node
//         16..19  child_base  index of the first interior child; interior children are consecutive in slot order
//         20..23  tri_base    first triangle (leaf order) of this node's leaf children, consecutive in slot order
//         24..31  meta[8]     leaf child: (count << 5) | offset from tri_base; otherwise 0
//         32..55  qlo[3][8]   per axis, per child: lower bound in cells
//         56..79  qhi[3][8]   per axis, per child: upper bound in cells  (empty slot: qlo 255 > qhi 0)
// Children sit in the slot whose three bits say on which side of the node's centre they lie, so that
// slot XOR ray-octant orders them front to back without sorting.
struct WideNode {
	float org[3];
	uint8_t e[3];
	uint8_t imask;
	uint32_t child_base;
	uint32_t tri_base;
	uint8_t meta[8];
	uint8_t qlo[3][8];
	uint8_t qhi[3][8];
	uint8_t pad[48];
};
static_assert(sizeof(WideNode) == 128, "one node per 128-byte line");

struct BvhWide {
	std::vector<WideNode> nodes;     // node 0 = root
	std::vector<uint32_t> tri_index; // leaf order -> fan triangle id (once per piece for a triangle whose reference was cut)
	uint32_t depth = 0;              // nodes on the longest root-to-leaf path
};

// builds the binary tree once and flattens it twice (out2 for introspection / work counting, wide for the device);
// cut: references of long triangles cut into pieces (presplit.h), or null
struct Presplit;
void build_bvh(const HostScene &scene, const BvhBuildOptions &opt, const Presplit *cut, Bvh2 &out2, BvhWide &wide);

inline int32_t make_leaf_link(uint32_t first, uint32_t count) { return ~(int32_t)(((count - 1u) << 27) | first); }

} // namespace mi355rt
