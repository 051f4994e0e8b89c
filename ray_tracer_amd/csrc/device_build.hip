// BVH construction on the device: Morton-ordered binary radix tree (one thread per internal node), bottom-up
// box fit, then a level-by-level collapse into the 8-wide quantised nodes the traversal kernel walks.
// Stands in for SAHBVHNode::SAHBVHNode (reference bvh.cpp:134-232) when the scene is large: the topology is free
// to differ from the reference's, because a closest hit does not depend on it (ties in t aside).
//
// Stages (n = fan triangles, m = references: the triangles plus the additional pieces of those the host has cut, presplit.h;
// all buffers in HBM):
//   k_tri_boxes     fp64 vertices -> fp32 boxes rounded outward, scene bounds of the box centres (ordered-int atomics)
//   k_place_refs    boxes of the pieces: a cut triangle's own reference gets its first piece's box, the others are references n .. m - 1
//   k_morton        63-bit Morton key of each box centre (21 bits per axis)
//   radix sort      eight stable passes over eight bits of the key (k_sort_hist, k_scan_*, k_sort_scatter)
//   k_hierarchy     binary radix tree over the sorted keys; ties between equal keys are broken by position
//   k_fit_*         bottom up in rounds (one launch per level of readiness): boxes, subtree costs, the cut into wide nodes
//   k_collapse      one thread per wide node and level: opens the child with the largest area until there are 8,
//                   assigns slots by octant, quantises the child boxes, reserves its interior children and its leaf
//                   triangles with one atomic each, and queues the interior children for the next level
//   k_gather        triangle records into leaf order
#include <hip/hip_runtime.h>
#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <string>
#include "accept_region.h"
#include "bvh_builder.h"
#include "device_build.h"
#include "presplit.h"
#include "trace.h"

namespace mi355rt {

namespace {

constexpr int kBlock = 256;
constexpr unsigned kStrideBlocks = 4096; // grid of the kernels that walk through their input in strides
constexpr uint32_t kMaxLeaf = 4; // triangles per leaf child: a wide node addresses 8 x 4 = 32 triangles
#ifndef BUILD_COST_TRI
#define BUILD_COST_TRI 1.5f
#endif
constexpr float kCostNode = 1.0f, kCostTri = BUILD_COST_TRI; // surface-area heuristic: one interior step, one triangle test

#define DB_TRY(x)                                                                       \
	do {                                                                                \
		hipError_t e_ = (x);                                                            \
		if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return e_; } \
	} while (0)

// order-preserving map float -> int, so that integer atomicMin/Max order floats
__device__ __forceinline__ int float_to_ordered(float f)
{
	int i = __float_as_int(f);
	return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__host__ __device__ inline float ordered_to_float(int i)
{
	int j = i >= 0 ? i : i ^ 0x7FFFFFFF;
#ifdef __HIP_DEVICE_COMPILE__
	return __int_as_float(j);
#else
	float f;
	memcpy(&f, &j, 4);
	return f;
#endif
}

struct Box4 {
	float4 lo, hi; // w unused
};

__global__ void __launch_bounds__(kBlock) k_tri_boxes(const double2 *__restrict__ tri_a, const double2 *__restrict__ tri_b, uint32_t n,
                                                      Box4 *__restrict__ boxes, int *__restrict__ bounds)
{
	float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
		const double2 a0 = tri_a[3 * (size_t)i], a1 = tri_a[3 * (size_t)i + 1], a2 = tri_a[3 * (size_t)i + 2];
		const double2 b0 = tri_b[3 * (size_t)i], b1 = tri_b[3 * (size_t)i + 1], b2 = tri_b[3 * (size_t)i + 2];
		// the box of everything the reference's test of this fan triangle can accept: the vertices' box for a planar polygon, more
		// for a polygon that is not (accept_region.h)
		const double v0[3] = {a0.x, a0.y, a1.x}, nf[3] = {a1.y, a2.x, a2.y}, v1[3] = {b0.x, b0.y, b1.x}, v2[3] = {b1.y, b2.x, b2.y};
		double rlo[3], rhi[3];
		accept_region_box(v0, v1, v2, nf, rlo, rhi);
		Box4 b;
		b.lo = make_float4(__double2float_rd(rlo[0]), __double2float_rd(rlo[1]), __double2float_rd(rlo[2]), 0.0f);
		b.hi = make_float4(__double2float_ru(rhi[0]), __double2float_ru(rhi[1]), __double2float_ru(rhi[2]), 0.0f);
		boxes[i] = b;
		const float cx = 0.5f * (b.lo.x + b.hi.x), cy = 0.5f * (b.lo.y + b.hi.y), cz = 0.5f * (b.lo.z + b.hi.z);
		clo[0] = fminf(clo[0], cx); chi[0] = fmaxf(chi[0], cx);
		clo[1] = fminf(clo[1], cy); chi[1] = fmaxf(chi[1], cy);
		clo[2] = fminf(clo[2], cz); chi[2] = fmaxf(chi[2], cz);
	}
	// wave reduction, workgroup reduction through LDS, then one atomic per workgroup and component: the grid is a few thousand
	// workgroups, because atomics on one address retire one after the other (one per wave of a 10 M-triangle launch took 10 ms)
	__shared__ float part[2][3][kBlock / 64];
	for (int a = 0; a < 3; a++) {
		float lo = clo[a], hi = chi[a];
		for (int off = 32; off > 0; off >>= 1) {
			lo = fminf(lo, __shfl_xor(lo, off));
			hi = fmaxf(hi, __shfl_xor(hi, off));
		}
		if ((threadIdx.x & 63) == 0) { part[0][a][threadIdx.x >> 6] = lo; part[1][a][threadIdx.x >> 6] = hi; }
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		float lo = part[0][threadIdx.x][0], hi = part[1][threadIdx.x][0];
		for (int w = 1; w < kBlock / 64; w++) { lo = fminf(lo, part[0][threadIdx.x][w]); hi = fmaxf(hi, part[1][threadIdx.x][w]); }
		if (lo <= hi) {
			atomicMin(&bounds[threadIdx.x], float_to_ordered(lo));
			atomicMax(&bounds[3 + threadIdx.x], float_to_ordered(hi));
		}
	}
}

// ref: piece k of the list goes to reference at[k] (a cut triangle's own slot, or n + k for the additional ones)
struct RefBox { float lo[3], hi[3]; };
__global__ void __launch_bounds__(kBlock) k_place_refs(const RefBox *__restrict__ pieces, const uint32_t *__restrict__ at, uint32_t base, uint32_t count, Box4 *__restrict__ boxes)
{
	const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= count) return;
	const RefBox r = pieces[k];
	Box4 b;
	b.lo = make_float4(r.lo[0], r.lo[1], r.lo[2], 0.0f);
	b.hi = make_float4(r.hi[0], r.hi[1], r.hi[2], 0.0f);
	boxes[at ? at[k] : base + k] = b;
}

__device__ __forceinline__ uint64_t spread21(uint64_t x)
{
	x &= 0x1FFFFFull;
	x = (x | (x << 32)) & 0x1F00000000FFFFull;
	x = (x | (x << 16)) & 0x1F0000FF0000FFull;
	x = (x | (x << 8)) & 0x100F00F00F00F00Full;
	x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}

__global__ void __launch_bounds__(kBlock) k_morton(const Box4 *__restrict__ boxes, uint32_t n, const int *__restrict__ bounds,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const Box4 b = boxes[i];
	const float c[3] = {0.5f * (b.lo.x + b.hi.x), 0.5f * (b.lo.y + b.hi.y), 0.5f * (b.lo.z + b.hi.z)};
	uint64_t q[3];
	for (int a = 0; a < 3; a++) {
		const float lo = ordered_to_float(bounds[a]), hi = ordered_to_float(bounds[3 + a]);
		const float ext = hi - lo;
		float u = ext > 0.0f ? (c[a] - lo) / ext : 0.0f;
		u = fminf(fmaxf(u, 0.0f), 1.0f);
		q[a] = (uint64_t)fminf(u * 2097152.0f, 2097151.0f);
	}
	keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
	vals[i] = i;
}

// ---- radix sort of (63-bit key, triangle) pairs: eight passes over eight bits, least significant first, each pass stable.
// Per pass: a digit histogram per tile of kSortTile keys, an exclusive prefix sum over the table in digit-major order (which is
// the order of the output: all keys of digit 0 by tile, then digit 1, ...), and a scatter in which every key finds its rank among
// the keys of its digit inside the tile.  32 bytes of traffic per key and pass.
constexpr int kSortPer = 16;                      // keys per thread
constexpr uint32_t kSortTile = kBlock * kSortPer; // keys per workgroup; wave w of the workgroup owns a contiguous quarter of them

__global__ void __launch_bounds__(kBlock) k_sort_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t *__restrict__ table, uint32_t n_tiles)
{
	__shared__ uint32_t hist[256];
	hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t base = blockIdx.x * kSortTile;
	for (int j = 0; j < kSortPer; j++) {
		const uint32_t i = base + (uint32_t)j * kBlock + threadIdx.x;
		if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 0xFFu], 1u);
	}
	__syncthreads();
	table[threadIdx.x * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

// exclusive prefix sum of m words in place: sums of tiles, their running sum by one workgroup, the tiles from their bases
constexpr uint32_t kScanPer = 8, kScanTile = kBlock * kScanPer;

__device__ __forceinline__ uint32_t block_inclusive_sum(uint32_t v, uint32_t *s_wave, uint32_t *total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t x = v;
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t t = __shfl_up(x, off);
		if (lane >= off) x += t;
	}
	__syncthreads(); // s_wave may still be read from the previous call
	if (lane == 63) s_wave[wv] = x;
	__syncthreads();
	uint32_t before = 0, all = 0;
	for (int w = 0; w < kBlock / 64; w++) {
		const uint32_t t = s_wave[w];
		if (w < wv) before += t;
		all += t;
	}
	*total = all;
	return x + before;
}

__global__ void __launch_bounds__(kBlock) k_scan_tiles(const uint32_t *__restrict__ data, uint32_t m, uint32_t *__restrict__ tile_sums)
{
	__shared__ uint32_t s_wave[kBlock / 64];
	const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
	uint32_t mine = 0;
	for (uint32_t k = 0; k < kScanPer; k++)
		if (first + k < m) mine += data[first + k];
	uint32_t total;
	(void)block_inclusive_sum(mine, s_wave, &total);
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) k_scan_spine(uint32_t *__restrict__ tile_sums, uint32_t n_tiles)
{
	__shared__ uint32_t s_wave[kBlock / 64];
	uint32_t carry = 0;
	for (uint32_t at = 0; at < n_tiles; at += kBlock) {
		const uint32_t i = at + threadIdx.x;
		const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
		uint32_t total;
		const uint32_t incl = block_inclusive_sum(v, s_wave, &total);
		if (i < n_tiles) tile_sums[i] = carry + incl - v;
		carry += total;
	}
}

__global__ void __launch_bounds__(kBlock) k_scan_apply(uint32_t *__restrict__ data, uint32_t m, const uint32_t *__restrict__ tile_base)
{
	__shared__ uint32_t s_wave[kBlock / 64];
	const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
	uint32_t c[kScanPer], mine = 0;
	for (uint32_t k = 0; k < kScanPer; k++) {
		c[k] = first + k < m ? data[first + k] : 0u;
		mine += c[k];
	}
	uint32_t total;
	uint32_t run = tile_base[blockIdx.x] + block_inclusive_sum(mine, s_wave, &total) - mine;
	for (uint32_t k = 0; k < kScanPer; k++) {
		if (first + k < m) data[first + k] = run;
		run += c[k];
	}
}

// Stable scatter of one tile.  Wave w owns keys [w * 1024, (w + 1) * 1024) of the tile and goes through them 64 at a time, in
// order: the lanes that hold the same digit find each other with eight ballots, a lane's rank among them is a population count,
// and the number of earlier keys of that digit in the wave's part is a counter per (wave, digit) in LDS that the lowest lane of
// each group advances.  The counters of the waves are then chained digit by digit and moved to the tile's bases in the output.
__global__ void __launch_bounds__(kBlock) k_sort_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n, int shift,
                                                         const uint32_t *__restrict__ table, uint32_t n_tiles,
                                                         uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out)
{
	__shared__ uint32_t count[kBlock / 64][256];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int w = 0; w < kBlock / 64; w++) count[w][threadIdx.x] = 0;
	__syncthreads();
	const uint32_t first = blockIdx.x * kSortTile + (uint32_t)wv * (64u * kSortPer) + (uint32_t)lane;
	uint64_t key[kSortPer];
	uint32_t local[kSortPer]; // digit << 16 | rank of the key among the keys of its digit in the wave's part
	const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
	for (int j = 0; j < kSortPer; j++) {
		const uint32_t i = first + 64u * (uint32_t)j;
		const bool valid = i < n;
		key[j] = valid ? keys[i] : ~0ull;
		const uint32_t d = (uint32_t)(key[j] >> shift) & 0xFFu;
		unsigned long long peers = __ballot(valid);
		for (int b = 0; b < 8; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __ballot(bit);
			peers &= bit ? m : ~m;
		}
		const uint32_t before = count[wv][d]; // read by every lane of the group before its lowest lane adds the group's size
		local[j] = (d << 16) | (before + (uint32_t)__popcll(peers & below));
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (valid && (peers & below) == 0ull) count[wv][d] = before + (uint32_t)__popcll(peers);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	__syncthreads();
	{ // thread d: where each wave's keys of digit d start in the output
		uint32_t run = table[threadIdx.x * n_tiles + blockIdx.x];
		for (int w = 0; w < kBlock / 64; w++) {
			const uint32_t c = count[w][threadIdx.x];
			count[w][threadIdx.x] = run;
			run += c;
		}
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < kSortPer; j++) {
		const uint32_t i = first + 64u * (uint32_t)j;
		if (i < n) {
			const uint32_t to = count[wv][local[j] >> 16] + (local[j] & 0xFFFFu);
			keys_out[to] = key[j];
			vals_out[to] = vals[i];
		}
	}
}

// Binary radix tree.  Internal node i in [0, n-1), leaf j in [0, n) = position in the sorted order.
// Child links: >= 0 internal node, < 0 leaf ~position.
struct BinNode {
	int32_t left, right;
	uint32_t first, last; // range of sorted positions under this node
};

__device__ __forceinline__ int delta(const uint64_t *__restrict__ keys, int n, int i, int j)
{
	if (j < 0 || j >= n) return -1;
	const uint64_t k = keys[i] ^ keys[j];
	if (k == 0) return 64 + __clz((unsigned)(i ^ j));
	return __clzll((long long)k);
}

__global__ void __launch_bounds__(kBlock) k_hierarchy(const uint64_t *__restrict__ keys, int n, BinNode *__restrict__ nodes,
                                                      int32_t *__restrict__ parent_of_node, int32_t *__restrict__ parent_of_leaf)
{
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n - 1) return;
	const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
	const int dmin = delta(keys, n, i, i - d);
	int lmax = 2;
	while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1)
		if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = delta(keys, n, i, j);
	int s = 0, t = l;
	do {
		t = (t + 1) >> 1;
		if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	const int gamma = i + s * d + min(d, 0);
	const int lo = min(i, j), hi = max(i, j);
	BinNode nd;
	nd.first = (uint32_t)lo;
	nd.last = (uint32_t)hi;
	if (lo == gamma) { nd.left = ~gamma; parent_of_leaf[gamma] = i; }
	else { nd.left = gamma; parent_of_node[gamma] = i; }
	if (hi == gamma + 1) { nd.right = ~(gamma + 1); parent_of_leaf[gamma + 1] = i; }
	else { nd.right = gamma + 1; parent_of_node[gamma + 1] = i; }
	nodes[i] = nd;
	if (i == 0) parent_of_node[0] = -1;
}

__device__ __forceinline__ Box4 box_union(const Box4 &a, const Box4 &b)
{
	Box4 r;
	r.lo = make_float4(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z), 0.0f);
	r.hi = make_float4(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z), 0.0f);
	return r;
}

__device__ __forceinline__ float half_area(const Box4 &b)
{
	const float dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z;
	return dx * dy + dy * dz + dz * dx;
}

// Bottom-up fit in rounds.  A node is ready once both of its children are settled; the ready nodes of one round are one launch,
// and whoever settles the second child of a node puts that node on the next round's list.  The number of rounds is the height of
// the binary tree.  (A single launch in which every leaf walks up and the second thread to arrive at a node carries on needs a
// device-wide fence per node, and on this chip, where each of the eight dies has its own L2, such a fence writes back and
// invalidates that L2: 92 ms for 10 M triangles against 4 ms for the rounds.)
//
// On the way the surface-area cost of each subtree is settled: a subtree of at most kMaxLeaf triangles becomes one leaf child of a
// wide node when testing all its triangles is estimated cheaper than descending further (same constants as the host builder).

// appends the nodes of the calling threads that have one (has) to list[*tail...]: one atomic per workgroup
__device__ __forceinline__ void append_ready(bool has, uint32_t node, uint32_t *__restrict__ list, unsigned int *__restrict__ tail)
{
	__shared__ unsigned int wave_count[kBlock / 64], wave_base[kBlock / 64];
	const unsigned long long m = __ballot(has);
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	__syncthreads(); // the arrays may still be read by the previous call
	if (lane == 0) wave_count[wv] = (unsigned int)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned int total = 0;
		for (int w = 0; w < kBlock / 64; w++) { wave_base[w] = total; total += wave_count[w]; }
		const unsigned int base = total ? atomicAdd(tail, total) : 0u;
		for (int w = 0; w < kBlock / 64; w++) wave_base[w] += base;
	}
	__syncthreads();
	if (has) list[wave_base[wv] + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = node;
}

__global__ void __launch_bounds__(kBlock) k_fit_leaves(const int32_t *__restrict__ parent_of_leaf, int n, unsigned int *__restrict__ arrived,
                                                       uint32_t *__restrict__ ready, unsigned int *__restrict__ tail)
{
	const int rounds = (n + (int)(gridDim.x * kBlock) - 1) / (int)(gridDim.x * kBlock);
	for (int r = 0; r < rounds; r++) { // every thread takes part in every append
		const int leaf = (int)((r * gridDim.x + blockIdx.x) * kBlock + threadIdx.x);
		bool second = false;
		int p = 0;
		if (leaf < n) {
			p = parent_of_leaf[leaf];
			second = atomicAdd(&arrived[p], 1u) == 1u;
		}
		append_ready(second, (uint32_t)p, ready, tail);
	}
}

// Settles one node of the binary tree whose children (links `left`, `right`) are settled: its box, and the cut into 8-wide nodes that
// minimises the surface-area cost of the wide tree, as in the host builder (bvh_builder.cpp, WideCut): cost[i - 1] is the cheapest way
// to represent the subtree by at most i roots, each a leaf child of at most kMaxLeaf triangles or an 8-wide node; cut[i - 1] = 0 means
// "as for i - 1" (a leaf when i = 1), k > 0 that the left subtree supplies k of the roots (of the eight children when i = 1).
// count: references under the node when they are a run of the sorted order (only a run can become a leaf child), else 0.
__device__ __forceinline__ Box4 fit_node(int node, int32_t left, int32_t right, uint32_t count, const Box4 *node_boxes, const float *node_cost,
                                         const Box4 *__restrict__ tri_boxes, const uint32_t *__restrict__ sorted_tri, Box4 *node_boxes_out, float *node_cost_out,
                                         uint8_t *node_cut, uint8_t *__restrict__ as_leaf)
{
	const Box4 a = left >= 0 ? node_boxes[left] : tri_boxes[sorted_tri[~left]];
	const Box4 b = right >= 0 ? node_boxes[right] : tri_boxes[sorted_tri[~right]];
	float cl[7], cr[7];
	for (int i = 0; i < 7; i++) {
		cl[i] = left >= 0 ? node_cost[7 * (size_t)left + i] : kCostTri * half_area(a);
		cr[i] = right >= 0 ? node_cost[7 * (size_t)right + i] : kCostTri * half_area(b);
	}
	const Box4 u = box_union(a, b);
	const float whole = count != 0u && count <= kMaxLeaf ? kCostTri * (float)count * half_area(u) : FLT_MAX;
	float best = FLT_MAX;
	int bk = 1;
	for (int k = 1; k <= 7; k++) {
		const float v = cl[k - 1] + cr[7 - k];
		if (v < best) { best = v; bk = k; }
	}
	const float split = kCostNode * half_area(u) + best;
	const bool leaf_wins = whole <= split;
	float c[7];
	uint8_t d[7];
	c[0] = leaf_wins ? whole : split;
	d[0] = leaf_wins ? 0 : (uint8_t)bk;
	for (int i = 2; i <= 7; i++) {
		float v = c[i - 2];
		int dk = 0;
		for (int k = 1; k < i; k++) {
			const float w = cl[k - 1] + cr[i - k - 1];
			if (w < v) { v = w; dk = k; }
		}
		c[i - 1] = v;
		d[i - 1] = (uint8_t)dk;
	}
	node_boxes_out[node] = u;
	for (int i = 0; i < 7; i++) { node_cost_out[7 * (size_t)node + i] = c[i]; node_cut[7 * (size_t)node + i] = d[i]; }
	as_leaf[node] = leaf_wins ? 1 : 0;
	return u;
}

// A node whose references are no run of the sorted order any more (after a rotation below) says so with this `last`.
constexpr uint32_t kNoRun = 0x7FFFFFFFu;

__global__ void __launch_bounds__(kBlock) k_fit_round(BinNode *nodes, const int32_t *__restrict__ parent_of_node,
                                                      const Box4 *__restrict__ tri_boxes, const uint32_t *__restrict__ sorted_tri,
                                                      const uint32_t *__restrict__ ready_in, unsigned int n_in, Box4 *node_boxes, float *node_cost,
                                                      uint8_t *node_cut, uint8_t *__restrict__ as_leaf, unsigned int *__restrict__ arrived,
                                                      uint32_t *__restrict__ ready_out, unsigned int *__restrict__ tail, int rotate)
{
	const unsigned int rounds = (n_in + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
	for (unsigned int r = 0; r < rounds; r++) {
		const unsigned int qi = (r * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
		bool second = false;
		int up = 0;
		if (qi < n_in) {
			const int node = (int)ready_in[qi];
			BinNode nd = nodes[node];
			auto box_of = [&](int32_t link) { return link >= 0 ? node_boxes[link] : tri_boxes[sorted_tri[~link]]; };
			auto first_of = [&](int32_t link) { return link >= 0 ? nodes[link].first : (uint32_t)~link; };
			auto last_of = [&](int32_t link) { return link >= 0 ? nodes[link].last : (uint32_t)~link; };
			// ---- a rotation (Kensler, "Tree Rotations for Improving Bounding Volume Hierarchies", 2008): a child of this node changes
			// places with a grandchild on the other side when that makes the box of the node in between smaller.  The Morton tree
			// splits space in the middle whatever lies there; this is the one local repair that fits into the bottom-up pass: both
			// children and all four grandchildren are settled, nobody else reads or writes them, and the node that is rebuilt (the
			// child that keeps its index and gets the other pair) is settled again right here.  Two of the four exchanges leave the
			// rebuilt node with references that are no run of the sorted order; such a node cannot become a leaf child (kNoRun).
			for (int turn = 0; turn < rotate; turn++) {
				const int32_t L = nd.left, R = nd.right;
				float gain = 0.0f;
				int which = 0;
				if (R >= 0) {
					const BinNode rn = nodes[R];
					const float now = half_area(node_boxes[R]);
					const Box4 lb = box_of(L);
					const float g1 = now - half_area(box_union(lb, box_of(rn.right))); // L <-> R.left:  R = (L, R.right)
					const float g2 = now - half_area(box_union(lb, box_of(rn.left)));  // L <-> R.right: R = (L, R.left)
					if (g1 > gain) { gain = g1; which = 1; }
					if (g2 > gain) { gain = g2; which = 2; }
				}
				if (L >= 0) {
					const BinNode ln = nodes[L];
					const float now = half_area(node_boxes[L]);
					const Box4 rb = box_of(R);
					const float g3 = now - half_area(box_union(box_of(ln.right), rb)); // R <-> L.left:  L = (L.right, R)
					const float g4 = now - half_area(box_union(box_of(ln.left), rb));  // R <-> L.right: L = (L.left, R)
					if (g3 > gain) { gain = g3; which = 3; }
					if (g4 > gain) { gain = g4; which = 4; }
				}
				if (which != 0) {
					const int32_t mid = which <= 2 ? R : L; // the node that is rebuilt
					BinNode mn = nodes[mid];
					int32_t out_link, in_link = which <= 2 ? L : R; // leaves mid, enters mid
					if (which == 1) { out_link = mn.left; mn.left = in_link; }
					else if (which == 2) { out_link = mn.right; mn.right = mn.left; mn.left = in_link; }      // (L, R.left): in sorted order
					else if (which == 3) { out_link = mn.left; mn.left = mn.right; mn.right = in_link; }       // (L.right, R): in sorted order
					else { out_link = mn.right; mn.right = in_link; }
					// a run of the sorted order only if both parts are runs and the second starts where the first ends
					const uint32_t f0 = first_of(mn.left), l0 = last_of(mn.left), f1 = first_of(mn.right), l1 = last_of(mn.right);
					const bool run = l0 != kNoRun && l1 != kNoRun && l0 + 1u == f1;
					mn.first = run ? f0 : 0u;
					mn.last = run ? l1 : kNoRun;
					nodes[mid] = mn;
					fit_node(mid, mn.left, mn.right, run ? l1 - f0 + 1u : 0u, node_boxes, node_cost, tri_boxes, sorted_tri, node_boxes, node_cost, node_cut, as_leaf);
					if (which <= 2) { nd.left = out_link; nd.right = mid; }
					else { nd.left = mid; nd.right = out_link; }
					// this node's own references are the same set as before: its run stands
					nodes[node] = nd;
				} else break;
			}
			const uint32_t count = nd.last != kNoRun ? nd.last - nd.first + 1u : 0u;
			fit_node(node, nd.left, nd.right, count, node_boxes, node_cost, tri_boxes, sorted_tri, node_boxes, node_cost, node_cut, as_leaf);
			up = parent_of_node[node];
			if (up >= 0) second = atomicAdd(&arrived[up], 1u) == 1u; // the other child was settled in an earlier launch, or is settled in this one by whoever came first
		}
		append_ready(second, (uint32_t)up, ready_out, tail);
	}
}

// smallest e with 2^e * 255 >= extent (the cell size of a wide node on one axis), clamped to the float exponent range
__device__ __forceinline__ int cell_exponent(double extent)
{
	if (!(extent > 0.0)) return -126;
	int e;
	const double f = frexp(extent / 255.0, &e); // extent / 255 = f * 2^e, f in [0.5, 1)
	if (f == 0.5) e -= 1;
	while (ldexp(1.0, e) * 255.0 < extent) e++;
	return max(-126, min(127, e));
}

struct Counters {
	unsigned int nodes;     // wide nodes handed out
	unsigned int tris;      // leaf-order positions handed out
	unsigned int queue_out; // entries appended to the next level's queue
	unsigned int overflow;  // a node needed more than 32 leaf triangles (cannot happen with kMaxLeaf 4)
};

// one thread per wide node of the current level; queue entries: x = binary node (or ~position of a single leaf for
// a one-triangle scene), y = index of the wide node to fill
__global__ void __launch_bounds__(kBlock) k_collapse(const BinNode *__restrict__ bin, const Box4 *__restrict__ node_boxes, const uint8_t *__restrict__ as_leaf,
                                                     const uint8_t *__restrict__ node_cut,
                                                     const Box4 *__restrict__ tri_boxes, const uint32_t *__restrict__ sorted_tri,
                                                     const uint2 *__restrict__ queue_in, uint32_t n_in, uint2 *__restrict__ queue_out,
                                                     WideNode *__restrict__ wide, uint32_t *__restrict__ leaf_tri, Counters *__restrict__ ctr)
{
	const uint32_t qi = blockIdx.x * kBlock + threadIdx.x;
	if (qi >= n_in) return;
	const uint2 job = queue_in[qi];
	// ---- pick up to 8 children: open the interior child with the largest area until full
	int32_t kids[8];
	Box4 kb[8];
	bool interior[8];
	int nk = 0;
	auto add = [&](int32_t link) {
		kids[nk] = link;
		if (link >= 0) {
			kb[nk] = node_boxes[link];
			interior[nk] = as_leaf[link] == 0;
		} else {
			kb[nk] = tri_boxes[sorted_tri[~link]];
			interior[nk] = false;
		}
		nk++;
	};
	const int root_k = node_cut[7 * (size_t)job.x];
	if (root_k > 0) {
		// the children the cost-minimising cut gives this node: the left subtree as root_k roots, the right one as 8 - root_k
		const BinNode root = bin[job.x];
		int32_t st_link[16];
		int st_i[16], sp = 0;
		st_link[sp] = root.right; st_i[sp++] = 8 - root_k;
		st_link[sp] = root.left; st_i[sp++] = root_k;
		while (sp > 0) {
			const int32_t link = st_link[--sp];
			int i = st_i[sp];
			if (link < 0) { add(link); continue; } // a single triangle
			while (i > 1 && node_cut[7 * (size_t)link + i - 1] == 0) i--;
			if (i == 1) { add(link); continue; }
			const int k = node_cut[7 * (size_t)link + i - 1];
			const BinNode b2 = bin[link];
			st_link[sp] = b2.right; st_i[sp++] = i - k;
			st_link[sp] = b2.left; st_i[sp++] = k;
		}
	} else { // a subtree that is a leaf as a whole reaches here only as the root of a scene of a few triangles
		const BinNode root = bin[job.x];
		add(root.left);
		add(root.right);
	}
	while (root_k == 0 && nk < 8) {
		int best = -1;
		float best_area = -1.0f;
		for (int i = 0; i < nk; i++) {
			if (!interior[i]) continue;
			const float dx = kb[i].hi.x - kb[i].lo.x, dy = kb[i].hi.y - kb[i].lo.y, dz = kb[i].hi.z - kb[i].lo.z;
			const float area = dx * dy + dy * dz + dz * dx;
			if (area > best_area) { best_area = area; best = i; }
		}
		if (best < 0) break;
		const BinNode open = bin[kids[best]];
		// the opened child's place is taken by its left child, the right child is appended
		const int keep = nk;
		nk = best;
		add(open.left);
		nk = keep;
		add(open.right);
	}
	// ---- node box and cell sizes
	Box4 box = kb[0];
	for (int i = 1; i < nk; i++) box = box_union(box, kb[i]);
	WideNode nd;
	{
		uint4 *z = reinterpret_cast<uint4 *>(&nd);
		for (int i = 0; i < 8; i++) z[i] = make_uint4(0, 0, 0, 0);
	}
	const float blo[3] = {box.lo.x, box.lo.y, box.lo.z}, bhi[3] = {box.hi.x, box.hi.y, box.hi.z};
	double scale[3];
	for (int a = 0; a < 3; a++) {
		nd.org[a] = blo[a];
		const int e = cell_exponent((double)bhi[a] - (double)blo[a]);
		nd.e[a] = (uint8_t)(e + 127);
		scale[a] = ldexp(1.0, e);
	}
	// ---- slots: child i sits best in the slot whose octant direction its centre's offset from the node's centre points along.
	// The assignment that maximises the sum over the children of offset . direction (Ylitie, Karras, Laine, HPG 2017, section 4.2) is
	// found exactly: best[mask] = the best total for the first popcount(mask) children placed in the slots of mask (2048 steps).
	// (Children taking the nearest free slot in turn, far ones first, traced 1-4 % slower.)
	int child_in_slot[8];
	for (int sl = 0; sl < 8; sl++) child_in_slot[sl] = -1;
	{
		float off[8][3];
		for (int i = 0; i < nk; i++) {
			const float clo[3] = {kb[i].lo.x, kb[i].lo.y, kb[i].lo.z}, chi[3] = {kb[i].hi.x, kb[i].hi.y, kb[i].hi.z};
			for (int a = 0; a < 3; a++) off[i][a] = 0.5f * (clo[a] + chi[a]) - 0.5f * (blo[a] + bhi[a]);
		}
		float best[256];
		uint8_t from[256];
		best[0] = 0.0f;
		for (int mask = 1; mask < 256; mask++) {
			const int i = __popc((unsigned)mask) - 1; // the child this step places
			float bv = -FLT_MAX;
			int bs = 0;
			if (i < nk)
				for (int sl = 0; sl < 8; sl++)
					if ((mask >> sl) & 1) {
						const float sc = ((sl & 1) ? off[i][0] : -off[i][0]) + ((sl & 2) ? off[i][1] : -off[i][1]) + ((sl & 4) ? off[i][2] : -off[i][2]);
						const float v = best[mask ^ (1 << sl)] + sc;
						if (v > bv) { bv = v; bs = sl; }
					}
			best[mask] = bv;
			from[mask] = (uint8_t)bs;
		}
		int top = -1;
		for (int mask = 0; mask < 256; mask++)
			if (__popc((unsigned)mask) == nk && (top < 0 || best[mask] > best[top])) top = mask;
		for (int mask = top, i = nk - 1; i >= 0; i--) {
			const int sl = from[mask];
			child_in_slot[sl] = i;
			mask ^= 1 << sl;
		}
	}
	// ---- quantise; count interior children and leaf triangles
	for (int a = 0; a < 3; a++)
		for (int sl = 0; sl < 8; sl++) { nd.qlo[a][sl] = 255; nd.qhi[a][sl] = 0; }
	uint32_t n_interior = 0, n_leaf_tris = 0;
	for (int sl = 0; sl < 8; sl++) {
		const int i = child_in_slot[sl];
		if (i < 0) continue;
		const float clo[3] = {kb[i].lo.x, kb[i].lo.y, kb[i].lo.z}, chi[3] = {kb[i].hi.x, kb[i].hi.y, kb[i].hi.z};
		for (int a = 0; a < 3; a++) {
			double lo = floor(((double)clo[a] - (double)nd.org[a]) / scale[a]);
			double hi = ceil(((double)chi[a] - (double)nd.org[a]) / scale[a]);
			lo = fmax(0.0, fmin(255.0, lo));
			hi = fmax(0.0, fmin(255.0, hi));
			while ((double)nd.org[a] + lo * scale[a] > (double)clo[a] && lo > 0.0) lo -= 1.0;
			while ((double)nd.org[a] + hi * scale[a] < (double)chi[a] && hi < 255.0) hi += 1.0;
			nd.qlo[a][sl] = (uint8_t)lo;
			nd.qhi[a][sl] = (uint8_t)hi;
		}
		if (interior[i]) {
			nd.imask |= (uint8_t)(1u << sl);
			n_interior++;
		} else {
			n_leaf_tris += kids[i] >= 0 ? bin[kids[i]].last - bin[kids[i]].first + 1u : 1u;
		}
	}
	if (n_leaf_tris > 32u) { atomicAdd(&ctr->overflow, 1u); return; }
	const uint32_t child_base = n_interior ? atomicAdd(&ctr->nodes, n_interior) : 0u;
	const uint32_t tri_base = n_leaf_tris ? atomicAdd(&ctr->tris, n_leaf_tris) : 0u;
	const uint32_t q_base = n_interior ? atomicAdd(&ctr->queue_out, n_interior) : 0u;
	nd.child_base = child_base;
	nd.tri_base = tri_base;
	uint32_t rel = 0, off = 0;
	for (int sl = 0; sl < 8; sl++) {
		const int i = child_in_slot[sl];
		if (i < 0) continue;
		if (interior[i]) {
			queue_out[q_base + rel] = make_uint2((uint32_t)kids[i], child_base + rel);
			rel++;
		} else {
			const uint32_t first = kids[i] >= 0 ? bin[kids[i]].first : (uint32_t)~kids[i];
			const uint32_t count = kids[i] >= 0 ? bin[kids[i]].last - bin[kids[i]].first + 1u : 1u;
			nd.meta[sl] = (uint8_t)((count << 5) | off);
			for (uint32_t k = 0; k < count; k++) leaf_tri[tri_base + off + k] = sorted_tri[first + k];
			off += count;
		}
	}
	uint4 *dst = reinterpret_cast<uint4 *>(&wide[job.y]);
	const uint4 *src = reinterpret_cast<const uint4 *>(&nd);
	for (int i = 0; i < 8; i++) dst[i] = src[i];
}

__global__ void __launch_bounds__(kBlock) k_gather(const double2 *__restrict__ a_in, const double2 *__restrict__ b_in,
                                                   const uint32_t *__restrict__ tri_face, const uint32_t *__restrict__ face_object,
                                                   const uint32_t *__restrict__ leaf_ref, const uint32_t *__restrict__ extra_tri, uint32_t n, uint32_t m,
                                                   uint32_t stride, double2 *__restrict__ rec_out, uint32_t *__restrict__ leaf_tri, uint32_t *__restrict__ tri_leaf)
{
	const uint32_t li = blockIdx.x * kBlock + threadIdx.x;
	if (li >= m) return;
	const uint32_t ref = leaf_ref[li];
	const uint32_t t = ref < n ? ref : extra_tri[ref - n]; // the triangle's own reference, or an additional piece of it
	leaf_tri[li] = t;
	double2 *rec = rec_out + (size_t)stride * li;
	for (int k = 0; k < 3; k++) {
		rec[k] = a_in[3 * (size_t)t + k];
		rec[3 + k] = b_in[3 * (size_t)t + k];
	}
	const uint32_t face = tri_face[t];
	reinterpret_cast<uint4 *>(rec)[6] = make_uint4(t, face, face_object[face], 0u);
	if (stride > 7) rec[7] = make_double2(0.0, 0.0);
	if (ref < n) tri_leaf[t] = li; // every triangle has exactly one reference of its own
}

template <class T>
hipError_t dalloc(T *&p, size_t count, std::vector<void *> &owned)
{
	void *q = nullptr;
	hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
	if (e == hipSuccess) { p = static_cast<T *>(q); owned.push_back(q); }
	return e;
}

} // namespace

hipError_t build_bvh_device(const double *d_tri_a_fan, const double *d_tri_b_fan, const uint32_t *d_tri_face, const uint32_t *d_face_object,
                            uint32_t n_tris, const Presplit *cut, hipStream_t stream, DeviceBvh &out, std::string &err)
{
	out = DeviceBvh();
	if (n_tris < 2) { err = "build_bvh_device: needs at least two triangles"; return hipErrorInvalidValue; }
	const size_t n_first = cut ? cut->first.size() : 0, n_extra = cut ? cut->extra.size() : 0;
	if ((uint64_t)n_tris + n_extra > 0x7FFFFFFFull) { err = "build_bvh_device: too many references"; return hipErrorInvalidValue; }
	const uint32_t n = n_tris + (uint32_t)n_extra; // references: the leaves of the tree
	auto t_start = std::chrono::steady_clock::now();
	std::vector<void *> tmp; // scratch freed on every exit path
	struct Cleanup {
		std::vector<void *> &v;
		~Cleanup() { for (void *p : v) (void)hipFree(p); }
	} cleanup{tmp};
	const unsigned blocks_n = (n + kBlock - 1) / kBlock;

	Box4 *tri_boxes = nullptr, *node_boxes = nullptr;
	int *bounds = nullptr;
	uint64_t *keys = nullptr, *keys_sorted = nullptr;
	uint32_t *vals = nullptr, *sorted_tri = nullptr;
	BinNode *bin = nullptr;
	int32_t *parent_of_node = nullptr, *parent_of_leaf = nullptr;
	unsigned int *arrived = nullptr;
	float *node_cost = nullptr;
	uint8_t *as_leaf = nullptr, *node_cut = nullptr;
	Counters *ctr = nullptr;
	uint2 *queue[2] = {nullptr, nullptr};
	DB_TRY(dalloc(tri_boxes, n, tmp));
	DB_TRY(dalloc(node_boxes, n, tmp));
	DB_TRY(dalloc(bounds, 6, tmp));
	DB_TRY(dalloc(keys, n, tmp));
	DB_TRY(dalloc(keys_sorted, n, tmp));
	DB_TRY(dalloc(vals, n, tmp));
	DB_TRY(dalloc(sorted_tri, n, tmp));
	DB_TRY(dalloc(bin, n, tmp));
	DB_TRY(dalloc(parent_of_node, n, tmp));
	DB_TRY(dalloc(parent_of_leaf, n, tmp));
	DB_TRY(dalloc(arrived, n, tmp));
	DB_TRY(dalloc(node_cost, 7 * (size_t)n, tmp));
	DB_TRY(dalloc(node_cut, 7 * (size_t)n, tmp));
	DB_TRY(dalloc(as_leaf, n, tmp));
	DB_TRY(dalloc(ctr, 1, tmp));
	DB_TRY(dalloc(queue[0], n, tmp));
	DB_TRY(dalloc(queue[1], n, tmp));

	// ---- boxes, keys, sort
	const int init_bounds[6] = {0x7F7FFFFF, 0x7F7FFFFF, 0x7F7FFFFF, (int)0x80800000, (int)0x80800000, (int)0x80800000}; // +FLT_MAX x3, ordered(-FLT_MAX) x3
	DB_TRY(hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(k_tri_boxes, dim3(std::min((n_tris + kBlock - 1) / kBlock, kStrideBlocks)), dim3(kBlock), 0, stream, (const double2 *)d_tri_a_fan, (const double2 *)d_tri_b_fan, n_tris, tri_boxes, bounds);
	uint32_t *extra_tri = nullptr;
	if (n_first + n_extra) {
		// the pieces' boxes lie inside the boxes just computed: the bounds of the centres stand
		static_assert(sizeof(SplitRef) == 28 && sizeof(RefBox) == 24, "a piece is its triangle and six floats");
		std::vector<RefBox> h_box(n_first + n_extra);
		std::vector<uint32_t> h_tri(n_first + n_extra);
		for (size_t k = 0; k < n_first + n_extra; k++) {
			const SplitRef &r = k < n_first ? cut->first[k] : cut->extra[k - n_first];
			if (r.tri >= n_tris) { err = "build_bvh_device: a piece names a triangle the scene does not have"; return hipErrorInvalidValue; }
			for (int a = 0; a < 3; a++) { h_box[k].lo[a] = r.lo[a]; h_box[k].hi[a] = r.hi[a]; }
			h_tri[k] = r.tri;
		}
		RefBox *d_box = nullptr;
		uint32_t *d_tri_of = nullptr;
		DB_TRY(dalloc(d_box, h_box.size(), tmp));
		DB_TRY(dalloc(d_tri_of, h_tri.size(), tmp));
		DB_TRY(hipMemcpyAsync(d_box, h_box.data(), h_box.size() * sizeof(RefBox), hipMemcpyHostToDevice, stream));
		DB_TRY(hipMemcpyAsync(d_tri_of, h_tri.data(), h_tri.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		if (n_first) hipLaunchKernelGGL(k_place_refs, dim3((unsigned)((n_first + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_box, d_tri_of, 0u, (uint32_t)n_first, tri_boxes);
		if (n_extra) hipLaunchKernelGGL(k_place_refs, dim3((unsigned)((n_extra + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_box + n_first, (const uint32_t *)nullptr, n_tris, (uint32_t)n_extra, tri_boxes);
		DB_TRY(hipStreamSynchronize(stream)); // the host arrays go out of scope
		extra_tri = d_tri_of + n_first;
	}
	hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(kBlock), 0, stream, tri_boxes, n, bounds, keys, vals);
	{
		const uint32_t n_tiles = (n + kSortTile - 1) / kSortTile, m = 256u * n_tiles, scan_tiles = (m + kScanTile - 1) / kScanTile;
		uint32_t *table = nullptr, *tile_sums = nullptr;
		DB_TRY(dalloc(table, m, tmp));
		DB_TRY(dalloc(tile_sums, scan_tiles, tmp));
		uint64_t *kbuf[2] = {keys, keys_sorted};
		uint32_t *vbuf[2] = {vals, sorted_tri};
		for (int pass = 0; pass < 8; pass++) {
			const int shift = 8 * pass, from = pass & 1, to = from ^ 1;
			hipLaunchKernelGGL(k_sort_hist, dim3(n_tiles), dim3(kBlock), 0, stream, kbuf[from], n, shift, table, n_tiles);
			hipLaunchKernelGGL(k_scan_tiles, dim3(scan_tiles), dim3(kBlock), 0, stream, table, m, tile_sums);
			hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(kBlock), 0, stream, tile_sums, scan_tiles);
			hipLaunchKernelGGL(k_scan_apply, dim3(scan_tiles), dim3(kBlock), 0, stream, table, m, tile_sums);
			hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(kBlock), 0, stream, kbuf[from], vbuf[from], n, shift, table, n_tiles, kbuf[to], vbuf[to]);
		}
		DB_TRY(hipGetLastError());
		// eight passes: the sorted pairs are back in the buffers the keys started in
		std::swap(keys, keys_sorted);
		std::swap(vals, sorted_tri);
	}

	// ---- binary tree and its boxes
	DB_TRY(hipMemsetAsync(arrived, 0, sizeof(unsigned int) * n, stream));
	hipLaunchKernelGGL(k_hierarchy, dim3(blocks_n), dim3(kBlock), 0, stream, keys_sorted, (int)n, bin, parent_of_node, parent_of_leaf);
	{
		// the ready lists live in the collapse queues, which are not in use yet; ctr->queue_out is the tail of the list being written
		uint32_t *ready[2] = {reinterpret_cast<uint32_t *>(queue[0]), reinterpret_cast<uint32_t *>(queue[1])};
		DB_TRY(hipMemsetAsync(ctr, 0, sizeof(Counters), stream));
		int rotate = 1;
		if (const char *e = std::getenv("MI355RT_ROTATE")) rotate = std::atoi(e); // experiments
		hipLaunchKernelGGL(k_fit_leaves, dim3(std::min(blocks_n, kStrideBlocks)), dim3(kBlock), 0, stream, parent_of_leaf, (int)n, arrived, ready[0], &ctr->queue_out);
		int cur = 0;
		uint32_t settled = 0;
		for (;;) {
			unsigned int n_ready = 0;
			DB_TRY(hipMemcpyAsync(&n_ready, &ctr->queue_out, sizeof(n_ready), hipMemcpyDeviceToHost, stream));
			DB_TRY(hipStreamSynchronize(stream));
			if (n_ready == 0) break;
			settled += n_ready;
			if (settled > n - 1) { err = "build_bvh_device: the bottom-up pass settled more nodes than the tree has"; return hipErrorUnknown; }
			DB_TRY(hipMemsetAsync(&ctr->queue_out, 0, sizeof(unsigned int), stream));
			const unsigned blocks_r = std::min((n_ready + kBlock - 1) / kBlock, kStrideBlocks);
			hipLaunchKernelGGL(k_fit_round, dim3(blocks_r), dim3(kBlock), 0, stream, bin, parent_of_node, tri_boxes, sorted_tri, ready[cur], n_ready, node_boxes,
			                   node_cost, node_cut, as_leaf, arrived, ready[cur ^ 1], &ctr->queue_out, rotate);
			cur ^= 1;
		}
		DB_TRY(hipGetLastError());
		if (settled != n - 1) { err = "build_bvh_device: the bottom-up pass did not reach every node"; return hipErrorUnknown; }
	}

	// ---- collapse, level by level.  A wide node has at least two children, so there are at most n - 1 of them.
	WideNode *wide = nullptr;
	uint32_t *leaf_tri = nullptr;
	DB_TRY(dalloc(wide, n, tmp));
	DB_TRY(dalloc(leaf_tri, n, tmp));
	Counters h_ctr{1u, 0u, 0u, 0u}; // node 0 = the root
	DB_TRY(hipMemcpyAsync(ctr, &h_ctr, sizeof(h_ctr), hipMemcpyHostToDevice, stream));
	const uint2 root_job = make_uint2(0u, 0u);
	DB_TRY(hipMemcpyAsync(queue[0], &root_job, sizeof(root_job), hipMemcpyHostToDevice, stream));
	uint32_t n_in = 1, depth = 0;
	int cur = 0;
	while (n_in > 0) {
		depth++;
		if (depth > kTraceMaxDepth) { err = "build_bvh_device: tree deeper than the traversal stack (" + std::to_string(kTraceMaxDepth) + " levels)"; return hipErrorUnknown; }
		hipLaunchKernelGGL(k_collapse, dim3((n_in + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, bin, node_boxes, as_leaf, node_cut, tri_boxes, sorted_tri,
		                   queue[cur], n_in, queue[cur ^ 1], wide, leaf_tri, ctr);
		DB_TRY(hipMemcpyAsync(&h_ctr, ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, stream));
		DB_TRY(hipStreamSynchronize(stream));
		if (h_ctr.overflow) { err = "build_bvh_device: a node holds more than 32 leaf triangles"; return hipErrorUnknown; }
		n_in = h_ctr.queue_out;
		DB_TRY(hipMemsetAsync(&ctr->queue_out, 0, sizeof(unsigned int), stream));
		cur ^= 1;
	}
	if (h_ctr.tris != n) { err = "build_bvh_device: leaf order does not cover every triangle"; return hipErrorUnknown; }

	// ---- results: right-sized node array, leaf order, triangle records in leaf order
	const uint32_t n_nodes = h_ctr.nodes;
	std::vector<void *> keep;
	WideNode *nodes_out = nullptr;
	uint32_t *leaf_out = nullptr, *tri_leaf = nullptr;
	double2 *rec_out = nullptr;
	hipError_t e = dalloc(nodes_out, n_nodes, keep);
	if (e == hipSuccess) e = dalloc(leaf_out, n, keep);
	if (e == hipSuccess) e = dalloc(tri_leaf, n_tris, keep);
	out.tri_stride = triangle_record_stride(n);
	if (e == hipSuccess) e = dalloc(rec_out, (size_t)out.tri_stride * n, keep);
	if (e == hipSuccess) e = hipMemcpyAsync(nodes_out, wide, sizeof(WideNode) * n_nodes, hipMemcpyDeviceToDevice, stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_gather, dim3(blocks_n), dim3(kBlock), 0, stream, (const double2 *)d_tri_a_fan, (const double2 *)d_tri_b_fan, d_tri_face, d_face_object,
		                   leaf_tri, extra_tri, n_tris, n, out.tri_stride, rec_out, leaf_out, tri_leaf);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(stream);
	if (e != hipSuccess) {
		for (void *p : keep) (void)hipFree(p);
		err = std::string("build_bvh_device: ") + hipGetErrorString(e);
		return e;
	}
	out.nodes = nodes_out;
	out.leaf_tri = leaf_out;
	out.tri_leaf = tri_leaf;
	out.tri = rec_out;
	out.n_nodes = n_nodes;
	out.n_refs = n;
	out.depth = depth;
	out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	return hipSuccess;
}

} // namespace mi355rt
