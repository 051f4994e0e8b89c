// Closest-hit and any-hit BVH traversal for gfx950 (wave64).
//
// One ray per lane.  Box tests run in fp32 against child boxes that were rounded outward on the
// host, with a relative slack that covers the fp32 slab arithmetic, so they never cull a triangle
// the exact test would accept.  The triangle test is the reference's own arithmetic
// (ConvexPolygon::HitTriangle, reference object.cpp:238-253) in fp64 without FMA contraction:
// MI355X issues fp64 vector ops at half the fp32 rate, so exactness costs ALU, not correctness.
// The traversal stack holds at most `depth` links; it lives in LDS laid out [entry][lane]
// (conflict-free ds_read/ds_write_b32) when the tree is shallow enough, else in per-lane scratch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include "device_scene.h"
#include "trace.h"

namespace mi355rt {

namespace {

// tunables (overridable at build time for experiments)
#ifndef TRACE_LDS_STACK
#define TRACE_LDS_STACK 16
#endif
// Two thresholds steer a wave, both launch arguments (they sit in scalar registers): the number of idle lanes at which it takes new
// rays, and the number of descending lanes below which it leaves the node phase for the triangle phase.  What is best depends on
// how many node steps a ray needs: a ray of the Cornell box needs 1.7, rays end in bulk, and late refills / early exits win (16 / 62);
// a ray inside a 490 k-triangle hall needs 13, and 8 / 40 is 30 % faster there (and slower on the box; 8 / 56 was the best before the
// waves took their rays from work counters: 6.11 against 6.41 Grays/s now); a soup that waits for memory wants 8 / 48 (62 costs it
// 25 %).
// A third one says how the 64-ray chunks of a launch are dealt to the eight XCDs (each has an L2 of its own): one contiguous eighth
// each (kXcdContiguous: a caller that has ordered its rays in space, and the library's own ordering by origin cell, keep one part of
// the scene per L2), or blocks of 2^xcd_block_log chunks in turn.  An integrator's rays come in pixel order, and an eighth of the
// image can cost twice what another costs (sky against a colonnade): with the waves' shares fixed at launch, the kernel then lasts
// as long as its most expensive eighth.  Dealt out in blocks of 32 chunks the eighths cost the same and neighbouring pixels still
// share an L2 (the hall's render rays: +6 % when shuffled at random, which gives up the neighbourhood as well).
struct TraceTuning { uint32_t refill_min, node_min_lanes, xcd_block_log; };
constexpr uint32_t kXcdContiguous = 0xFFu;
// the second batch of a launch (rays == nullptr: none)
struct TraceSecond { const rt_ray *rays; uint64_t n_bound; rt_hit *hits; const uint32_t *n_dev; };
#ifndef TRACE_COOP_LEAF
#define TRACE_COOP_LEAF 1
#endif
#ifndef TRACE_WAVES_PER_SIMD
#define TRACE_WAVES_PER_SIMD 4
#endif
constexpr int kBlock = 256;
constexpr int kLdsStack = TRACE_LDS_STACK; // stack entries (8 bytes) per lane held in LDS: one per tree level
constexpr int kScratchStack = (int)kTraceMaxDepth; // deeper trees (the builders refuse what does not fit)

struct RayD {
	double ox, oy, oz, dx, dy, dz, tmin;
}; // the float ray widened to double: the same values the reference computes with

// The reference's plane distance: t = dot(v0 - o, n) / dot(d, n)   (object.cpp:244)
__device__ __forceinline__ double plane_t(const RayD &r, double v0x, double v0y, double v0z, double nx, double ny, double nz)
{
#pragma clang fp contract(off)
	double ax = v0x - r.ox, ay = v0y - r.oy, az = v0z - r.oz;
	double num = ax * nx + ay * ny + az * nz;
	double den = r.dx * nx + r.dy * ny + r.dz * nz;
	return num / den;
}

// The reference's inside test (object.cpp:248-253): p = o + t d; three cross products; strict > 0.
__device__ __forceinline__ bool inside_exact(const RayD &r, double t, double v0x, double v0y, double v0z, double v1x,
                                             double v1y, double v1z, double v2x, double v2y, double v2z)
{
#pragma clang fp contract(off)
	double px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;
	// result0 = cross(v1 - v0, p - v1)
	double ax = v1x - v0x, ay = v1y - v0y, az = v1z - v0z;
	double bx = px - v1x, by = py - v1y, bz = pz - v1z;
	double r0x = ay * bz - az * by, r0y = -(ax * bz - az * bx), r0z = ax * by - ay * bx;
	// result1 = cross(v2 - v1, p - v2)
	ax = v2x - v1x; ay = v2y - v1y; az = v2z - v1z;
	bx = px - v2x; by = py - v2y; bz = pz - v2z;
	double r1x = ay * bz - az * by, r1y = -(ax * bz - az * bx), r1z = ax * by - ay * bx;
	double d01 = r0x * r1x + r0y * r1y + r0z * r1z;
	if (!(d01 > 0.0)) return false;
	// result2 = cross(v0 - v2, p - v0)
	ax = v0x - v2x; ay = v0y - v2y; az = v0z - v2z;
	bx = px - v0x; by = py - v0y; bz = pz - v0z;
	double r2x = ay * bz - az * by, r2y = -(ax * bz - az * bx), r2z = ax * by - ay * bx;
	double d12 = r1x * r2x + r1y * r2y + r1z * r2z;
	return d12 > 0.0;
}

// The optional watertight test (rt_set_triangle_test; no counterpart in the reference): Woop, Benthin, Wald, "Watertight
// Ray/Triangle Intersection" (JCGT 2013) in fp64 without FMA contraction.  The vertices are translated to the ray origin, the axes
// permuted so that the direction's largest component is z, the direction sheared onto +z, and the three 2D edge functions
// evaluated.  The edge two triangles share gives both the same two products, subtracted the other way round: exactly opposite
// values, so no ray slips between them.  Edges and vertices hit, nothing is culled by orientation, t is the sheared depth.
__device__ __forceinline__ bool watertight_t(const RayD &r, double tmax, double v0x, double v0y, double v0z, double v1x, double v1y,
                                             double v1z, double v2x, double v2y, double v2z, double &t_out)
{
#pragma clang fp contract(off)
	const double ax = fabs(r.dx), ay = fabs(r.dy), az = fabs(r.dz);
	const int kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
	const double dkz = kz == 0 ? r.dx : (kz == 1 ? r.dy : r.dz);
	int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
	if (dkz < 0.0) { const int tmp = kx; kx = ky; ky = tmp; }
#define PICK(k, x, y, z) ((k) == 0 ? (x) : ((k) == 1 ? (y) : (z)))
	const double dkx = PICK(kx, r.dx, r.dy, r.dz), dky = PICK(ky, r.dx, r.dy, r.dz);
	const double sx = dkx / dkz, sy = dky / dkz, sz = 1.0 / dkz;
	const double a0 = v0x - r.ox, a1 = v0y - r.oy, a2 = v0z - r.oz;
	const double b0 = v1x - r.ox, b1 = v1y - r.oy, b2 = v1z - r.oz;
	const double c0 = v2x - r.ox, c1 = v2y - r.oy, c2 = v2z - r.oz;
	const double akz = PICK(kz, a0, a1, a2), bkz = PICK(kz, b0, b1, b2), ckz = PICK(kz, c0, c1, c2);
	const double Ax = PICK(kx, a0, a1, a2) - sx * akz, Ay = PICK(ky, a0, a1, a2) - sy * akz;
	const double Bx = PICK(kx, b0, b1, b2) - sx * bkz, By = PICK(ky, b0, b1, b2) - sy * bkz;
	const double Cx = PICK(kx, c0, c1, c2) - sx * ckz, Cy = PICK(ky, c0, c1, c2) - sy * ckz;
#undef PICK
	const double U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
	if ((U < 0.0 || V < 0.0 || W < 0.0) && (U > 0.0 || V > 0.0 || W > 0.0)) return false;
	const double det = U + V + W;
	if (det == 0.0) return false;
	const double Az = sz * akz, Bz = sz * bkz, Cz = sz * ckz;
	const double t = (U * Az + V * Bz + W * Cz) / det;
	if (!(t >= r.tmin && t <= tmax)) return false;
	t_out = t;
	return true;
}

// one (ray, triangle) test in either mode: the hit distance when the triangle is hit inside [tmin, tmax]
template <bool WT>
__device__ __forceinline__ bool triangle_test(const RayD &r, double tmax, const double2 &a0, const double2 &a1, const double2 &a2,
                                              const double2 &b0, const double2 &b1, const double2 &b2, double &t_out)
{
	// a0 = (v0x, v0y), a1 = (v0z, nx), a2 = (ny, nz); b0 = (v1x, v1y), b1 = (v1z, v2x), b2 = (v2y, v2z)
	if (WT) return watertight_t(r, tmax, a0.x, a0.y, a1.x, b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, t_out);
	const double t = plane_t(r, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
	// the second half of the record stays in the batch of loads that brought the first: nine in ten plane distances lie in range and
	// need it, and left alone the compiler requests it only behind the range check, a second round trip to the cache
	asm volatile("" : : "v"(b0.x), "v"(b0.y), "v"(b1.x), "v"(b1.y), "v"(b2.x), "v"(b2.y));
	if (!(t >= r.tmin && t <= tmax)) return false;
	if (!inside_exact(r, t, a0.x, a0.y, a1.x, b0.x, b0.y, b1.x, b1.y, b2.x, b2.y)) return false;
	t_out = t;
	return true;
}

__device__ __forceinline__ float next_up(float x) // smallest float > x for finite positive/negative x
{
	int i = __float_as_int(x);
	if (x == 0.0f) return 1.4e-45f;
	i += (i >= 0) ? 1 : -1;
	return __int_as_float(i);
}

constexpr float kBigInv = 1.2089258196146292e24f;   // 2^80: stands in for 1/d on an axis the ray does not move along
constexpr float kTinyDir = 8.673617379884035e-19f;  // 2^-60, relative to the largest component

// bit p of the result = bit (p ^ oct) of x, for 8-bit x: children sit in slots whose bits say on which side of
// the node centre they lie; after this permutation the lowest set bit is the child nearest along the ray
__device__ __forceinline__ uint32_t permute_by_octant(uint32_t x, uint32_t oct)
{
	uint32_t s1 = ((x & 0x55u) << 1) | ((x >> 1) & 0x55u);
	x = (oct & 1u) ? s1 : x;
	uint32_t s2 = ((x & 0x33u) << 2) | ((x >> 2) & 0x33u);
	x = (oct & 2u) ? s2 : x;
	uint32_t s4 = ((x & 0x0Fu) << 4) | ((x >> 4) & 0x0Fu);
	x = (oct & 4u) ? s4 : x;
	return x;
}

// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive moves: four shifted adds inside each row of 16
// lanes, then the last lane of a row broadcast into the following rows (row_bcast:15 to rows 1 and 3, row_bcast:31 to rows 2 and 3).
// Seven vector instructions and no LDS traffic, against six dependent ds_bpermute round trips for a shuffle-based scan.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
	return (uint32_t)x;
}

// the same for the maximum of unsigned values (0 is the identity: lanes that a step does not reach keep their value)
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v)
{
	int x = (int)v;
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
	x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
	return (uint32_t)x;
}

// Triangle counts of four leaf children as four bytes: meta holds (count << 5 | offset) per child, hits4 says which of the four the
// ray's box test hit (bit i -> byte i: multiplying by 1 + 2^7 + 2^14 + 2^21 puts bit i at position 8 i, among others that the mask drops)
__device__ __forceinline__ uint32_t child_counts(uint32_t meta4, uint32_t hits4)
{
	const uint32_t sel = ((hits4 & 0xFu) * 0x00204081u) & 0x01010101u;
	return ((meta4 >> 5) & 0x07070707u) & (sel * 7u);
}
__device__ __forceinline__ uint32_t byte_sum(uint32_t c) { return (c * 0x01010101u) >> 24; } // sums stay below 256: at most 4 x 7

__device__ __forceinline__ float byte_to_float(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); } // v_cvt_f32_ubyteK

// Persistent waves: each wave owns an interleaved share of the 64-ray chunks and keeps its 64 lanes busy by
// handing idle lanes the next rays of its share (no global work counter, so no same-address atomics).
//
// The tree is 8 wide with quantised child boxes, one 128-byte line per node.  A lane's traversal state is a
// "group": the children of one node that are still to be visited (base index + a bit mask already ordered
// front to back), so the stack holds at most one entry per tree level.  Leaf children that the ray's box test
// hits are kept as a parked group (triangle base + per-child counts + hit mask) while the lane goes on
// descending; the fp64 triangle phase runs when a lane holds two parked groups or has run out of nodes,
// so that it runs with most lanes holding work.
template <bool LDS_STACK, bool ANY, bool WT, bool RESIDENT>
__global__ void __launch_bounds__(kBlock, TRACE_WAVES_PER_SIMD) k_trace(DeviceGeom g, TraceTuning tune, const rt_ray *__restrict__ rays, uint64_t n_bound,
                                                  rt_hit *__restrict__ hits, double *__restrict__ t64,
                                                  uint8_t *__restrict__ any_out, const uint32_t *__restrict__ n_dev, uint32_t *__restrict__ work, TraceSecond second)
{
	// A launch can take a second batch of rays with its own results (closest hit only): an integrator's connection rays of one
	// bounce and its path rays of the next are ready at the same time, and one launch over both has one tail where two have two.
	// The chunks of the second batch follow those of the first; `n` counts rays as if both were padded to whole chunks.
	const uint64_t n_a = n_dev ? (uint64_t)*n_dev : n_bound;
	const uint64_t n_b = second.rays ? (second.n_dev ? (uint64_t)*second.n_dev : second.n_bound) : 0;
	const uint64_t chunks_a = (n_a + 63) >> 6;
	const uint64_t n = second.rays ? (chunks_a + ((n_b + 63) >> 6)) * 64 : n_a;
#ifdef TRACE_STATS
	// diagnostic build only: per-lane trip counts of the three loops (summed into the t64 buffer at the end)
	unsigned st_ray_steps = 0, st_ray_outer = 0; // node steps and passes of the main loop of the lane's current ray
	unsigned long long st_node_iter = 0, st_node_lanes = 0, st_leaf_iter = 0, st_leaf_lanes = 0, st_outer = 0, st_outer_lanes = 0, st_in_iter = 0, st_in_lanes = 0;
#define TRACE_COUNT(iter, lanes) do { if (lane == __ffsll((long long)__ballot(true)) - 1) iter++; lanes++; } while (0)
	// shader-clock time per section of the main loop, accumulated by lane 0 of every wave (sections end where the next begins)
	long long st_clk[6] = {0, 0, 0, 0, 0, 0}, st_t0 = clock64();
	const unsigned long long st_wall0 = wall_clock64();
#define TRACE_LAP(k) do { const long long now_ = clock64(); st_clk[k] += now_ - st_t0; st_t0 = now_; } while (0)
#else
#define TRACE_COUNT(iter, lanes) do { } while (0)
#define TRACE_LAP(k) do { } while (0)
#endif
#ifdef TRACE_ENDTIMES
	const unsigned long long et_wall0 = wall_clock64();
#endif
#if TRACE_COOP_LEAF
	__shared__ uint32_t coop_owner[kBlock / 64][64];
	__shared__ unsigned long long coop_t[kBlock / 64][64], coop_id[kBlock / 64][64];
#endif
	__shared__ uint32_t lds_base[LDS_STACK ? kLdsStack * kBlock : 1];
	__shared__ uint32_t lds_mask[LDS_STACK ? kLdsStack * kBlock : 1];
	uint32_t scr_base[LDS_STACK ? 1 : kScratchStack], scr_mask[LDS_STACK ? 1 : kScratchStack];
	const int lane = threadIdx.x & 63;
	const unsigned long long lt = (1ull << lane) - 1ull;
	// Workgroups go to the eight XCDs round robin (workgroup b runs on XCD b mod 8), and every XCD has an L2 of
	// its own.  Each XCD therefore takes one share of the 64-ray chunks -- a contiguous eighth, or blocks dealt in turn (TraceTuning)
	// -- and its waves walk through that share together: when the caller has ordered the rays spatially, the waves that share an L2
	// work in the same part of the tree at the same time.  Inside the share a wave takes its chunks one at a time: from the work
	// counter of its group of workgroups (`work` set: chunk k * groups + group for the k the counter hands out), or, without
	// counters, the fixed sequence wave, wave + W, ...
	const bool by_xcd = (gridDim.x & 7u) == 0;
	const uint64_t n_chunks = (n + 63) >> 6;
	const uint32_t wgs = by_xcd ? gridDim.x >> 3 : gridDim.x, wg = by_xcd ? blockIdx.x >> 3 : blockIdx.x; // workgroups of this share, and which one this is
	const uint64_t n_waves = (uint64_t)wgs * (kBlock / 64);
	const uint64_t wave_id = (uint64_t)wg * (kBlock / 64) + (threadIdx.x >> 6);
	const bool dealt = by_xcd && tune.xcd_block_log != kXcdContiguous;
	const uint64_t chunk_lo = by_xcd && !dealt ? n_chunks * (blockIdx.x & 7u) / 8 : 0;
	const uint64_t chunk_hi = by_xcd && !dealt ? n_chunks * ((blockIdx.x & 7u) + 1) / 8 : n_chunks;
	// the v-th chunk of this XCD's share, v = 0, 1, ...; a value >= chunk_hi says there is none (and none for any larger v)
	const uint32_t xlog = dealt ? tune.xcd_block_log : 0u;
	const uint64_t xcd = by_xcd ? (blockIdx.x & 7u) : 0u;
	auto chunk_of = [&](uint64_t v) -> uint64_t {
		return dealt ? ((((v >> xlog) << 3) + xcd) << xlog) + (v & ((1ull << xlog) - 1ull)) : chunk_lo + v;
	};
	const uint32_t n_groups = wgs < (uint32_t)kWorkGroups ? wgs : (uint32_t)kWorkGroups, group = wg % n_groups;
	uint32_t *const counter = work ? work + ((size_t)xcd * kWorkGroups + group) * kWorkStride : nullptr;
	uint64_t taken = 0;        // chunks this wave has taken so far (without counters: its position in the fixed sequence)
	uint64_t cur_chunk = 0;    // the chunk the wave is handing out
	uint32_t cur_pos = 64;     // rays of it already handed out (64: none left)
	bool out_of_work = false;
	uint32_t k_ahead = 0;      // lane 0: the counter's answer for the chunk after the current one
	if (counter && lane == 0) k_ahead = atomicAdd(counter, 1u);

	bool active = false;
	uint64_t ray_id = 0;
	float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, tmin_f = 0, tfar_f = 0, rscale = 1.0f;
	double t_best = 0.0;
	uint32_t best = RT_MISS, best_tri = RT_MISS, oct = 0;
	uint2 best_fo = make_uint2(RT_MISS, RT_MISS); // (face, object) of the best hit, requested when the hit is accepted: nothing waits for it until the result is written
	bool found_any = false;
	int sp = 0;
	uint32_t g_base = 0, g_mask = 0;               // current group: low byte = pending children (ordered), next byte = imask
	uint32_t l0_base = 0, l0_m0 = 0, l0_m1 = 0, l0_hits = 0; // parked leaf group 0: tri_base, meta[0..3], meta[4..7], hit children
	uint32_t l1_base = 0, l1_m0 = 0, l1_m1 = 0, l1_hits = 0; // parked leaf group 1

	for (;;) {
		// ---- hand new rays to idle lanes: what is left of the wave's current chunk, then the start of its next one.  The chunk after
		// that is asked for at the same time (the counter's answer travels with the ray loads and is not needed before the next
		// change of chunk), so a change of chunk costs no round trip of its own.
		{
			const unsigned long long idle = __ballot(!active);
			if (idle != 0 && !out_of_work && __popcll(idle) >= (int)tune.refill_min) {
				const uint32_t nidle = (uint32_t)__popcll(idle), rank = (uint32_t)__popcll(idle & lt);
				const uint32_t rem = 64u - cur_pos; // rays left in the current chunk
				uint64_t chunk_b = ~0ull;
				if (nidle > rem) {
					uint64_t v;
					if (counter) {
						const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_ahead);
						v = (uint64_t)k * n_groups + group;
						if (lane == 0) k_ahead = atomicAdd(counter, 1u);
					} else {
						v = taken * n_waves + wave_id;
					}
					taken++;
					chunk_b = chunk_of(v);
					if (chunk_b >= chunk_hi) { out_of_work = true; chunk_b = ~0ull; }
				}
				uint64_t mine = ~0ull;
				if (rank < rem) mine = cur_chunk * 64 + cur_pos + rank;
				else if (chunk_b != ~0ull) mine = chunk_b * 64 + (rank - rem);
				if (nidle > rem) { cur_chunk = chunk_b; cur_pos = chunk_b != ~0ull ? nidle - rem : 64u; }
				else cur_pos += nidle;
				// which batch the ray belongs to, and where in it (the second batch's flag rides in the top bit of the ray's id)
				const bool in_b = second.rays != nullptr && mine != ~0ull && (mine >> 6) >= chunks_a;
				const uint64_t local = in_b ? mine - chunks_a * 64 : mine;
				if (!active && mine != ~0ull && local < (in_b ? n_b : n_a)) {
					const float4 *src = reinterpret_cast<const float4 *>(in_b ? second.rays : rays);
					const float4 ra = src[2 * local];
					const float4 rb = src[2 * local + 1];
					ray_id = local | (in_b ? 1ull << 63 : 0ull);
					ox = ra.x; oy = ra.y; oz = ra.z; tmin_f = ra.w;
					dx = rb.x; dy = rb.y; dz = rb.z; tfar_f = rb.w;
					// Box tests run on the direction scaled by a power of two so that its largest component lies in
					// [1, 2): t_box = t * rscale.  The scaling is exact and makes the constants below independent of the
					// length of the direction (the reference's connection rays are not normalised).
					// A component that is zero or negligible against the largest one would turn slab distances into NaN or
					// overflow, and NaN drops out of min/max: every box would pass and the ray would walk the whole tree.
					// Such an axis gets a large finite inverse and is flagged (bits 3..5 of oct): the node test then widens
					// the box by one grid cell on that axis, which makes it "origin inside the slab, any t" (the ray moves
					// less than 2^-60 of its travel along that axis, far less than a cell).
					int e2;
					(void)frexpf(fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)), &e2);
					rscale = ldexpf(1.0f, e2 - 1);
					const float sdx = ldexpf(dx, 1 - e2), sdy = ldexpf(dy, 1 - e2), sdz = ldexpf(dz, 1 - e2);
					const bool zx = fabsf(sdx) < kTinyDir, zy = fabsf(sdy) < kTinyDir, zz = fabsf(sdz) < kTinyDir;
					ix = zx ? kBigInv : 1.0f / sdx; iy = zy ? kBigInv : 1.0f / sdy; iz = zz ? kBigInv : 1.0f / sdz;
					oct = (!zx && dx < 0.0f ? 1u : 0u) | (!zy && dy < 0.0f ? 2u : 0u) | (!zz && dz < 0.0f ? 4u : 0u) |
					      (zx ? 8u : 0u) | (zy ? 16u : 0u) | (zz ? 32u : 0u);
					tfar_f = isinf(rb.w) ? rb.w : next_up(rb.w * rscale); // rounded up: a product that underflows must not shrink the range
					t_best = isinf(rb.w) ? DBL_MAX : (double)rb.w; // t_max; shrinks to the closest accepted t
					best = RT_MISS;
					best_tri = RT_MISS;
					found_any = false;
					sp = 0;
					g_base = 0;                       // the root as a one-child group
					g_mask = (1u << (oct & 7u)) | (1u << 8); // slot 0 lands on position 0 ^ octant
					// A NaN anywhere in the ray, a zero direction or an infinite direction component makes every plane distance
					// of the exact test NaN, zero or infinite with a NaN hit point: the reference finds nothing.  The box tests would over-accept such a
					// ray at every node (NaN drops out of min/max), i.e. walk the whole tree, so it is answered here.
					const bool sound = ra.x == ra.x && ra.y == ra.y && ra.z == ra.z && ra.w == ra.w && rb.x == rb.x && rb.y == rb.y &&
					                   rb.z == rb.z && rb.w == rb.w && !(rb.x == 0.0f && rb.y == 0.0f && rb.z == 0.0f) &&
					                   !isinf(rb.x) && !isinf(rb.y) && !isinf(rb.z);
					if (!sound) g_mask = 0;
					l0_hits = 0;
					l1_hits = 0;
					active = true;
				}
			}
		}
		TRACE_LAP(0); // refill
		if (__ballot(active) == 0) {
			if (out_of_work) break;
			continue; // the chunk ended short of the batch's last ray, or on a lane boundary: take the next one
		}
		if (active) {
			TRACE_COUNT(st_outer, st_outer_lanes);
			// ---- node phase
			for (;;) {
				if ((g_mask & 0xFFu) == 0) {
					if (sp == 0) break; // out of nodes
					sp--;
					g_base = LDS_STACK ? lds_base[sp * kBlock + threadIdx.x] : scr_base[sp];
					g_mask = LDS_STACK ? lds_mask[sp * kBlock + threadIdx.x] : scr_mask[sp];
				}
				if (l1_hits != 0) break; // two parked leaf groups: wait for the triangle phase
				// an occlusion query ends with its first hit: its lanes wait with one parked group (descending on, as a closest-hit ray
				// does to keep the triangle phases full, costs them more steps than it saves rounds: +4 % on the Cornell box's shadow rays)
				if (ANY && l0_hits != 0) break;
				// nearest pending child of the group
				TRACE_COUNT(st_node_iter, st_node_lanes);
#ifdef TRACE_STATS
				st_ray_steps++;
#endif
				const uint32_t pos = (uint32_t)__ffs((int)(g_mask & 0xFFu)) - 1u;
				g_mask &= ~(1u << pos);
				const uint32_t slot = pos ^ (oct & 7u);
				const uint32_t rel = (uint32_t)__popc((g_mask >> 8) & ((1u << slot) - 1u));
				const uint32_t node = g_base + rel;
				if ((g_mask & 0xFFu) != 0) { // siblings left: keep the group for later
					if (LDS_STACK) { lds_base[sp * kBlock + threadIdx.x] = g_base; lds_mask[sp * kBlock + threadIdx.x] = g_mask; }
					else { scr_base[sp] = g_base; scr_mask[sp] = g_mask; }
					sp++;
				}
				const uint4 *np = reinterpret_cast<const uint4 *>(g.nodes) + 8 * (size_t)node;
				const uint4 h0 = np[0], h1 = np[1], q0 = np[2], q1 = np[3], q2 = np[4];
				// h0 = org.xyz, e|imask ; h1 = child_base, tri_base, meta[0..3], meta[4..7]
				// q0 = qlo.x[0..7], qlo.y[0..7] ; q1 = qlo.z[0..7], qhi.x[0..7] ; q2 = qhi.y[0..7], qhi.z[0..7]
				const float sx = __uint_as_float((h0.w & 0xFFu) << 23), sy = __uint_as_float(((h0.w >> 8) & 0xFFu) << 23);
				const float sz = __uint_as_float(((h0.w >> 16) & 0xFFu) << 23);
				const uint32_t imask = h0.w >> 24;
				// t(q) = q * (cell * inv) + (org - o) * inv.  cell * inv is exact (power of two); the second term
				// carries two roundings and can cancel against the first, so it is widened by 2^-22 of itself
				const float ax = (__uint_as_float(h0.x) - ox) * ix, ay = (__uint_as_float(h0.y) - oy) * iy, az = (__uint_as_float(h0.z) - oz) * iz;
				const float kx = sx * ix, ky = sy * iy, kz = sz * iz;
				const float wid = 2.384185791015625e-7f; // 2^-22
				float axn = fmaf(-wid, fabsf(ax), ax), axf = fmaf(wid, fabsf(ax), ax);
				float ayn = fmaf(-wid, fabsf(ay), ay), ayf = fmaf(wid, fabsf(ay), ay);
				float azn = fmaf(-wid, fabsf(az), az), azf = fmaf(wid, fabsf(az), az);
				{ // an axis the ray does not move along: widen by one cell (k* is then +cell * kBigInv).  Branch-free on purpose:
				  // a conditional block here splits the node test and costs 4-11 % on every ray (even one the whole wave skips: -16 % on a soup), six FMAs cost nothing measurable
					const float wx = (oct & 8u) ? 1.0f : 0.0f, wy = (oct & 16u) ? 1.0f : 0.0f, wz = (oct & 32u) ? 1.0f : 0.0f;
					axn = fmaf(-wx, kx, axn); axf = fmaf(wx, kx, axf);
					ayn = fmaf(-wy, ky, ayn); ayf = fmaf(wy, ky, ayf);
					azn = fmaf(-wz, kz, azn); azf = fmaf(wz, kz, azf);
				}
				// entry plane is the lower bound when the direction component is positive, else the upper bound
				const bool px = (oct & 1u) == 0, py = (oct & 2u) == 0, pz = (oct & 4u) == 0;
				const uint32_t nx0 = px ? q0.x : q1.z, nx1 = px ? q0.y : q1.w, fx0 = px ? q1.z : q0.x, fx1 = px ? q1.w : q0.y;
				const uint32_t ny0 = py ? q0.z : q2.x, ny1 = py ? q0.w : q2.y, fy0 = py ? q2.x : q0.z, fy1 = py ? q2.y : q0.w;
				const uint32_t nz0 = pz ? q1.x : q2.z, nz1 = pz ? q1.y : q2.w, fz0 = pz ? q2.z : q1.x, fz1 = pz ? q2.w : q1.y;
				uint32_t hitmask = 0;
				const float tmin_s = tmin_f * rscale; // exact, or rounded towards zero when it underflows: the safe side
#pragma unroll
				for (int j = 0; j < 8; j++) {
					const int k = j & 3;
					const float tnx = fmaf(byte_to_float(j < 4 ? nx0 : nx1, k), kx, axn);
					const float tny = fmaf(byte_to_float(j < 4 ? ny0 : ny1, k), ky, ayn);
					const float tnz = fmaf(byte_to_float(j < 4 ? nz0 : nz1, k), kz, azn);
					const float tfx = fmaf(byte_to_float(j < 4 ? fx0 : fx1, k), kx, axf);
					const float tfy = fmaf(byte_to_float(j < 4 ? fy0 : fy1, k), ky, ayf);
					const float tfz = fmaf(byte_to_float(j < 4 ? fz0 : fz1, k), kz, azf);
					const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin_s));
					const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tfar_f));
					// relative slack for the fma roundings; an empty slot has qlo 255 > qhi 0 on every axis and never passes
					if (tn <= fmaf(fabsf(tf), 1e-6f, tf)) hitmask |= 1u << j;
				}
				// an empty slot that passes (possible only for rays with zero direction components, where distances
				// turn into inf/NaN and drop out of min/max) is harmless: its imask bit is 0 and its meta count is 0
				const uint32_t leaf_hits = hitmask & ~imask;
				// a scene that lives in the caches keeps the whole header in the one batch of loads at the top: left alone, the compiler
				// moves the triangle base and the per-child counts into the branch below, a second, dependent round trip to the cache
				// for every node with a leaf hit (+7 % on the Cornell box; on a scene that waits for memory the same change costs 10 %)
				if (RESIDENT) asm volatile("" : : "v"(h1.y), "v"(h1.z), "v"(h1.w));
				g_base = h1.x;
				g_mask = permute_by_octant(hitmask & imask, oct & 7u) | (imask << 8);
				if (leaf_hits != 0) {
					if (l0_hits == 0) { l0_base = h1.y; l0_m0 = h1.z; l0_m1 = h1.w; l0_hits = leaf_hits; }
					else { l1_base = h1.y; l1_m0 = h1.z; l1_m1 = h1.w; l1_hits = leaf_hits; }
				}
				if (__ballot(l0_hits == 0) == 0) break; // every lane still descending already holds triangles to test
				// too few lanes left descending.  Once the wave has no more rays to take, leaving early buys nothing (no lane that is freed
				// will be used again) and costs a pass through the rest of the main loop per node step of the last, long rays.
				if (__popcll(__ballot(true)) < (int)(out_of_work ? 1u : tune.node_min_lanes)) break;
			}
#if TRACE_COOP_LEAF
		}
		TRACE_LAP(1); // node phase
		// ---- triangle phase, shared by the whole wave.  Lanes hold very different numbers of parked triangles (0 to 8 and more), so
		// a per-lane loop runs the fp64 tests with a quarter of the lanes.  Instead every (ray, triangle) pair of the wave gets a lane of
		// its own: pairs are numbered by a prefix sum over the per-lane counts, and lane j tests pair j, which it finds in its owner's
		// registers (see below), with the owner's ray fetched by cross-lane reads.  Hits are merged per owner with LDS atomics: minimum of the
		// bit pattern of t (positive doubles order like their bits), then, among the hits at that t, minimum of (fan triangle << 32 | leaf
		// position), which is the tie rule "lowest fan triangle".  Idle lanes take part as workers.
		{
			const int wv = threadIdx.x >> 6;
			uint32_t n_mine = 0;
			if (active) n_mine = byte_sum(child_counts(l0_m0, l0_hits)) + byte_sum(child_counts(l0_m1, l0_hits >> 4)) +
			                     byte_sum(child_counts(l1_m0, l1_hits)) + byte_sum(child_counts(l1_m1, l1_hits >> 4));
			const uint32_t incl = wave_inclusive_sum(n_mine);
			const uint32_t off = incl - n_mine, total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			if (total != 0) {
				const unsigned long long none_id = ~0ull;
				unsigned long long my_t = (unsigned long long)__double_as_longlong(t_best);
				unsigned long long my_id = best == RT_MISS ? none_id : (((unsigned long long)best_tri << 32) | best);
				if (n_mine) { coop_t[wv][lane] = my_t; coop_id[wv][lane] = my_id; }
				uint32_t carry = 0; // owner (lane + 1) of the pair just before this round's first
				for (uint32_t r0 = 0; r0 < total; r0 += 64) {
					// Pair q belongs to the lane whose run of pairs [off, off + n_mine) contains it.  Every owner whose run starts in this
					// round marks the start, a running maximum over the marks (data-parallel moves, no loop) tells each worker its owner,
					// and the worker works out which of the owner's parked triangles it has from the owner's registers: no lane walks
					// through its own list (that loop ran every wave for its longest list, 100-200 instructions per round).
					coop_owner[wv][lane] = 0u;
					if (n_mine && off >= r0 && off < r0 + 64u) coop_owner[wv][off - r0] = (uint32_t)lane + 1u;
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
					// workers: one pair each
					const bool work = r0 + (uint32_t)lane < total;
					const uint32_t mark = max(wave_inclusive_max(coop_owner[wv][lane]), carry);
					carry = (uint32_t)__builtin_amdgcn_readlane((int)mark, 63);
					const uint32_t o = work ? mark - 1u : (uint32_t)lane;
					uint32_t li = 0;
					{
						const uint32_t k = r0 + (uint32_t)lane - (uint32_t)__shfl((int)off, (int)o); // position in the owner's run
						const uint32_t o_hits = (uint32_t)__shfl((int)(l0_hits | (l1_hits << 8)), (int)o);
						const uint32_t m00 = (uint32_t)__shfl((int)l0_m0, (int)o), m01 = (uint32_t)__shfl((int)l0_m1, (int)o);
						const uint32_t m10 = (uint32_t)__shfl((int)l1_m0, (int)o), m11 = (uint32_t)__shfl((int)l1_m1, (int)o);
						const uint32_t b0 = (uint32_t)__shfl((int)l0_base, (int)o), b1 = (uint32_t)__shfl((int)l1_base, (int)o);
						const uint32_t c00 = child_counts(m00, o_hits), c01 = child_counts(m01, o_hits >> 4);
						const uint32_t c10 = child_counts(m10, o_hits >> 8), c11 = child_counts(m11, o_hits >> 12);
						const uint32_t s1 = byte_sum(c00), s2 = s1 + byte_sum(c01), s3 = s2 + byte_sum(c10);
						// which four children, then which of the four: the first whose running count exceeds the position
						const uint32_t d = (k >= s1 ? 1u : 0u) + (k >= s2 ? 1u : 0u) + (k >= s3 ? 1u : 0u);
						const uint32_t kk = k - (d == 0 ? 0u : (d == 1 ? s1 : (d == 2 ? s2 : s3)));
						const uint32_t c = d == 0 ? c00 : (d == 1 ? c01 : (d == 2 ? c10 : c11));
						const uint32_t m = d == 0 ? m00 : (d == 1 ? m01 : (d == 2 ? m10 : m11));
						const uint32_t p = c * 0x01010101u; // byte i = count of children 0..i
						const uint32_t i = (kk >= (p & 0xFFu) ? 1u : 0u) + (kk >= ((p >> 8) & 0xFFu) ? 1u : 0u) + (kk >= ((p >> 16) & 0xFFu) ? 1u : 0u);
						const uint32_t before = ((p >> (8 * i)) & 0xFFu) - ((c >> (8 * i)) & 0xFFu);
						li = (d < 2 ? b0 : b1) + (((m >> (8 * i)) & 0xFFu) & 31u) + (kk - before);
						if (!work) li = 0;
					}
					TRACE_LAP(2); // counting, prefix sum, finding each worker's pair
					RayD r;
					r.ox = __shfl(ox, (int)o); r.oy = __shfl(oy, (int)o); r.oz = __shfl(oz, (int)o);
					r.dx = __shfl(dx, (int)o); r.dy = __shfl(dy, (int)o); r.dz = __shfl(dz, (int)o);
					r.tmin = __shfl(tmin_f, (int)o);
					bool hit = false;
					unsigned long long bits = 0, id = 0;
					if (work) {
						TRACE_COUNT(st_leaf_iter, st_leaf_lanes);
						const double cur_best = __longlong_as_double((long long)coop_t[wv][o]);
						// the whole record is requested at once: nine in ten plane distances lie in range and need the second half
						const double2 *rec = g.tri + (size_t)g.tri_stride * li; // one 128-byte line per triangle
						const double2 a0 = rec[0], a1 = rec[1], a2 = rec[2], b0 = rec[3], b1 = rec[4], b2 = rec[5];
						// a0 = (v0x, v0y), a1 = (v0z, nx), a2 = (ny, nz); b0 = (v1x, v1y), b1 = (v1z, v2x), b2 = (v2y, v2z)
						const uint32_t fan = ANY ? 0u : reinterpret_cast<const uint32_t *>(rec)[24]; // fan triangle: the tie rule needs it for every hit
						double t;
						if (triangle_test<WT>(r, cur_best, a0, a1, a2, b0, b1, b2, t)) {
							TRACE_COUNT(st_in_iter, st_in_lanes);
							hit = true;
							bits = (unsigned long long)__double_as_longlong(t + 0.0); // -0.0 -> +0.0: keeps bit order = numeric order
							if (RESIDENT) asm volatile("" : : "v"(fan));
							id = ANY ? 0ull : (((unsigned long long)fan << 32) | li);
							__hip_atomic_fetch_min(&coop_t[wv][o], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
					TRACE_LAP(3); // ray fetch, triangle loads, exact tests
					// an owner whose t came down forgets the triangle it held; then the hits at the minimum settle the tie
					if (n_mine && coop_t[wv][lane] != my_t) { my_t = coop_t[wv][lane]; coop_id[wv][lane] = none_id; }
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
					if (hit && bits == coop_t[wv][o]) __hip_atomic_fetch_min(&coop_id[wv][o], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
				}
				if (n_mine) {
					const unsigned long long fin = coop_id[wv][lane];
					if (fin != my_id) { // something was hit that beats what the lane held
						if (ANY) found_any = true;
						else {
							t_best = __longlong_as_double((long long)my_t);
							best = (uint32_t)fin;
							best_tri = (uint32_t)(fin >> 32);
							{ const uint4 m = reinterpret_cast<const uint4 *>(g.tri + (size_t)g.tri_stride * best)[6]; best_fo = make_uint2(m.y, m.z); }
							const double ts = t_best * (double)rscale;
							float tf = (float)ts;
							if ((double)tf < ts) tf = next_up(tf);
							tfar_f = tf;
						}
					}
					l0_hits = 0;
					l1_hits = 0;
				}
			}
		}
		TRACE_LAP(4); // merging the hits, reading them back
		if (active) {
#else
			// ---- triangle phase: both parked groups
			if (l0_hits != 0) {
				RayD r;
				r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz; r.tmin = tmin_f;
				while (l0_hits != 0) {
					const uint32_t c = (uint32_t)__ffs((int)l0_hits) - 1u;
					l0_hits &= l0_hits - 1u;
					const uint32_t meta = ((c < 4 ? l0_m0 : l0_m1) >> (8 * (c & 3))) & 0xFFu;
					const uint32_t first = l0_base + (meta & 31u), count = meta >> 5;
					for (uint32_t k = 0; k < count; k++) {
						TRACE_COUNT(st_leaf_iter, st_leaf_lanes);
						const uint32_t li = first + k;
						const double2 *rec = g.tri + (size_t)g.tri_stride * li;
						const double2 a0 = rec[0], a1 = rec[1], a2 = rec[2], b0 = rec[3], b1 = rec[4], b2 = rec[5];
						double t;
						if (!triangle_test<WT>(r, t_best, a0, a1, a2, b0, b1, b2, t)) continue;
						TRACE_COUNT(st_in_iter, st_in_lanes);
						if (ANY) { found_any = true; break; }
						const uint4 meta = reinterpret_cast<const uint4 *>(rec)[6];
						const uint32_t tri = meta.x;
						if (t < t_best || best == RT_MISS || tri < best_tri) { // ties in t: lowest fan triangle
							t_best = t;
							best = li;
							best_tri = tri;
							best_fo = make_uint2(meta.y, meta.z);
							const double ts = t * (double)rscale;
							float tf = (float)ts;
							if ((double)tf < ts) tf = next_up(tf);
							tfar_f = tf;
						}
					}
					if (ANY && found_any) break;
					if (l0_hits == 0 && l1_hits != 0) { // second group takes the first one's place
						l0_base = l1_base; l0_m0 = l1_m0; l0_m1 = l1_m1; l0_hits = l1_hits;
						l1_hits = 0;
					}
				}
			}
#endif
			// ---- finished: write the result, free the lane
			const bool out_of_nodes = (g_mask & 0xFFu) == 0 && sp == 0;
#ifdef TRACE_STATS
			st_ray_outer++;
#endif
			if ((ANY && found_any) || (out_of_nodes && l0_hits == 0 && l1_hits == 0)) {
#ifdef TRACE_STATS
				if (t64 && n > 1000000) { // histograms over the rays: node steps (powers of two) and passes of the main loop the ray was resident for
					unsigned long long *sink = reinterpret_cast<unsigned long long *>(t64);
					atomicAdd(&sink[32 + (st_ray_steps ? 32 - __clz(st_ray_steps) : 0)], 1ull);
					atomicAdd(&sink[72 + (st_ray_outer ? 32 - __clz(st_ray_outer) : 0)], 1ull);
				}
				st_ray_steps = 0;
				st_ray_outer = 0;
#endif
				if (ANY) {
					any_out[ray_id] = found_any ? 1 : 0;
				} else {
					rt_hit h;
					if (best != RT_MISS) {
						h.t = (float)t_best;
						h.prim = best_fo.x;
						h.tri = best_tri;
						h.object = best_fo.y;
					} else {
						h.t = __int_as_float(0x7f800000);
						h.prim = RT_MISS;
						h.tri = RT_MISS;
						h.object = RT_MISS;
					}
					const bool to_b = (ray_id >> 63) != 0;
					const uint64_t at = ray_id & ~(1ull << 63);
					reinterpret_cast<uint4 *>(to_b ? second.hits : hits)[at] = *reinterpret_cast<uint4 *>(&h);
#if !defined(TRACE_STATS) && !defined(TRACE_ENDTIMES)
					if (t64 && !to_b) t64[at] = best != RT_MISS ? t_best : (double)__int_as_float(0x7f800000);
#endif
				}
				l0_hits = 0;
				l1_hits = 0;
				active = false;
			}
		}
		TRACE_LAP(5); // writing results
	}
#ifdef TRACE_ENDTIMES
	// diagnostic build: when the waves ended, and nothing else (the full statistics slow the kernel down threefold)
	if (t64 && n > 1000000 && lane == 0) {
		const unsigned long long t_end = wall_clock64();
		unsigned long long *sink = reinterpret_cast<unsigned long long *>(t64);
		atomicMin(&sink[14], et_wall0); atomicMax(&sink[15], et_wall0);
		atomicAdd(&sink[16], t_end); atomicMax(&sink[17], t_end); atomicAdd(&sink[18], 1ull);
	}
#endif
#ifdef TRACE_STATS
	const unsigned long long st_wall1 = wall_clock64(); // before the counters below are added up: those atomics are not the kernel's
	if (t64 && n > 1000000) {
		unsigned long long *sink = reinterpret_cast<unsigned long long *>(t64);
		atomicAdd(&sink[0], st_node_iter); atomicAdd(&sink[1], st_node_lanes);
		atomicAdd(&sink[2], st_leaf_iter); atomicAdd(&sink[3], st_leaf_lanes);
		atomicAdd(&sink[4], st_outer); atomicAdd(&sink[5], st_outer_lanes);
		atomicAdd(&sink[6], st_in_iter); atomicAdd(&sink[7], st_in_lanes);
		if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd(&sink[8 + k], (unsigned long long)st_clk[k]);
		// when the waves ended, on the chip-wide 100 MHz counter: first and last start, sum and last of the ends, number of waves
		if (lane == 0) {
			const unsigned long long t_end = st_wall1;
			atomicMin(&sink[14], st_wall0); atomicMax(&sink[15], st_wall0);
			atomicAdd(&sink[16], t_end); atomicMax(&sink[17], t_end); atomicAdd(&sink[18], 1ull);
		}
	}
#endif
}

// grid: enough waves to fill the chip, each with a contiguous share of the rays
struct TraceGrid { unsigned blocks; uint32_t rays_per_wave; };
inline TraceGrid trace_grid(uint64_t n)
{
	const uint64_t waves_per_block = kBlock / 64;
	const uint64_t max_blocks = 256ull * TRACE_WAVES_PER_SIMD; // 256 CUs x resident workgroups (one wave of each per SIMD)
	uint64_t blocks = (n + kBlock - 1) / kBlock;
	if (blocks > max_blocks) blocks = max_blocks;
	const uint64_t waves = blocks * waves_per_block;
	TraceGrid tg;
	tg.blocks = (unsigned)blocks;
	tg.rays_per_wave = (uint32_t)((n + waves - 1) / waves);
	return tg;
}

} // namespace

namespace {
template <bool ANY>
hipError_t launch_trace(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n_first, rt_hit *d_hits, double *d_t64, uint8_t *d_any, hipStream_t stream, const uint32_t *d_n,
                        TraceSecond second = TraceSecond{nullptr, 0, nullptr, nullptr})
{
	if (second.rays && second.n_bound == 0) second = TraceSecond{nullptr, 0, nullptr, nullptr};
	if (n_first == 0 && second.rays) { // only the second batch has rays: it is the launch's first
		d_rays = second.rays; n_first = second.n_bound; d_hits = second.hits; d_n = second.n_dev;
		second = TraceSecond{nullptr, 0, nullptr, nullptr};
	}
	if (n_first == 0) return hipSuccess;
	const uint64_t n = n_first + (second.rays ? second.n_bound + 64 : 0); // sizes the grid: both batches, the first padded to whole chunks
	if (g.depth > (uint32_t)kScratchStack) return hipErrorInvalidValue;
	const TraceGrid tg = trace_grid(n);
	const bool lds = g.depth <= (uint32_t)kLdsStack;
	// Which variant: the one that fetches a node's whole header at once saves a round trip to the cache per node with a leaf hit and
	// costs 12 bytes per lane and node otherwise.  It wins where most visits find a leaf and the tree is served by the caches (the
	// Cornell box +7 %, a 870 k-triangle mesh in the box +4 %) and loses where rays wander through a soup that waits for memory (-10 %
	// at 1M triangles: one visit in ten finds a leaf).  Chosen for scenes small enough to live in the caches (their builders pack the
	// triangle records) and for an integrator's own rays.
	bool resident = g.tri_stride == 7 || g.surface_rays != 0;
	static const char *const force = std::getenv("MI355RT_TRACE_RESIDENT"); // experiments: 0 / 1 overrides the choice
	if (force && (force[0] == '0' || force[0] == '1')) resident = force[0] == '1';
	TraceTuning tune = g.tri_stride == 7 ? TraceTuning{16u, 62u, kXcdContiguous} : (g.surface_rays ? TraceTuning{8u, 40u, kXcdContiguous} : TraceTuning{8u, 48u, kXcdContiguous});
	if (g.surface_rays && !g.ordered_rays) tune.xcd_block_log = 5; // an integrator's rays, in pixel order: blocks of 32 chunks to the XCDs in turn
	static const char *const t_refill = std::getenv("MI355RT_TRACE_REFILL"), *const t_lanes = std::getenv("MI355RT_TRACE_NODE_LANES"), // experiments
	                  *const t_xcd = std::getenv("MI355RT_TRACE_XCD_BLOCK");
	if (t_refill) tune.refill_min = (uint32_t)std::max(1, std::min(64, std::atoi(t_refill)));
	if (t_lanes) tune.node_min_lanes = (uint32_t)std::max(0, std::min(64, std::atoi(t_lanes)));
	if (t_xcd) tune.xcd_block_log = std::atoi(t_xcd) < 0 ? kXcdContiguous : (uint32_t)std::min(20, std::atoi(t_xcd));
	// the launch's set of work counters: zeroed on the stream, behind whichever launch used the set last
	uint32_t *d_work = nullptr;
	int set = -1;
	static const char *const t_dyn = std::getenv("MI355RT_TRACE_DYNAMIC"); // experiments: 0 = fixed shares per wave
	if (g.work && g.work->d_counters && !(t_dyn && t_dyn[0] == '0')) {
		TraceWork &w = *g.work;
		set = w.next;
		w.next = (w.next + 1) % kWorkSets;
		hipError_t e = w.used[set] ? hipStreamWaitEvent(stream, w.done[set], 0) : hipSuccess;
		d_work = w.d_counters + (size_t)set * TraceWork::set_words();
		if (e == hipSuccess) e = hipMemsetAsync(d_work, 0, TraceWork::set_words() * sizeof(uint32_t), stream);
		if (e != hipSuccess) return e;
	}
#define TRACE_LAUNCH(L, W, R) hipLaunchKernelGGL((k_trace<L, ANY, W, R>), dim3(tg.blocks), dim3(kBlock), 0, stream, g, tune, d_rays, n_first, d_hits, d_t64, d_any, d_n, d_work, second)
#define TRACE_LAUNCH_R(L, W) do { if (resident) TRACE_LAUNCH(L, W, true); else TRACE_LAUNCH(L, W, false); } while (0)
	if (g.watertight) { if (lds) TRACE_LAUNCH_R(true, true); else TRACE_LAUNCH_R(false, true); }
	else { if (lds) TRACE_LAUNCH_R(true, false); else TRACE_LAUNCH_R(false, false); }
#undef TRACE_LAUNCH_R
#undef TRACE_LAUNCH
	hipError_t e = hipGetLastError();
	if (set >= 0 && e == hipSuccess) {
		e = hipEventRecord(g.work->done[set], stream);
		g.work->used[set] = e == hipSuccess;
	}
	return e;
}
} // namespace

hipError_t launch_trace_closest(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits, double *d_t64, hipStream_t stream, const uint32_t *d_n)
{
	return launch_trace<false>(g, d_rays, n, d_hits, d_t64, nullptr, stream, d_n);
}

hipError_t launch_trace_closest_two(const DeviceGeom &g, const rt_ray *d_rays_a, uint64_t n_a, rt_hit *d_hits_a, const uint32_t *d_n_a,
                                    const rt_ray *d_rays_b, uint64_t n_b, rt_hit *d_hits_b, const uint32_t *d_n_b, hipStream_t stream)
{
	return launch_trace<false>(g, d_rays_a, n_a, d_hits_a, nullptr, nullptr, stream, d_n_a, TraceSecond{d_rays_b, n_b, d_hits_b, d_n_b});
}

hipError_t launch_trace_any(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, uint8_t *d_out, hipStream_t stream, const uint32_t *d_n)
{
	return launch_trace<true>(g, d_rays, n, nullptr, nullptr, d_out, stream, d_n);
}

// ---- staging that needs no allocation (rt_trace_closest_static)
namespace {
constexpr size_t kStaticRayBytes = kStaticRays * sizeof(rt_ray), kStaticHitBytes = kStaticRays * sizeof(rt_hit), kStaticT64Bytes = kStaticRays * sizeof(double);
// 56 MiB per GPU, held from the moment the library is loaded on that GPU until the process ends
static __device__ __align__(256) unsigned char scratch[kStaticRayBytes + kStaticHitBytes + kStaticT64Bytes];
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices]; // per device: every GPU holds its own copy of the array at its own address
std::mutex g_scratch_mutex;
} // namespace

hipError_t TraceWork::init()
{
	destroy();
	hipError_t e = hipMalloc((void **)&d_counters, (size_t)kWorkSets * set_words() * sizeof(uint32_t));
	for (int i = 0; i < kWorkSets && e == hipSuccess; i++) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
	if (e != hipSuccess) destroy();
	return e;
}

void TraceWork::destroy()
{
	for (int i = 0; i < kWorkSets; i++) {
		if (done[i]) (void)hipEventDestroy(done[i]);
		done[i] = nullptr;
		used[i] = false;
	}
	if (d_counters) (void)hipFree(d_counters);
	d_counters = nullptr;
	next = 0;
}

hipError_t static_stage(int device, StaticStage &out)
{
	if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
	unsigned char *p = nullptr;
	{
		std::lock_guard<std::mutex> lock(g_scratch_mutex);
		if (!g_scratch_ptr[device]) {
			// the array's name is a host-side handle, not a device address: ask the runtime where this device keeps it
			void *sym = nullptr;
			hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
			if (e != hipSuccess) return e;
			if (!sym) return hipErrorInvalidSymbol;
			g_scratch_ptr[device] = static_cast<unsigned char *>(sym);
		}
		p = g_scratch_ptr[device];
	}
	out.rays = reinterpret_cast<rt_ray *>(p);
	out.hits = reinterpret_cast<rt_hit *>(p + kStaticRayBytes);
	out.t64 = reinterpret_cast<double *>(p + kStaticRayBytes + kStaticHitBytes);
	return hipSuccess;
}

} // namespace mi355rt
