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
#include <cfloat>
#include <cstdint>
#include "device_scene.h"
#include "trace.h"

namespace mi355rt {

namespace {

// tunables (overridable at build time for experiments)
#ifndef TRACE_LDS_STACK
#define TRACE_LDS_STACK 32
#endif
#ifndef TRACE_REFILL_MIN
#define TRACE_REFILL_MIN 16
#endif
#ifndef TRACE_NODE_MIN_LANES
#define TRACE_NODE_MIN_LANES 0
#endif
#ifndef TRACE_WAVES_PER_SIMD
#define TRACE_WAVES_PER_SIMD 4
#endif
constexpr int kBlock = 256;
constexpr int kLdsStack = TRACE_LDS_STACK; // entries per lane held in LDS: trees up to that depth + 1
constexpr int kScratchStack = 96;          // deeper trees (bounded by the builder)

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

__device__ __forceinline__ float next_up(float x) // smallest float > x for finite positive/negative x
{
	int i = __float_as_int(x);
	if (x == 0.0f) return 1.4e-45f;
	i += (i >= 0) ? 1 : -1;
	return __int_as_float(i);
}

constexpr int kDone = (int)0x80000000; // no leaf link has this value (first < 2^27 and count <= 16)
constexpr int kRefillMin = TRACE_REFILL_MIN; // idle lanes in a wave before it takes new rays

// Persistent waves: each wave owns a contiguous range of rays and keeps its 64 lanes busy by handing
// idle lanes the next rays of the range (no global work counter, so no same-address atomics).
// Traversal is "while-while" with one postponed leaf per lane: a lane that reaches a leaf parks it and
// keeps descending until it reaches a second leaf or runs out of nodes, so the expensive fp64 triangle
// phase runs with most lanes holding work.
template <bool LDS_STACK, bool ANY>
__global__ void __launch_bounds__(kBlock, TRACE_WAVES_PER_SIMD) k_trace(DeviceGeom g, const rt_ray *__restrict__ rays, uint64_t n,
                                                  rt_hit *__restrict__ hits, double *__restrict__ t64,
                                                  uint8_t *__restrict__ any_out, uint32_t rays_per_wave)
{
#ifdef TRACE_STATS
	// diagnostic build only: wave-level loop trip counts and the lanes active in them, printed by wave 0 lanes
	unsigned long long st_node_iter = 0, st_node_lanes = 0, st_leaf_iter = 0, st_leaf_lanes = 0, st_outer = 0, st_outer_lanes = 0;
#endif
	__shared__ int lds_stack[LDS_STACK ? kLdsStack * kBlock : 1];
	int scratch_stack[LDS_STACK ? 1 : kScratchStack];
	const int lane = threadIdx.x & 63;
	const unsigned long long lt = (1ull << lane) - 1ull;
	// wave w owns the 64-ray chunks w, w + W, w + 2W, ... (W = waves in the grid): neighbouring rays stay
	// together in a wave, and every wave sees a sample of the whole batch, so the work per wave evens out
	const uint64_t wave_id = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / 64);
	uint64_t next = 0; // position in this wave's own sequence of rays
	(void)rays_per_wave;

	bool active = false;
	uint64_t ray_id = 0;
	float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, tmin_f = 0, tfar_f = 0;
	double t_best = 0.0;
	uint32_t best = RT_MISS, best_tri = RT_MISS;
	bool found_any = false;
	int sp = 0, cur = kDone, leaf = 0;

	for (;;) {
		// ---- hand new rays to idle lanes
		const unsigned long long idle = __ballot(!active);
		if (idle != 0 && ((next >> 6) * n_waves + wave_id) * 64 < n) {
			const int nidle = __popcll(idle);
			if (nidle >= kRefillMin) {
				const uint64_t seq = next + (uint64_t)__popcll(idle & lt);
				const uint64_t mine = ((seq >> 6) * n_waves + wave_id) * 64 + (seq & 63);
				if (!active && mine < n) {
					const float4 ra = reinterpret_cast<const float4 *>(rays)[2 * mine];
					const float4 rb = reinterpret_cast<const float4 *>(rays)[2 * mine + 1];
					ray_id = mine;
					ox = ra.x; oy = ra.y; oz = ra.z; tmin_f = ra.w;
					dx = rb.x; dy = rb.y; dz = rb.z; tfar_f = rb.w;
					ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;
					t_best = isinf(rb.w) ? DBL_MAX : (double)rb.w; // t_max; shrinks to the closest accepted t
					best = RT_MISS;
					best_tri = RT_MISS;
					found_any = false;
					sp = 0;
					cur = 0;
					leaf = 0;
					active = true;
				}
				next += (uint64_t)nidle;
			}
		}
		if (__ballot(active) == 0) break;
#ifdef TRACE_STATS
		{ const unsigned long long m_ = __ballot(active); if (lane == __ffsll((long long)__ballot(true)) - 1) st_outer++; st_outer_lanes += active ? 1 : 0; (void)m_; }
#endif
		if (active) {
			// ---- node phase: descend until this lane holds a parked leaf and stands on another, or is out of nodes
			while (cur >= 0) {
#ifdef TRACE_STATS
				if (lane == __ffsll((long long)__ballot(true)) - 1) st_node_iter++; st_node_lanes++;
#endif
				const float4 q0 = g.nodes[4 * (size_t)cur], q1 = g.nodes[4 * (size_t)cur + 1];
				const float4 q2 = g.nodes[4 * (size_t)cur + 2], q3 = g.nodes[4 * (size_t)cur + 3];
				// child 0: lo = q0.xyz, hi = (q0.w, q1.x, q1.y); child 1: lo = (q1.z, q1.w, q2.x), hi = q2.yzw
				float a0 = (q0.x - ox) * ix, b0 = (q0.w - ox) * ix;
				float a1 = (q0.y - oy) * iy, b1 = (q1.x - oy) * iy;
				float a2 = (q0.z - oz) * iz, b2 = (q1.y - oz) * iz;
				const float n0 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), tmin_f));
				const float f0 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), tfar_f));
				a0 = (q1.z - ox) * ix; b0 = (q2.y - ox) * ix;
				a1 = (q1.w - oy) * iy; b1 = (q2.z - oy) * iy;
				a2 = (q2.x - oz) * iz; b2 = (q2.w - oz) * iz;
				const float n1 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), tmin_f));
				const float f1 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), tfar_f));
				// slack: each slab distance carries <= 3 roundings (2^-24 each)
				const bool h0 = n0 <= fmaf(fabsf(f0), 1e-6f, f0);
				const bool h1 = n1 <= fmaf(fabsf(f1), 1e-6f, f1);
				const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
				if (h0 && h1) {
					const bool swap = n1 < n0;
					const int farc = swap ? c0 : c1;
					if (LDS_STACK) lds_stack[sp * kBlock + threadIdx.x] = farc;
					else scratch_stack[sp] = farc;
					sp++;
					cur = swap ? c1 : c0;
				} else if (h0) {
					cur = c0;
				} else if (h1) {
					cur = c1;
				} else if (sp == 0) {
					cur = kDone;
				} else {
					sp--;
					cur = LDS_STACK ? lds_stack[sp * kBlock + threadIdx.x] : scratch_stack[sp];
				}
				if (cur < 0 && cur != kDone && leaf == 0) { // park the first leaf and keep going
					leaf = cur;
					if (sp == 0) cur = kDone;
					else {
						sp--;
						cur = LDS_STACK ? lds_stack[sp * kBlock + threadIdx.x] : scratch_stack[sp];
					}
				}
				if (__ballot(leaf == 0) == 0) break; // every lane still descending already holds a leaf
				if (TRACE_NODE_MIN_LANES > 0 && __popcll(__ballot(true)) < TRACE_NODE_MIN_LANES) break; // too few lanes left
			}
			// ---- leaf phase: the parked leaf, then the leaf this lane stands on (if any)
			if (leaf == 0 && cur < 0 && cur != kDone) {
				leaf = cur;
				if (sp == 0) cur = kDone;
				else {
					sp--;
					cur = LDS_STACK ? lds_stack[sp * kBlock + threadIdx.x] : scratch_stack[sp];
				}
			}
			while (leaf != 0) {
#ifdef TRACE_STATS
				if (lane == __ffsll((long long)__ballot(true)) - 1) st_leaf_iter++; st_leaf_lanes++;
#endif
				const uint32_t c = (uint32_t)~leaf;
				const uint32_t first = c & 0x07FFFFFFu, count = (c >> 27) + 1u;
				RayD r;
				r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz; r.tmin = tmin_f;
				for (uint32_t k = 0; k < count; k++) {
					const uint32_t li = first + k;
					const double2 a0 = g.tri_a[3 * (size_t)li], a1 = g.tri_a[3 * (size_t)li + 1], a2 = g.tri_a[3 * (size_t)li + 2];
					// a0 = (v0x, v0y), a1 = (v0z, nx), a2 = (ny, nz)
					const double t = plane_t(r, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
					if (!(t >= r.tmin && t <= t_best)) continue;
					const double2 b0 = g.tri_b[3 * (size_t)li], b1 = g.tri_b[3 * (size_t)li + 1], b2 = g.tri_b[3 * (size_t)li + 2];
					// b0 = (v1x, v1y), b1 = (v1z, v2x), b2 = (v2y, v2z)
					if (!inside_exact(r, t, a0.x, a0.y, a1.x, b0.x, b0.y, b1.x, b1.y, b2.x, b2.y)) continue;
					if (ANY) { found_any = true; break; }
					const uint32_t tri = g.leaf_tri[li];
					if (t < t_best || best == RT_MISS || tri < best_tri) { // ties in t: lowest fan triangle
						t_best = t;
						best = li;
						best_tri = tri;
						float tf = (float)t;
						if ((double)tf < t) tf = next_up(tf);
						tfar_f = tf;
					}
				}
				leaf = 0;
				if (ANY && found_any) { cur = kDone; break; }
				if (cur < 0 && cur != kDone) {
					leaf = cur;
					if (sp == 0) cur = kDone;
					else {
						sp--;
						cur = LDS_STACK ? lds_stack[sp * kBlock + threadIdx.x] : scratch_stack[sp];
					}
				}
			}
			// ---- finished: write the result, free the lane
			if (cur == kDone) {
				if (ANY) {
					any_out[ray_id] = found_any ? 1 : 0;
				} else {
					rt_hit h;
					if (best != RT_MISS) {
						const uint32_t face = g.tri_face[best_tri];
						h.t = (float)t_best;
						h.prim = face;
						h.tri = best_tri;
						h.object = g.face_object[face];
					} else {
						h.t = __int_as_float(0x7f800000);
						h.prim = RT_MISS;
						h.tri = RT_MISS;
						h.object = RT_MISS;
					}
					reinterpret_cast<uint4 *>(hits)[ray_id] = *reinterpret_cast<uint4 *>(&h);
#ifndef TRACE_STATS
					if (t64) t64[ray_id] = best != RT_MISS ? t_best : (double)__int_as_float(0x7f800000);
#endif
				}
				active = false;
			}
		}
	}
#ifdef TRACE_STATS
	if (t64 && n > 1000000) { // every lane adds its own counts (t64 doubles as the stats sink in this build)
		unsigned long long *sink = reinterpret_cast<unsigned long long *>(t64);
		atomicAdd(&sink[0], st_node_iter); atomicAdd(&sink[1], st_node_lanes);
		atomicAdd(&sink[2], st_leaf_iter); atomicAdd(&sink[3], st_leaf_lanes);
		atomicAdd(&sink[4], st_outer); atomicAdd(&sink[5], st_outer_lanes);
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

hipError_t launch_trace_closest(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits, double *d_t64, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	if (g.depth > (uint32_t)kScratchStack) return hipErrorInvalidValue;
	const TraceGrid tg = trace_grid(n);
	if (g.depth <= (uint32_t)kLdsStack)
		hipLaunchKernelGGL((k_trace<true, false>), dim3(tg.blocks), dim3(kBlock), 0, stream, g, d_rays, n, d_hits, d_t64, (uint8_t *)nullptr, tg.rays_per_wave);
	else
		hipLaunchKernelGGL((k_trace<false, false>), dim3(tg.blocks), dim3(kBlock), 0, stream, g, d_rays, n, d_hits, d_t64, (uint8_t *)nullptr, tg.rays_per_wave);
	return hipGetLastError();
}

hipError_t launch_trace_any(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, uint8_t *d_out, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	if (g.depth > (uint32_t)kScratchStack) return hipErrorInvalidValue;
	const TraceGrid tg = trace_grid(n);
	if (g.depth <= (uint32_t)kLdsStack)
		hipLaunchKernelGGL((k_trace<true, true>), dim3(tg.blocks), dim3(kBlock), 0, stream, g, d_rays, n, (rt_hit *)nullptr, (double *)nullptr, d_out, tg.rays_per_wave);
	else
		hipLaunchKernelGGL((k_trace<false, true>), dim3(tg.blocks), dim3(kBlock), 0, stream, g, d_rays, n, (rt_hit *)nullptr, (double *)nullptr, d_out, tg.rays_per_wave);
	return hipGetLastError();
}

} // namespace mi355rt
