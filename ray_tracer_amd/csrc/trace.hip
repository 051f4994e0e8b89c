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

constexpr int kBlock = 256;
constexpr int kLdsStack = 32;     // entries per lane held in LDS: trees up to depth 33
constexpr int kScratchStack = 96; // deeper trees (bounded by the builder)

struct RayD {
	double ox, oy, oz, dx, dy, dz, tmin;
};

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

template <bool LDS_STACK, bool ANY>
__global__ void __launch_bounds__(kBlock) k_trace(DeviceGeom g, const rt_ray *__restrict__ rays, uint64_t n,
                                                  rt_hit *__restrict__ hits, double *__restrict__ t64,
                                                  uint8_t *__restrict__ any_out)
{
	__shared__ int lds_stack[LDS_STACK ? kLdsStack * kBlock : 1];
	int scratch_stack[LDS_STACK ? 1 : kScratchStack];
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;

	const float4 ra = reinterpret_cast<const float4 *>(rays)[2 * i];
	const float4 rb = reinterpret_cast<const float4 *>(rays)[2 * i + 1];
	RayD r;
	r.ox = ra.x; r.oy = ra.y; r.oz = ra.z; r.tmin = ra.w;
	r.dx = rb.x; r.dy = rb.y; r.dz = rb.z;
	double t_best = isinf(rb.w) ? DBL_MAX : (double)rb.w; // t_max; shrinks to the closest accepted t
	uint32_t best = RT_MISS;                               // leaf-order index of the closest triangle
	uint32_t best_tri = RT_MISS;                           // its fan-triangle id (tie rule: lowest wins)

	const float ox = ra.x, oy = ra.y, oz = ra.z;
	const float ix = 1.0f / rb.x, iy = 1.0f / rb.y, iz = 1.0f / rb.z;
	const float tmin_f = ra.w;
	float tfar_f = rb.w; // float bound >= t_best
	bool found_any = false;

	int sp = 0;
	int cur = 0;
	for (;;) {
		if (cur >= 0) {
			const float4 q0 = g.nodes[4 * (size_t)cur], q1 = g.nodes[4 * (size_t)cur + 1];
			const float4 q2 = g.nodes[4 * (size_t)cur + 2], q3 = g.nodes[4 * (size_t)cur + 3];
			// child 0: lo = q0.xyz, hi = (q0.w, q1.x, q1.y); child 1: lo = (q1.z, q1.w, q2.x), hi = q2.yzw
			float a0 = (q0.x - ox) * ix, b0 = (q0.w - ox) * ix;
			float a1 = (q0.y - oy) * iy, b1 = (q1.x - oy) * iy;
			float a2 = (q0.z - oz) * iz, b2 = (q1.y - oz) * iz;
			float n0 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), tmin_f));
			float f0 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), tfar_f));
			a0 = (q1.z - ox) * ix; b0 = (q2.y - ox) * ix;
			a1 = (q1.w - oy) * iy; b1 = (q2.z - oy) * iy;
			a2 = (q2.x - oz) * iz; b2 = (q2.w - oz) * iz;
			float n1 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), tmin_f));
			float f1 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), tfar_f));
			// slack: each slab distance carries <= 3 roundings (2^-24 each)
			const bool h0 = n0 <= fmaf(fabsf(f0), 1e-6f, f0);
			const bool h1 = n1 <= fmaf(fabsf(f1), 1e-6f, f1);
			const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
			if (h0 && h1) {
				const bool swap = n1 < n0;
				const int nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
				if (LDS_STACK) lds_stack[sp * kBlock + threadIdx.x] = farc;
				else scratch_stack[sp] = farc;
				sp++;
				cur = nearc;
				continue;
			} else if (h0) {
				cur = c0;
				continue;
			} else if (h1) {
				cur = c1;
				continue;
			}
		} else {
			const uint32_t c = (uint32_t)~cur;
			const uint32_t first = c & 0x07FFFFFFu, count = (c >> 27) + 1u;
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
				if (t < t_best || best == RT_MISS || tri < best_tri) {
					t_best = t;
					best = li;
					best_tri = tri;
					float tf = (float)t;
					if ((double)tf < t) tf = next_up(tf);
					tfar_f = tf;
				}
			}
			if (ANY && found_any) break;
		}
		if (sp == 0) break;
		sp--;
		cur = LDS_STACK ? lds_stack[sp * kBlock + threadIdx.x] : scratch_stack[sp];
	}

	if (ANY) {
		any_out[i] = found_any ? 1 : 0;
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
		reinterpret_cast<uint4 *>(hits)[i] = *reinterpret_cast<uint4 *>(&h);
		if (t64) t64[i] = best != RT_MISS ? t_best : (double)__int_as_float(0x7f800000);
	}
}

} // namespace

hipError_t launch_trace_closest(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits, double *d_t64, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	if (g.depth > (uint32_t)kScratchStack) return hipErrorInvalidValue;
	const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
	if (g.depth <= (uint32_t)kLdsStack)
		hipLaunchKernelGGL((k_trace<true, false>), dim3(blocks), dim3(kBlock), 0, stream, g, d_rays, n, d_hits, d_t64, (uint8_t *)nullptr);
	else
		hipLaunchKernelGGL((k_trace<false, false>), dim3(blocks), dim3(kBlock), 0, stream, g, d_rays, n, d_hits, d_t64, (uint8_t *)nullptr);
	return hipGetLastError();
}

hipError_t launch_trace_any(const DeviceGeom &g, const rt_ray *d_rays, uint64_t n, uint8_t *d_out, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	if (g.depth > (uint32_t)kScratchStack) return hipErrorInvalidValue;
	const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
	if (g.depth <= (uint32_t)kLdsStack)
		hipLaunchKernelGGL((k_trace<true, true>), dim3(blocks), dim3(kBlock), 0, stream, g, d_rays, n, (rt_hit *)nullptr, (double *)nullptr, d_out);
	else
		hipLaunchKernelGGL((k_trace<false, true>), dim3(blocks), dim3(kBlock), 0, stream, g, d_rays, n, (rt_hit *)nullptr, (double *)nullptr, d_out);
	return hipGetLastError();
}

} // namespace mi355rt
