// see ray_bins.h
#include "ray_bins.h"

namespace mi355rt {
namespace {

constexpr int kBinBlock = 512, kBinPer = 8; // 4096 rays per workgroup: runs of 64 rays per cell on average

// Cell of a ray's origin in a 4 x 4 x 4 grid over the scene's box, which is read off the root of the 8-wide tree: the node's origin
// and power-of-two cell sizes, and the 8-bit grid coordinates of its children's boxes (an unused slot has lower 255 > upper 0).
// Origins outside, NaN included, land in a border cell.
struct Grid { float ox, oy, oz, ix, iy, iz; };
__device__ __forceinline__ void byte_range(uint32_t lo_a, uint32_t lo_b, uint32_t hi_a, uint32_t hi_b, uint32_t &lo, uint32_t &hi)
{
	lo = 255u; hi = 0u;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const uint32_t l = ((k < 4 ? lo_a : lo_b) >> (8 * (k & 3))) & 0xFFu, h = ((k < 4 ? hi_a : hi_b) >> (8 * (k & 3))) & 0xFFu;
		if (l <= h) { lo = min(lo, l); hi = max(hi, h); }
	}
	if (lo > hi) { lo = 0u; hi = 255u; } // no child at all: the node's whole grid
}
__device__ __forceinline__ Grid grid_of(const float4 *root_node)
{
	const uint4 *np = reinterpret_cast<const uint4 *>(root_node);
	const uint4 h0 = np[0], q0 = np[2], q1 = np[3], q2 = np[4];
	// q0 = lower x[0..7], lower y[0..7]; q1 = lower z[0..7], upper x[0..7]; q2 = upper y[0..7], upper z[0..7]
	const float sx = __uint_as_float((h0.w & 0xFFu) << 23), sy = __uint_as_float(((h0.w >> 8) & 0xFFu) << 23), sz = __uint_as_float(((h0.w >> 16) & 0xFFu) << 23);
	uint32_t lx, hx, ly, hy, lz, hz;
	byte_range(q0.x, q0.y, q1.z, q1.w, lx, hx);
	byte_range(q0.z, q0.w, q2.x, q2.y, ly, hy);
	byte_range(q1.x, q1.y, q2.z, q2.w, lz, hz);
	Grid g;
	g.ox = __uint_as_float(h0.x) + (float)lx * sx; g.oy = __uint_as_float(h0.y) + (float)ly * sy; g.oz = __uint_as_float(h0.z) + (float)lz * sz;
	const float ex = (float)(hx - lx) * sx, ey = (float)(hy - ly) * sy, ez = (float)(hz - lz) * sz;
	g.ix = ex > 0.0f ? 4.0f / ex : 0.0f; g.iy = ey > 0.0f ? 4.0f / ey : 0.0f; g.iz = ez > 0.0f ? 4.0f / ez : 0.0f; // a flat scene has one layer of cells
	return g;
}
__device__ __forceinline__ uint32_t axis_cell(float v) { return (uint32_t)min(max((int)v, 0), 3); } // (int)NaN is 0
__device__ __forceinline__ uint32_t spread3(uint32_t v) { return (v & 1u) | ((v & 2u) << 2); }
__device__ __forceinline__ uint32_t cell_of(const Grid &g, float4 o)
{
	return spread3(axis_cell((o.x - g.ox) * g.ix)) | (spread3(axis_cell((o.y - g.oy) * g.iy)) << 1) | (spread3(axis_cell((o.z - g.oz) * g.iz)) << 2);
}

__global__ void __launch_bounds__(kBinBlock) k_bin_count(const float4 *__restrict__ root_node, const float4 *__restrict__ rays, uint32_t n_bound, const uint32_t *__restrict__ n_dev,
                                                         uint32_t *__restrict__ counts)
{
	const uint32_t n = n_dev ? min(n_bound, *n_dev) : n_bound; // the count may live on the device (a queue of an integrator): n_bound sizes the grid
	__shared__ uint32_t h[kRayBins];
	for (int b = threadIdx.x; b < kRayBins; b += kBinBlock) h[b] = 0;
	__syncthreads();
	const Grid g = grid_of(root_node);
	const uint64_t base = (uint64_t)blockIdx.x * (kBinBlock * kBinPer);
#pragma unroll
	for (int j = 0; j < kBinPer; j++) {
		const uint64_t i = base + (uint64_t)j * kBinBlock + threadIdx.x;
		if (i < n) atomicAdd(&h[cell_of(g, rays[2 * i])], 1u);
	}
	__syncthreads();
	for (int b = threadIdx.x; b < kRayBins; b += kBinBlock)
		if (h[b]) atomicAdd(&counts[b], h[b]);
}

// cursor[b] = number of rays in the cells before b
__global__ void __launch_bounds__(kRayBins) k_bin_scan(const uint32_t *__restrict__ counts, uint32_t *__restrict__ cursor)
{
	__shared__ uint32_t s[kRayBins];
	s[threadIdx.x] = counts[threadIdx.x];
	__syncthreads();
	uint32_t before = 0;
	for (int b = 0; b < (int)threadIdx.x; b++) before += s[b];
	cursor[threadIdx.x] = before;
}

__global__ void __launch_bounds__(kBinBlock) k_bin_scatter(const float4 *__restrict__ root_node, const float4 *__restrict__ rays, uint32_t n_bound, const uint32_t *__restrict__ n_dev,
                                                           uint32_t *__restrict__ cursor, float4 *__restrict__ sorted, uint32_t *__restrict__ perm)
{
	const uint32_t n = n_dev ? min(n_bound, *n_dev) : n_bound;
	__shared__ uint32_t h[kRayBins]; // first the workgroup's count per cell, then the first slot of its run in that cell
	for (int b = threadIdx.x; b < kRayBins; b += kBinBlock) h[b] = 0;
	__syncthreads();
	const Grid g = grid_of(root_node);
	const uint64_t base = (uint64_t)blockIdx.x * (kBinBlock * kBinPer);
	float4 ra[kBinPer];
	uint32_t cell[kBinPer], rank[kBinPer];
#pragma unroll
	for (int j = 0; j < kBinPer; j++) {
		const uint64_t i = base + (uint64_t)j * kBinBlock + threadIdx.x;
		if (i < n) {
			ra[j] = rays[2 * i];
			cell[j] = cell_of(g, ra[j]);
			rank[j] = atomicAdd(&h[cell[j]], 1u);
		}
	}
	__syncthreads();
	for (int b = threadIdx.x; b < kRayBins; b += kBinBlock) {
		const uint32_t c = h[b];
		h[b] = c ? atomicAdd(&cursor[b], c) : 0u; // one reservation per workgroup and cell
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < kBinPer; j++) {
		const uint64_t i = base + (uint64_t)j * kBinBlock + threadIdx.x;
		if (i < n) {
			const uint32_t slot = h[cell[j]] + rank[j]; // < n: the cursors start at the cells' offsets and the counts add up to n
			sorted[2 * (uint64_t)slot] = ra[j];
			sorted[2 * (uint64_t)slot + 1] = rays[2 * i + 1];
			perm[i] = slot;
		}
	}
}

__global__ void __launch_bounds__(kBinBlock) k_unbin(const uint32_t *__restrict__ perm, uint32_t n_bound, const uint32_t *__restrict__ n_dev, const uint4 *__restrict__ sorted_hits,
                                                     const double *__restrict__ sorted_t64, uint4 *__restrict__ hits, double *__restrict__ t64)
{
	const uint32_t n = n_dev ? min(n_bound, *n_dev) : n_bound;
	const uint64_t i = (uint64_t)blockIdx.x * kBinBlock + threadIdx.x;
	if (i >= n) return;
	const uint32_t p = perm[i];
	hits[i] = sorted_hits[p];
	if (t64) t64[i] = sorted_t64[p];
}

__global__ void __launch_bounds__(kBinBlock) k_unbin_bytes(const uint32_t *__restrict__ perm, uint32_t n, const uint8_t *__restrict__ sorted_out, uint8_t *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBinBlock + threadIdx.x;
	if (i < n) out[i] = sorted_out[perm[i]];
}

} // namespace

hipError_t bin_rays(const float4 *root_node, const rt_ray *rays, uint32_t n, rt_ray *sorted, uint32_t *perm, uint32_t *counters, hipStream_t stream, const uint32_t *n_dev)
{
	static_assert(sizeof(rt_ray) == 32 && sizeof(rt_hit) == 16, "records are moved as 16-byte pieces");
	if (n == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(counters, 0, 2 * kRayBins * sizeof(uint32_t), stream);
	if (e != hipSuccess) return e;
	const unsigned blocks = (unsigned)(((uint64_t)n + kBinBlock * kBinPer - 1) / (kBinBlock * kBinPer));
	hipLaunchKernelGGL(k_bin_count, dim3(blocks), dim3(kBinBlock), 0, stream, root_node, reinterpret_cast<const float4 *>(rays), n, n_dev, counters);
	hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kRayBins), 0, stream, counters, counters + kRayBins);
	hipLaunchKernelGGL(k_bin_scatter, dim3(blocks), dim3(kBinBlock), 0, stream, root_node, reinterpret_cast<const float4 *>(rays), n, n_dev, counters + kRayBins,
	                   reinterpret_cast<float4 *>(sorted), perm);
	return hipGetLastError();
}

hipError_t unbin_results(const uint32_t *perm, uint32_t n, const rt_hit *sorted_hits, const double *sorted_t64, rt_hit *hits, double *t64, hipStream_t stream, const uint32_t *n_dev)
{
	if (n == 0) return hipSuccess;
	const unsigned blocks = (unsigned)(((uint64_t)n + kBinBlock - 1) / kBinBlock);
	hipLaunchKernelGGL(k_unbin, dim3(blocks), dim3(kBinBlock), 0, stream, perm, n, n_dev, reinterpret_cast<const uint4 *>(sorted_hits), sorted_t64,
	                   reinterpret_cast<uint4 *>(hits), t64);
	return hipGetLastError();
}

hipError_t unbin_bytes(const uint32_t *perm, uint32_t n, const uint8_t *sorted_out, uint8_t *out, hipStream_t stream)
{
	if (n == 0) return hipSuccess;
	const unsigned blocks = (unsigned)(((uint64_t)n + kBinBlock - 1) / kBinBlock);
	hipLaunchKernelGGL(k_unbin_bytes, dim3(blocks), dim3(kBinBlock), 0, stream, perm, n, sorted_out, out);
	return hipGetLastError();
}

} // namespace mi355rt
