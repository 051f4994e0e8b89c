// Bilateral filter of the retouch panel (reference filter.cpp:13-51), one thread per pixel, fp64 like the reference.
// The image is width x height RGBA doubles, row major.  (The reference addresses rows as height - y - 1 on both the
// read and the write side and its window is symmetric, so the flip cancels.)  Window coordinates are clamped to the
// image, and the spatial weight uses the clamped offset, as filter.cpp:24-36 does.  Each colour channel has its own
// range weight; alpha is set to 1.
// Work per pixel: (2 window + 1)^2 taps x 3 channels, one fp64 exp each: the kernel is bound by the fp64 exp rate,
// not by memory (the taps of neighbouring pixels overlap and are served by the vector cache).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "filter.h"

namespace mi355rt {

namespace {

constexpr int kTile = 16;

__global__ void __launch_bounds__(kTile * kTile) k_bilateral(const double4 *__restrict__ img, int width, int height, int window,
                                                            double inv_2sd2, double inv_2sr2, double4 *__restrict__ out)
{
#pragma clang fp contract(off)
	const int x = blockIdx.x * kTile + threadIdx.x, y = blockIdx.y * kTile + threadIdx.y;
	if (x >= width || y >= height) return;
	const double4 c = img[(size_t)y * width + x];
	double sum[3] = {0.0, 0.0, 0.0}, W[3] = {0.0, 0.0, 0.0};
	for (int i = -window; i <= window; i++) {
		int wx = x + i;
		wx = wx < 0 ? 0 : (wx >= width ? width - 1 : wx);
		for (int j = -window; j <= window; j++) {
			// the reference walks wy = y + j over its flipped rows, i.e. row y - j here; the window is symmetric but the
			// floating-point sums are not: keep the reference's order of accumulation
			int wy = (height - 1 - y) + j;
			wy = wy < 0 ? 0 : (wy >= height ? height - 1 : wy);
			const int ry = height - 1 - wy;
			const double4 v = img[(size_t)ry * width + wx];
			const int ddx = wx - x, ddy = wy - (height - 1 - y);
			const double spatial = -(double)(ddx * ddx + ddy * ddy) * inv_2sd2;
			const double cv[3] = {c.x, c.y, c.z}, vv[3] = {v.x, v.y, v.z};
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const double diff = cv[k] - vv[k];
				const double w = exp(spatial + -((diff * diff) * inv_2sr2));
				sum[k] += vv[k] * w;
				W[k] += w;
			}
		}
	}
	out[(size_t)y * width + x] = make_double4(sum[0] / W[0], sum[1] / W[1], sum[2] / W[2], 1.0);
}

} // namespace

hipError_t launch_bilateral(const double *d_rgba, int width, int height, int window, double sigma_d, double sigma_r, double *d_out, hipStream_t stream)
{
	if (width <= 0 || height <= 0) return hipSuccess;
	const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile);
	hipLaunchKernelGGL(k_bilateral, grid, dim3(kTile, kTile), 0, stream, (const double4 *)d_rgba, width, height, window,
	                   1.0 / (2.0 * sigma_d * sigma_d), 1.0 / (2.0 * sigma_r * sigma_r), (double4 *)d_out);
	return hipGetLastError();
}

} // namespace mi355rt
