// Calibration: how fast can the chip serve random 128-byte-aligned reads (the access pattern of one BVH node
// per lane), and how many memory-side read requests does the L2 count per line?  Each lane reads BYTES bytes
// (64 or 128) from a pseudo-random line of a table of the given size, ITER times, dependent on nothing but
// its own counter, so the latency is hidden by occupancy alone (the same way the trace kernel hides it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

template <int QUADS, bool DEPENDENT>
__global__ void __launch_bounds__(256) k_lines(const float4 *__restrict__ table, uint64_t n_lines, int iter, float *__restrict__ out)
{
	const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t s = mix64(gid + 1);
	float acc = 0.0f;
	for (int i = 0; i < iter; i++) {
		s = mix64(s + 0x9E3779B97F4A7C15ULL);
		const uint64_t line = s % n_lines;
		const float4 *p = table + line * 8;
#pragma unroll
		for (int q = 0; q < QUADS; q++) {
			const float4 v = p[q];
			acc += v.x + v.y + v.z + v.w;
		}
		if (DEPENDENT) s ^= (uint64_t)__float_as_uint(acc); // next address depends on the data: a pointer chase
	}
	if (acc == 12345.678f) out[gid] = acc;
}

template <int QUADS, bool DEP>
void run(const char *name, const float4 *table, uint64_t n_lines, int blocks, int iter, float *out)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	k_lines<QUADS, DEP><<<blocks, 256>>>(table, n_lines, iter, out);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	k_lines<QUADS, DEP><<<blocks, 256>>>(table, n_lines, iter, out);
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	const double lines = (double)blocks * 256 * iter;
	printf("%-28s table %7.1f MB  %6.2f ms  %7.2f G lines/s  %7.2f TB/s (at %d B per line)\n", name, n_lines * 128 / 1e6, ms,
	       lines / ms / 1e6, lines * QUADS * 16 / ms / 1e9, QUADS * 16);
}

int main(int argc, char **argv)
{
	const double mb = argc > 1 ? atof(argv[1]) : 1400.0;
	const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
	const int iter = argc > 3 ? atoi(argv[3]) : 256;
	const uint64_t n_lines = (uint64_t)(mb * 1e6 / 128);
	float4 *table; float *out;
	CHECK(hipMalloc(&table, n_lines * 128));
	CHECK(hipMemset(table, 0, n_lines * 128));
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	run<8, false>("128 B per lane, independent", table, n_lines, blocks, iter, out);
	run<4, false>("64 B per lane, independent", table, n_lines, blocks, iter, out);
	run<8, true>("128 B per lane, chase", table, n_lines, blocks, iter, out);
	run<4, true>("64 B per lane, chase", table, n_lines, blocks, iter, out);
	return 0;
}
