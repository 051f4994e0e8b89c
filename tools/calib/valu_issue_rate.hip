// Sustained issue rate of vector instructions on one MI355X, in wave-level instructions per second over the whole chip, for the
// instruction mixes of the traversal kernel and at 1, 2, 4 and 8 waves per SIMD: the roof against which a kernel that is bound by
// its own instruction stream is priced (SQ_INSTS_VALU / time / this rate).
//   fma        v_fma_f32 with three register operands, seven independent chains
//   node       the box test of one child of an 8-wide node as the traversal kernel compiles it: 6 v_cvt_f32_ubyteN, 6 v_fma_f32,
//              v_max3_f32 + v_max_f32, v_min3_f32 + v_min_f32, the slack v_fma_f32, v_cmp + v_cndmask + v_or_b32 (20 instructions,
//              dependent as in the kernel: convert -> multiply-add -> max/min -> compare)
//   f64        the exact triangle test's arithmetic: v_mul_f64 and v_add_f64 in equal parts, three independent chains
//   cndmask    v_cndmask_b32 on a mask held in a scalar register pair (the selects of the octant permutation)
// build: hipcc -O3 --offload-arch=gfx950 -o valu_issue_rate valu_issue_rate.hip ; run: ./valu_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float *out, unsigned seed, int iters)
{
	float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
	const float k = 1.0f + seed * 1e-9f, tmin = 1e-3f, tfar = 1e30f;
	unsigned q0 = seed * 2654435761u + threadIdx.x, q1 = q0 * 40503u + 1u, q2 = q1 * 40503u + 1u, hit = 0;
	double d0 = 1.0 + threadIdx.x * 1e-6, d1 = d0 + 1.0, d2 = d0 + 2.0;
	const double dk = 1.0 + seed * 1e-12;
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {
#define STEP(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k), "v"(a0));
			STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2)
			STEP(a3) STEP(a4) STEP(a5) STEP(a6)
#undef STEP
		} else if (MODE == 1) { // one child's box test: 20 instructions
			float c0, c1, c2, c3, c4, c5, tn, tf;
			asm volatile(
			    "v_cvt_f32_ubyte0 %0, %9\n v_cvt_f32_ubyte1 %1, %9\n v_cvt_f32_ubyte2 %2, %10\n v_cvt_f32_ubyte3 %3, %10\n v_cvt_f32_ubyte0 %4, %11\n v_cvt_f32_ubyte1 %5, %11\n"
			    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %14\n v_fma_f32 %2, %2, %12, %15\n v_fma_f32 %3, %3, %12, %13\n v_fma_f32 %4, %4, %12, %14\n v_fma_f32 %5, %5, %12, %15\n"
			    "v_max3_f32 %6, %0, %1, %2\n v_max_f32 %6, %6, %16\n v_min3_f32 %7, %3, %4, %5\n v_min_f32 %7, %7, %17\n"
			    "v_fma_f32 %7, |%7|, %18, %7\n v_cmp_le_f32 vcc, %6, %7\n v_cndmask_b32 %6, 0, 1, vcc\n v_or_b32 %8, %8, %6\n"
			    : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5), "=&v"(tn), "=&v"(tf), "+v"(hit)
			    : "v"(q0), "v"(q1), "v"(q2), "v"(k), "v"(a1), "v"(a2), "v"(a3), "v"(tmin), "v"(tfar), "v"(1e-6f)
			    : "vcc");
		} else if (MODE == 2) { // fp64: multiply and add in equal parts, 20 instructions
#define STEP(d) asm volatile("v_mul_f64 %0, %0, %1\n v_add_f64 %0, %0, %2" : "+v"(d) : "v"(dk), "v"(d0));
			STEP(d1) STEP(d2) STEP(d1) STEP(d2) STEP(d1) STEP(d2) STEP(d1) STEP(d2) STEP(d1) STEP(d2)
#undef STEP
		} else { // selects on a mask in scalar registers, 20 instructions
			const bool p = (seed & 1u) != 0;
			unsigned long long m = __ballot(p);
#define STEP(a, b) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m));
			STEP(a1, a2) STEP(a3, a4) STEP(a5, a6) STEP(a7, a1) STEP(a2, a3) STEP(a4, a5) STEP(a6, a7) STEP(a1, a3) STEP(a2, a4) STEP(a5, a7)
			STEP(a1, a2) STEP(a3, a4) STEP(a5, a6) STEP(a7, a1) STEP(a2, a3) STEP(a4, a5) STEP(a6, a7) STEP(a1, a3) STEP(a2, a4) STEP(a5, a7)
#undef STEP
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)hit + (float)(d1 + d2);
}

template <int MODE>
double rate(float *d, int waves_per_simd, int iters)
{
	// 256 CUs x 4 SIMDs; a workgroup of 256 threads puts one wave on each SIMD of its CU, and the dispatcher deals workgroups to
	// the CUs evenly: 256 x W workgroups give every SIMD W waves, all resident from the start
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const dim3 grid(256 * waves_per_simd);
	hipLaunchKernelGGL(k_rate<MODE>, grid, dim3(256), 0, 0, d, 1u, iters);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k_rate<MODE>, grid, dim3(256), 0, 0, d, 1u, iters);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	const double instr = 20.0 * iters * 256.0 * waves_per_simd * 4.0; // wave-level instructions of the measured part
	return instr / (ms * 1e-3);
}

int main()
{
	float *d;
	(void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
	const int iters = 20000;
	printf("sustained vector-instruction issue, whole chip (1024 SIMDs), G wave-instructions/s; in brackets cycles per instruction per SIMD at 2.4 GHz\n");
	printf("%-10s %14s %14s %14s %14s\n", "mix", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
	const char *names[4] = {"fma", "node", "f64", "cndmask"};
	for (int mode = 0; mode < 4; mode++) {
		printf("%-10s", names[mode]);
		for (int w = 1; w <= 8; w *= 2) {
			const double r = mode == 0 ? rate<0>(d, w, iters) : mode == 1 ? rate<1>(d, w, iters) : mode == 2 ? rate<2>(d, w, iters) : rate<3>(d, w, iters);
			printf(" %7.1f [%4.2f]", r / 1e9, 2.4e9 * 1024.0 / r);
		}
		printf("\n");
	}
	(void)hipFree(d);
	return 0;
}
