// issue rate of v_pk_fma_f32 (two fp32 multiply-adds per instruction) and of v_fma_mix_f32 (an fp16 operand converted inside the fma) against v_cvt_f32_ubyte0 + v_fma_f32 and plain v_fma_f32
// build: hipcc -O3 --offload-arch=gfx950 -o fma_mix_rate fma_mix_rate.hip ; run: ./fma_mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float *out, unsigned seed, int iters)
{
	float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
	const float k = 1.0f + seed * 1e-9f;
	typedef float v2f __attribute__((ext_vector_type(2)));
	v2f p0 = {a0, a0 + 0.5f}, p1 = {a1, a1 + 0.5f}, p2 = {a2, a2 + 0.5f}, p3 = {a3, a3 + 0.5f}, p4 = {a4, a4 + 0.5f}, p5 = {a5, a5 + 0.5f}, p6 = {a6, a6 + 0.5f},
	    p7 = {a7, a7 + 0.5f}, kk = {k, k};
	unsigned q = seed * 2654435761u + threadIdx.x; // bytes / halves to convert
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) { // plain fma
#define STEP(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k), "v"(a0));
			STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2)
#undef STEP
		} else if (MODE == 1) { // cvt + fma
#define STEP(a) { float t; asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t) : "v"(q)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(t), "v"(k)); }
			STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2)
#undef STEP
		} else if (MODE == 3) { // packed fp32 fma: two multiply-adds per instruction on an aligned register pair
#define STEP(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(kk), "v"(p0));
			STEP(p1) STEP(p2) STEP(p3) STEP(p4) STEP(p5) STEP(p6) STEP(p7) STEP(p1) STEP(p2) STEP(p3) STEP(p4) STEP(p5) STEP(p6) STEP(p7) STEP(p1) STEP(p2)
#undef STEP
		} else { // fma_mix with the first operand an fp16 (high half of q)
#define STEP(a) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(q), "v"(k));
			STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) STEP(a1) STEP(a2)
#undef STEP
		}
	}
	const v2f ps = p1 + p2 + p3 + p4 + p5 + p6 + p7;
	out[blockIdx.x * 256 + threadIdx.x] = a1 + a2 + a3 + a4 + a5 + a6 + a7 + ps.x + ps.y;
}

template <int MODE>
float run(float *d, int iters)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, iters);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main()
{
	float *d;
	hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
	const int iters = 20000;
	const double steps = 16.0 * iters * 256 * 8 * 4; // wave-level steps: 4 waves per workgroup
	const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
	// 1024 SIMDs; cycles per wave-instruction-step per SIMD at 2.4 GHz
	printf("v_fma_f32            %.2f ms  %.2f cycles per step per SIMD\n", t0, t0 * 1e-3 * 2.4e9 * 1024 / steps);
	printf("v_cvt_ubyte + v_fma  %.2f ms  %.2f\n", t1, t1 * 1e-3 * 2.4e9 * 1024 / steps);
	printf("v_fma_mix_f32 (f16)  %.2f ms  %.2f\n", t2, t2 * 1e-3 * 2.4e9 * 1024 / steps);
	printf("v_pk_fma_f32 (2 fma) %.2f ms  %.2f\n", t3, t3 * 1e-3 * 2.4e9 * 1024 / steps);
	// value check of the mix form: high half of 0x5BF80000 is 255.0 in fp16
	return 0;
}
