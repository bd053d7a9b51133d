// Dev micro-benchmark: issue cost of v_fmac_f32 vs v_pk_fma_f32 vs v_pk_mov_b32 per wave64 on one SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc) {
	float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
	float t = 1.0001f;
	f2 tt = {t, t};
	long long c0 = clock64();
	for (int i = 0; i < iters; ++i) {
		if (MODE == 0) {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a0) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a1) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a2) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a3) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a4) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a5) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a6) : "v"(t));
				asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(a7) : "v"(t));
			}
		} else if (MODE == 1) {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p0) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p1) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p2) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p3) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p4) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p5) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p6) : "v"(tt));
				asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(p7) : "v"(tt));
			}
		} else if (MODE == 2) {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p0) : "v"(p1), "v"(p2));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p3) : "v"(p4), "v"(p5));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p6) : "v"(p7), "v"(p1));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p2) : "v"(p4), "v"(p5));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p0) : "v"(p1), "v"(p7));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p3) : "v"(p4), "v"(p5));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p6) : "v"(p7), "v"(p1));
				asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p2) : "v"(p4), "v"(p5));
			}
		} else {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				asm volatile("v_add_f32 %0, %1, %0" : "+v"(a0) : "v"(t));
				asm volatile("v_mov_b32 %0, %1" : "=v"(a1) : "v"(a0));
				asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a2) : "v"(a3));
				asm volatile("v_floor_f32 %0, %1" : "=v"(a3) : "v"(a4));
				asm volatile("v_add_u32 %0, %1, %0" : "+v"(a4) : "v"(t));
				asm volatile("v_max_f32 %0, %1, %0" : "+v"(a5) : "v"(t));
				asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a6) : "v"(a5));
				asm volatile("v_and_b32 %0, %1, %0" : "+v"(a7) : "v"(t));
			}
		}
	}
	long long c1 = clock64();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}
template <int MODE>
void run(const char *name, int waves_per_simd) {
	float *out; long long *cyc, h;
	const int blocks = 256 * waves_per_simd; // 4 waves per block = one per SIMD
	hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
	const int iters = 2000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, 256>>>(out, iters, cyc); hipDeviceSynchronize();
	hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
	const double n = (double)iters * 64;
	printf("%-12s waves/SIMD=%d  %.3f ms  => %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc), clock64 delta/instr=%.2f\n", name,
	       waves_per_simd, ms, ms * 1e6 / (n * waves_per_simd), ms * 1e6 / (n * waves_per_simd) * 2.4, (double)h / n);
	hipFree(out); hipFree(cyc);
}
int main() {
	for (int w : {1, 2, 4}) {
		run<0>("v_fmac_f32", w);
		run<1>("v_pk_fma_f32", w);
		run<2>("v_pk_mov_b32", w);
		run<3>("misc valu", w);
	}
	return 0;
}
