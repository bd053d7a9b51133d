// Dev check: is a wait state needed between a packed-FP32 VALU write of a register pair and a 32-bit VALU read of one half
// of it (the compiler puts `s_nop 0` there in its own code; inline asm is opaque to its hazard recogniser)?
// Runs both forms back to back over random data and compares with the host.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_hazard scripts/micro/pk_hazard.hip && /tmp/pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NOP>
__global__ void k(const v2f *in, float *out, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	v2f a = in[2 * i], b = in[2 * i + 1];
	float lo, hi, ry;
	// the packed result lands in a fixed register pair so that its halves can be named in the same asm block
	if (NOP)
		asm volatile("v_pk_add_f32 v[100:101], %3, %4\n\ts_nop 0\n\tv_mul_f32 %0, 3.0, v100\n\tv_mul_f32 %1, 5.0, v100\n\tv_mov_b32 %2, v101"
		             : "=&v"(lo), "=&v"(hi), "=&v"(ry) : "v"(a), "v"(b) : "v100", "v101");
	else
		asm volatile("v_pk_add_f32 v[100:101], %3, %4\n\tv_mul_f32 %0, 3.0, v100\n\tv_mul_f32 %1, 5.0, v100\n\tv_mov_b32 %2, v101"
		             : "=&v"(lo), "=&v"(hi), "=&v"(ry) : "v"(a), "v"(b) : "v100", "v101");
	// second form: read the HIGH half right behind the packed write
	float h2, r2y;
	if (NOP)
		asm volatile("v_pk_mul_f32 v[102:103], %2, %3\n\ts_nop 0\n\tv_add_f32 %0, 1.0, v103\n\tv_mov_b32 %1, v102" : "=&v"(h2), "=&v"(r2y) : "v"(a), "v"(b) : "v102", "v103");
	else
		asm volatile("v_pk_mul_f32 v[102:103], %2, %3\n\tv_add_f32 %0, 1.0, v103\n\tv_mov_b32 %1, v102" : "=&v"(h2), "=&v"(r2y) : "v"(a), "v"(b) : "v102", "v103");
	out[4 * i] = lo;
	out[4 * i + 1] = hi;
	out[4 * i + 2] = h2;
	out[4 * i + 3] = ry + r2y;
}

int main() {
	const int n = 1 << 22;
	std::vector<float> h(4 * n);
	std::mt19937 g(7);
	std::uniform_real_distribution<float> u(-100.f, 100.f);
	for (auto &v : h) v = u(g);
	v2f *din;
	float *dout;
	hipMalloc(&din, 4 * n * 4);
	hipMalloc(&dout, 4 * n * 4);
	hipMemcpy(din, h.data(), 4 * n * 4, hipMemcpyHostToDevice);
	std::vector<float> got(4 * n);
	for (int nop = 0; nop < 2; ++nop) {
		long bad = 0;
		for (int rep = 0; rep < 20; ++rep) {
			if (nop) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
			else hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
			hipMemcpy(got.data(), dout, 4 * n * 4, hipMemcpyDeviceToHost);
			for (int i = 0; i < n; ++i) {
				const float ax = h[4 * i], ay = h[4 * i + 1], bx = h[4 * i + 2], by = h[4 * i + 3];
				const float lo = 3.0f * (ax + bx), hi = 5.0f * (ax + bx), h2 = 1.0f + (ay * by), s = (ay + by) + (ax * bx);
				bad += (got[4 * i] != lo) + (got[4 * i + 1] != hi) + (got[4 * i + 2] != h2) + (got[4 * i + 3] != s);
			}
		}
		printf("%s s_nop: %ld mismatches in %d x 20 lanes\n", nop ? "with" : "without", bad, n);
	}
	return 0;
}
