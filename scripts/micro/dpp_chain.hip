// Dev micro-benchmark: a float sum over the 64 lanes of a wavefront in LANE ORDER ((..((p0 + p1) + p2)..) + p63), two ways:
// v_readlane + v_add per lane (what the canceller's library-ordered sums did), and a systolic chain of
// v_add_f32_dpp wave_shr:1 (lane l is final after step l; later steps recompute the same value).  Checks both against
// the host's sequential float sum bit for bit, then times them.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/dpp_chain scripts/micro/dpp_chain.hip && /tmp/dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>

__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float shr1z(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

template <int MODE, int CH>
__global__ void chain(const float *in, float *out, int reps) {
	const int lane = threadIdx.x & 63;
	const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	float p[CH][2];
	for (int c = 0; c < CH; ++c)
		for (int k = 0; k < 2; ++k) p[c][k] = in[(c * 64 + lane) * 2 + k];
	float tot[CH];
	for (int c = 0; c < CH; ++c) tot[c] = 0;
	for (int r = 0; r < reps; ++r) {
		float s[CH];
		for (int c = 0; c < CH; ++c) s[c] = 0;
		if (MODE == 0) {
#pragma unroll 2
			for (int l = 0; l < 64; ++l)
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					s[c] = s[c] + rdlane(p[c][0], l);
					s[c] = s[c] + rdlane(p[c][1], l);
				}
		} else {
#pragma unroll 4
			for (int l = 0; l < 64; ++l)
#pragma unroll
				for (int c = 0; c < CH; ++c) {
					s[c] = shr1z(s[c]) + p[c][0];
					s[c] = s[c] + p[c][1];
				}
#pragma unroll
			for (int c = 0; c < CH; ++c) s[c] = rdlane(s[c], 63);
		}
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			tot[c] = s[c];
			p[c][0] += tot[c] * 0.f; // keep the loop from being hoisted
		}
	}
	if (lane == 0)
		for (int c = 0; c < CH; ++c) out[w * CH + c] = tot[c];
}

int main() {
	const int CH = 4;
	std::vector<float> h(CH * 128);
	std::mt19937 g(1);
	std::normal_distribution<float> nd(0.f, 1000.f);
	for (auto &v : h) v = nd(g);
	float *din, *dout;
	const int waves = 256 * 8;
	hipMalloc(&din, h.size() * 4);
	hipMalloc(&dout, waves * CH * 4);
	hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	float want[CH];
	for (int c = 0; c < CH; ++c) {
		volatile float s = 0;
		for (int l = 0; l < 64; ++l) {
			s = s + h[(c * 64 + l) * 2];
			s = s + h[(c * 64 + l) * 2 + 1];
		}
		want[c] = s;
	}
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int mode = 0; mode < 2; ++mode) {
		const int reps = 2000;
		auto launch = [&](int r) {
			if (mode == 0) hipLaunchKernelGGL((chain<0, CH>), dim3(waves), dim3(64), 0, 0, din, dout, r);
			else hipLaunchKernelGGL((chain<1, CH>), dim3(waves), dim3(64), 0, 0, din, dout, r);
		};
		launch(1);
		hipDeviceSynchronize();
		std::vector<float> got(CH);
		hipMemcpy(got.data(), dout, CH * 4, hipMemcpyDeviceToHost);
		bool ok = true;
		for (int c = 0; c < CH; ++c) ok &= (memcmp(&got[c], &want[c], 4) == 0);
		hipEventRecord(e0);
		launch(reps);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%s: bit-exact %s, %.3f us per set of %d chains of 128 adds (8 waves per CU resident)\n", mode ? "dpp wave_shr chain" : "readlane chain   ",
		       ok ? "yes" : "NO", ms * 1e3 / reps, CH);
	}
	return 0;
}
