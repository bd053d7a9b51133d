// Dev micro-benchmark (not part of the product): what HBM rate does the canceller's ACCESS PATTERN reach with no
// arithmetic at all?  One wavefront per leg, per block j of M: read X(j), W(j), FG(j) (2 KB each, 32 B per lane),
// write W(j) -- the streaming pass of aec_tick_kernel<256> -- at the canceller's residency (8 waves per CU through a
// 20 KB LDS allocation) and at variations of it, next to a plain float4 copy.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/stream_pattern scripts/ubench/stream_pattern.hip
//   ./scripts/ubench/stream_pattern [legs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e__ = (x);                                                          \
		if (e__ != hipSuccess) {                                                       \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
			exit(1);                                                                   \
		}                                                                              \
	} while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int M = 24, BLK = 2048; // blocks per leg, bytes per block (N = 512 floats)

// DEPTH blocks in flight per wave; PAIR = blocks handled per iteration (1: the kernel's frame-1 pass, 2: its frame-2 pass)
template <int PAIR, bool WRITE, int NT = 0> // NT: 1 = non-temporal stores, 2 = non-temporal loads of X and FG, 3 = both
__global__ __launch_bounds__(64) void legs_kernel(const v4f *__restrict__ X, v4f *__restrict__ W, const v4f *__restrict__ FG, int legs,
                                                  v4f *sink) {
	extern __shared__ char lds[];
	const int lane = threadIdx.x;
	v4f acc = {0, 0, 0, 0};
	for (int s = blockIdx.x; s < legs; s += gridDim.x) {
		const v4f *x = X + (size_t)s * (M * BLK / 16), *fg = FG + (size_t)s * (M * BLK / 16);
		v4f *w = W + (size_t)s * (M * BLK / 16);
		for (int j = 0; j < M; j += PAIR) {
			v4f a[PAIR][2], b[PAIR][2], c[PAIR][2];
#pragma unroll
			for (int p = 0; p < PAIR; ++p)
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int o = (j + p) * (BLK / 16) + h * 64 + lane;
					a[p][h] = (NT & 2) ? __builtin_nontemporal_load(x + o) : x[o];
					b[p][h] = w[o];
					c[p][h] = (NT & 2) ? __builtin_nontemporal_load(fg + o) : fg[o];
				}
#pragma unroll
			for (int p = 0; p < PAIR; ++p)
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int o = (j + p) * (BLK / 16) + h * 64 + lane;
					v4f r = a[p][h] + b[p][h];
					acc.x += c[p][h].x + c[p][h].y + c[p][h].z + c[p][h].w;
					if (WRITE && (NT & 1)) __builtin_nontemporal_store(r, w + o);
					else if (WRITE) w[o] = r;
					else acc.y += r.x + r.y + r.z + r.w;
				}
		}
	}
	if (acc.x == 12345.678f) sink[lane] = acc; // keeps the loads alive
	if (lds[lane] == 77 && acc.y == 1.5f) sink[lane + 64] = acc;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	launch();
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int r = 0; r < reps; ++r) {
		CK(hipEventRecord(e0));
		launch();
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		best = ms < best ? ms : best;
	}
	return best;
}

int main(int argc, char **argv) {
	const int legs = argc > 1 ? atoi(argv[1]) : 65536;
	const size_t per = (size_t)M * BLK, bytes = per * legs;
	v4f *X, *W, *FG, *sink;
	CK(hipMalloc(&X, bytes));
	CK(hipMalloc(&W, bytes));
	CK(hipMalloc(&FG, bytes));
	CK(hipMalloc(&sink, 4096));
	CK(hipMemset(X, 0, bytes));
	CK(hipMemset(W, 0, bytes));
	CK(hipMemset(FG, 0, bytes));
	printf("%d legs, %d blocks of %d B per array and leg: %.2f GB read x3, written x1 per pass\n", legs, M, BLK, bytes / 1e9);
	struct V {
		const char *name;
		int lds, grid;
	};
	for (int lds_kb : {20, 10, 5, 0}) {
		const int waves = lds_kb ? 160 / lds_kb : 16; // per CU
		for (int grid : {legs, 256 * waves}) {
			float t1 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<1, true>), dim3(grid), dim3(64), lds_kb * 1024, 0, X, W, FG, legs, sink); });
			float t2 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<2, true>), dim3(grid), dim3(64), lds_kb * 1024, 0, X, W, FG, legs, sink); });
			float t3 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<1, false>), dim3(grid), dim3(64), lds_kb * 1024, 0, X, W, FG, legs, sink); });
			printf("LDS %2d KB (<= %2d waves per CU) grid %6d: block by block %.3f ms %.2f TB/s | block pairs %.3f ms %.2f TB/s | reads only %.3f ms %.2f TB/s\n",
			       lds_kb, waves, grid, t1, 4 * bytes / t1 / 1e9, t2, 4 * bytes / t2 / 1e9, t3, 3 * bytes / t3 / 1e9);
		}
	}
	for (int grid : {legs, 2048}) {
		float t1 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<1, true, 1>), dim3(grid), dim3(64), 20 * 1024, 0, X, W, FG, legs, sink); });
		float t2 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<1, true, 2>), dim3(grid), dim3(64), 20 * 1024, 0, X, W, FG, legs, sink); });
		float t3 = time_ms([&] { hipLaunchKernelGGL((legs_kernel<1, true, 3>), dim3(grid), dim3(64), 20 * 1024, 0, X, W, FG, legs, sink); });
		printf("LDS 20 KB grid %6d, block by block: non-temporal stores %.3f ms %.2f TB/s | non-temporal loads of X and FG %.3f ms %.2f TB/s | both %.3f ms %.2f TB/s\n",
		       grid, t1, 4 * bytes / t1 / 1e9, t2, 4 * bytes / t2 / 1e9, t3, 4 * bytes / t3 / 1e9);
	}
	const size_t n = bytes / 16;
	float tc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)X, (float4 *)W, n); });
	printf("float4 copy, grid-stride, 4096 x 256 threads: %.3f ms %.2f TB/s\n", tc, 2 * bytes / tc / 1e9);
	float tm = time_ms([&] { CK(hipMemcpyAsync(W, X, bytes, hipMemcpyDeviceToDevice, 0)); });
	printf("hipMemcpy device to device: %.3f ms %.2f TB/s\n", tm, 2 * bytes / tm / 1e9);
	return 0;
}
