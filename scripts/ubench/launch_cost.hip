// Dev micro-benchmark: per-kernel cost of an (almost) empty kernel inside a hipGraph, by grid shape.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1; }
int main() {
	int *p; hipMalloc(&p, 4);
	hipStream_t st; hipStreamCreate(&st);
	const int shapes[][2] = {{4096, 64}, {2048, 128}, {1024, 256}, {512, 512}, {256, 1024}, {16384, 64}, {65536, 64}, {1, 64}};
	for (auto &sh : shapes) {
		hipGraph_t g; hipGraphExec_t e;
		hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
		for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k, dim3(sh[0]), dim3(sh[1]), 0, st, p);
		hipStreamEndCapture(st, &g);
		hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
		hipGraphLaunch(e, st); hipStreamSynchronize(st);
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		float best = 1e9;
		for (int r = 0; r < 5; ++r) {
			hipEventRecord(a, st); hipGraphLaunch(e, st); hipEventRecord(b, st); hipStreamSynchronize(st);
			float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
		}
		printf("grid %6d x block %4d : %.3f us per kernel\n", sh[0], sh[1], best * 1000 / 400);
		hipGraphExecDestroy(e); hipGraphDestroy(g);
	}
	return 0;
}
