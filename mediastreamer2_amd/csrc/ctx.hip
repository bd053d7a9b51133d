// ctx.hip -- context, memory and timing entry points of the C ABI.
#include "common.hpp"

#include <algorithm>

namespace mi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
} // namespace mi

int mi_ctx::activate() const {
	MI_HIP(hipSetDevice(device));
	return MI_OK;
}

int mi_ctx::ensure_scratch(int slot, size_t bytes, void **out) {
	if (bytes > scratch_bytes[slot]) {
		if (scratch[slot]) (void)hipFree(scratch[slot]);
		scratch[slot] = nullptr;
		scratch_bytes[slot] = 0;
		MI_HIP(hipMalloc(&scratch[slot], bytes));
		scratch_bytes[slot] = bytes;
	}
	*out = scratch[slot];
	return MI_OK;
}

// ---- copies between PINNED host memory (mi_host_alloc: mapped into the device's address space) and device memory as a
// kernel of this library on the context's stream, not hipMemcpyAsync.  Why: the plugin's tick path is a handful of small
// copies around its launches, and now and then ONE hipMemcpyAsync of a few KB keeps its caller on the CPU for ~10 ms inside
// the runtime (an ioctl under the HSA copy path in one thread, sched_yield loops in the threads beside it; tests/host/
// plugin_bench PLUGIN_BENCH_STACKS, profiles/r04_plugin_stacks.txt) -- longer than the tick it belongs to.  A kernel launch is
// one AQL packet: nothing to allocate, map or wait for on the way.  (profiles/r04_plugin_copies_kernel_vs_hip.txt holds the A/B.)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void copy_kernel(T *__restrict__ dst, const T *__restrict__ src, size_t n) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int copy_mapped(mi_ctx *c, void *dst, const void *src, size_t n, hipMemcpyKind kind) {
	if (n == 0) return MI_OK;
	(void)kind;
	const uintptr_t al = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | (uintptr_t)n;
	auto grid = [](size_t units) { return dim3((unsigned)std::min<size_t>((units + 255) / 256, 2048)); };
	if ((al & 15) == 0) hipLaunchKernelGGL(copy_kernel<uint4>, grid(n / 16), dim3(256), 0, c->stream, (uint4 *)dst, (const uint4 *)src, n / 16);
	else if ((al & 3) == 0) hipLaunchKernelGGL(copy_kernel<uint32_t>, grid(n / 4), dim3(256), 0, c->stream, (uint32_t *)dst, (const uint32_t *)src, n / 4);
	else hipLaunchKernelGGL(copy_kernel<uint8_t>, grid(n), dim3(256), 0, c->stream, (uint8_t *)dst, (const uint8_t *)src, n);
	MI_LAUNCH_CHECK();
	return MI_OK;
}
} // namespace

namespace {
__global__ void stamp_kernel(unsigned long long *dst) { *dst = wall_clock64(); }
} // namespace

extern "C" {

// debug entries (not in the public header): the device's constant-rate clock written by a one-thread launch on the context's
// stream -- inside a captured graph it brackets the graph's work on the DEVICE's time line (scripts/paced_events_probe.py tells a
// tick the device took long over from one the host was slow to submit); the clock's rate in kHz
int mi_debug_stamp(mi_ctx *c, unsigned long long *d_dst) {
	MI_CHECK_ARG(c && d_dst);
	if (c->activate() != MI_OK) return MI_ENODEV;
	hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, c->stream, d_dst);
	MI_LAUNCH_CHECK();
	return MI_OK;
}
int mi_debug_wall_clock_khz(mi_ctx *c) {
	int khz = 0;
	if (!c || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess) return 0;
	return khz;
}

int mi_abi_version(void) { return MSMI355X_ABI_VERSION; }
const char *mi_last_error(void) { return mi::g_err; }

int mi_device_count(void) {
	int ndev = 0;
	const hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) {
		mi::set_error("no HIP device available (%s); libmsmi355x has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
		return 0;
	}
	return ndev;
}

int mi_ctx_create(int device, void *hip_stream, mi_ctx **out) {
	MI_CHECK_ARG(out != nullptr);
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) {
		mi::set_error("no HIP device available (%s); libmsmi355x has no CPU fallback",
		              e != hipSuccess ? hipGetErrorString(e) : "device count 0");
		return MI_ENODEV;
	}
	MI_CHECK_ARG(device >= 0 && device < ndev);
	MI_HIP(hipSetDevice(device));
	mi_ctx *c = new mi_ctx();
	c->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
		c->cu_count = prop.multiProcessorCount;
		c->hbm_bytes = prop.totalGlobalMem;
		snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
	}
	if (hip_stream) {
		c->stream = (hipStream_t)hip_stream;
		c->own_stream = false;
	} else {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
			mi::set_error("hipStreamCreate failed");
			delete c;
			return MI_ENODEV;
		}
		c->own_stream = true;
	}
	if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
		mi::set_error("hipEventCreate failed");
		delete c;
		return MI_ENODEV;
	}
	*out = c;
	return MI_OK;
}

void mi_ctx_destroy(mi_ctx *c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (int i = 0; i < 4; ++i)
		if (c->scratch[i]) (void)hipFree(c->scratch[i]);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

int mi_ctx_sync(mi_ctx *c) {
	MI_CHECK_ARG(c != nullptr);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

void *mi_ctx_stream(mi_ctx *c) { return c ? (void *)c->stream : nullptr; }
int mi_ctx_device(mi_ctx *c) { return c ? c->device : -1; }

int mi_ctx_props(mi_ctx *c, int *cu_count, size_t *hbm_bytes, char *name, int name_cap) {
	MI_CHECK_ARG(c != nullptr);
	if (cu_count) *cu_count = c->cu_count;
	if (hbm_bytes) *hbm_bytes = c->hbm_bytes;
	if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", c->name);
	return MI_OK;
}

void *mi_dev_alloc(mi_ctx *c, size_t bytes) {
	if (!c || c->activate() != MI_OK) return nullptr;
	void *p = nullptr;
	if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
		mi::set_error("hipMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void mi_dev_free(mi_ctx *c, void *p) {
	if (!c || !p) return;
	(void)hipSetDevice(c->device);
	(void)hipFree(p);
}

void *mi_host_alloc(mi_ctx *c, size_t bytes) {
	if (!c || c->activate() != MI_OK) return nullptr;
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		mi::set_error("hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void mi_host_free(mi_ctx *c, void *p) {
	if (!p) return;
	if (c) (void)hipSetDevice(c->device);
	(void)hipHostFree(p); // (pinned memory belongs to no device: a buffer may outlive the context it was allocated through)
}

int mi_copy_h2d(mi_ctx *c, void *d, const void *h, size_t n) {
	MI_CHECK_ARG(c && d && h);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
	return MI_OK;
}

int mi_copy_d2h(mi_ctx *c, void *h, const void *d, size_t n) {
	MI_CHECK_ARG(c && d && h);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
	return MI_OK;
}

int mi_copy_h2d_pinned(mi_ctx *c, void *d, const void *h_pinned, size_t n) {
	MI_CHECK_ARG(c && d && h_pinned);
	if (c->activate() != MI_OK) return MI_ENODEV;
	return copy_mapped(c, d, h_pinned, n, hipMemcpyHostToDevice);
}

int mi_copy_d2h_pinned(mi_ctx *c, void *h_pinned, const void *d, size_t n) {
	MI_CHECK_ARG(c && d && h_pinned);
	if (c->activate() != MI_OK) return MI_ENODEV;
	return copy_mapped(c, h_pinned, d, n, hipMemcpyDeviceToHost);
}

int mi_memset(mi_ctx *c, void *d, int value, size_t n) {
	MI_CHECK_ARG(c && d);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemsetAsync(d, value, n, c->stream));
	return MI_OK;
}

int mi_ctx_capture_begin(mi_ctx *c) {
	MI_CHECK_ARG(c != nullptr);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
	return MI_OK;
}

int mi_ctx_capture_end(mi_ctx *c, mi_graph **out) {
	MI_CHECK_ARG(c && out);
	*out = nullptr;
	if (c->activate() != MI_OK) return MI_ENODEV;
	hipGraph_t graph = nullptr;
	MI_HIP(hipStreamEndCapture(c->stream, &graph));
	mi_graph *g = new mi_graph();
	g->ctx = c;
	g->graph = graph;
	hipError_t e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
	if (e != hipSuccess) {
		mi::set_error("hipGraphInstantiate -> %s", hipGetErrorString(e));
		(void)hipGraphDestroy(graph);
		delete g;
		return MI_ENODEV;
	}
	*out = g;
	return MI_OK;
}

int mi_graph_launch(mi_graph *g) {
	MI_CHECK_ARG(g != nullptr);
	if (g->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
	return MI_OK;
}

void mi_graph_destroy(mi_graph *g) {
	if (!g) return;
	(void)hipSetDevice(g->ctx->device);
	if (g->exec) (void)hipGraphExecDestroy(g->exec);
	if (g->graph) (void)hipGraphDestroy(g->graph);
	delete g;
}

int mi_timer_start(mi_ctx *c) {
	MI_CHECK_ARG(c != nullptr);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipEventRecord(c->ev0, c->stream));
	return MI_OK;
}

int mi_timer_stop(mi_ctx *c, float *ms) {
	MI_CHECK_ARG(c && ms);
	if (c->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipEventRecord(c->ev1, c->stream));
	MI_HIP(hipEventSynchronize(c->ev1));
	MI_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
	return MI_OK;
}

} // extern "C"

// ---- code objects loaded ahead of the first launch (common.hpp: WarmEntry)
namespace mi {
static std::vector<const void *> &warm_list() {
	static std::vector<const void *> v; // (function-local: the units' static initialisers may run before this one's)
	return v;
}
void warm_register(const void *kernel) { warm_list().push_back(kernel); }
} // namespace mi
static const mi::WarmEntry g_warm_ctx(reinterpret_cast<const void *>(&stamp_kernel));

extern "C" int mi_warmup(mi_ctx *c) {
	MI_CHECK_ARG(c);
	int rc;
	if ((rc = c->activate()) != MI_OK) return rc;
	for (const void *k : mi::warm_list()) {
		hipFuncAttributes at;
		MI_HIP(hipFuncGetAttributes(&at, k)); // (makes the runtime load the kernel's code object on this device, now)
	}
	return MI_OK;
}
