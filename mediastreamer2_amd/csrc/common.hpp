// common.hpp -- shared host-side plumbing of libmsmi355x (error reporting,
// context object, small RAII helpers).  gfx950 only; no CUDA/other-backend paths.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/msmi355x.h"

namespace mi {

void set_error(const char *fmt, ...);

#define MI_HIP(expr)                                                                        \
	do {                                                                                    \
		hipError_t e__ = (expr);                                                            \
		if (e__ != hipSuccess) {                                                            \
			mi::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
			return MI_ENODEV;                                                               \
		}                                                                                   \
	} while (0)

#define MI_CHECK_ARG(cond)                                                           \
	do {                                                                             \
		if (!(cond)) {                                                               \
			mi::set_error("%s:%d invalid argument: %s", __FILE__, __LINE__, #cond);  \
			return MI_EINVAL;                                                        \
		}                                                                            \
	} while (0)

#define MI_LAUNCH_CHECK()                                                                  \
	do {                                                                                   \
		hipError_t e__ = hipGetLastError();                                                \
		if (e__ != hipSuccess) {                                                           \
			mi::set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
			return MI_ENODEV;                                                              \
		}                                                                                  \
	} while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Workgroups are observed to be dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with its own L2.  A
// kernel whose neighbouring work items touch the same cache lines (rows of 960 bytes are not multiples of the 128-byte
// line) launches xcd_grid(n) blocks and turns blockIdx.x into its work item with xcd_item(): every XCD then owns one
// contiguous run of items, and a line two neighbours share is fetched into one L2 instead of two.  Speed only -- any
// placement computes the same thing.  Items >= n (the padding of the last run) return at once.
inline unsigned xcd_grid(unsigned n) { return (n + 7u) / 8u * 8u; }
__device__ __forceinline__ unsigned xcd_item(unsigned block, unsigned grid) { return (block & 7u) * (grid >> 3) + (block >> 3); }
inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
// One kernel per translation unit, registered at load time: mi_warmup (ctx.hip) asks the runtime for its attributes, which makes it load
// the unit's code object on the context's device THEN -- not under the first launch of a ticker's first tick (HIP loads a code object
// lazily, behind one process-wide lock: sixteen ticker threads' first ticks waited 0.1 - 0.3 s for it).
void warm_register(const void *kernel);
struct WarmEntry {
	explicit WarmEntry(const void *k) { warm_register(k); }
};

} // namespace mi

struct mi_graph {
	struct mi_ctx *ctx = nullptr;
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
};

// MSBufferizer for a batch of streams on the device (fifo.hip).  Defined here because the canceller and the volume
// kernel read / write the rings directly (no separate pop / push launch).
struct mi_fifo {
	struct mi_ctx *ctx = nullptr;
	int nstreams = 0, capacity = 0;
	int16_t *d_ring = nullptr; // [nstreams][capacity]
	int2 *d_pos = nullptr;     // x = head (index of the oldest sample, < capacity), y = level (samples held)
	int32_t *d_overflow = nullptr;
};
// what a kernel needs of one
struct FifoView {
	int16_t *ring;
	int2 *pos;
	int32_t *overflow;
	int cap;
};
inline FifoView fifo_view(const mi_fifo *f) {
	FifoView v;
	v.ring = f ? f->d_ring : nullptr;
	v.pos = f ? f->d_pos : nullptr;
	v.overflow = f ? f->d_overflow : nullptr;
	v.cap = f ? f->capacity : 0;
	return v;
}

// What the canceller's tick kernel needs of an up-sampling mi_resampler to run it as its own first phase (MSResample folded
// into MSSpeexEC's launch, mi_aec_process_fifos_resampled).  Filled by mi_resampler_view (resample.hip); ok = the resampler
// is an integer up-sampler with the 48-tap direct table whose streams all sit on a whole output period.
struct ResamplerView {
	int16_t *hist = nullptr;     // [nstreams][hist_stride] the last filt-1 input samples of every stream
	const float *table = nullptr; // [den][filt] polyphase rows
	int hist_stride = 0, den = 0, filt = 0, nstreams = 0;
	int device = -1;
	bool ok = false;
};
struct mi_resampler;
void mi_resampler_view(mi_resampler *r, ResamplerView *v);

// What the fused volume + conference mix kernel (volume.hip) needs of an mi_mixer: its per-pin controls
struct MixerView {
	const uint8_t *flags = nullptr; // [nconf][mm] MI_MIX_*
	const float *gain = nullptr;    // [nconf][mm]
	int nconf = 0, mm = 0, ns = 0, device = -1;
};
struct mi_mixer;
void mi_mixer_view(const mi_mixer *m, MixerView *v);

struct mi_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	int cu_count = 0;
	size_t hbm_bytes = 0;
	char name[128] = {0};
	// scratch for *_host entry points (grown on demand)
	void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t scratch_bytes[4] = {0, 0, 0, 0};
	int ensure_scratch(int slot, size_t bytes, void **out);
	int activate() const;
};
