// fifo.hip -- batched per-stream sample FIFOs on the device: MSBufferizer (src/base/msqueue.c:70-113) for a whole
// batch of streams, so that filters with different block sizes can be chained without leaving HBM -- the resampler
// hands over 480-sample ticks, the echo canceller eats 256-sample frames (speexec.c:252-257,:288), the mixer
// wants ticks again (audiomixer.c:78-90).
//
// One ring of `capacity` int16 samples per stream, [nstreams][capacity] in HBM, plus (head index, fill level) per
// stream -- both always below `capacity`, so a stream can run for any length of time (free-running sample counters
// would wrap after 2^32 samples = 24.8 h at 48 kHz and, with a capacity that is no power of two, land on another slot).
// push appends a block (optionally a per-stream count: 0 = nothing this round); pop is all-or-nothing like
// ms_bufferizer_read (msqueue.c:83): a stream with fewer than `frame` samples keeps them and reports ok = 0
// (its output row is zero-filled on request, the way the filters inject silence: speexec.c:261-272, audiomixer.c:88).
// One wavefront per stream; pure copies, HBM-bound.
#include "common.hpp"
#include <vector>

namespace {

struct FifoArgs {
	int16_t *ring;
	int2 *pos; // x = head (index of the oldest sample, < capacity), y = level (samples held, <= capacity)
	const uint8_t *nframes; // push_frames: frames to push per stream; pop_frames with a gate: frames wanted per stream
	uint8_t *nframes_out;   // pop_frames: frames delivered per stream
	int max_frames;
	int nstreams, capacity;
	const int16_t *in;
	const int32_t *count; // per-stream samples to push, or null = nsamples for all
	int16_t *out;
	uint8_t *ok;
	const uint8_t *gate; // pop only where gate != 0 (others: ok = 0), or null
	int nsamples, stride, zero_fill;
	int vec; // 16-byte copies allowed: capacity, stride multiples of 8 and 16-byte aligned rows
	int32_t *levels;
	int32_t *overflow; // number of pushes refused because the ring was full
};

// Stream s's share of `phases` re-framing phases: a multiplicative hash (a bijection of the 32-bit slot index) decides,
// so that ANY regular arrangement of slots -- consecutive legs of a conference, every eighth slot, one bank -- gets every
// phase equally often.  Host and device use the same function.
__host__ __device__ inline unsigned fifo_phase_of(unsigned s, unsigned phases) { return ((s * 0x9E3779B1u) >> 16) % phases; }

// `unit * phase(s)` samples of silence appended to streams [first, first + count): the lead a leg starts with
__global__ void fifo_rewind_kernel(int2 *pos, int count, int head) { // mi_fifo_reset_range_at
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i < count) pos[i] = make_int2(head, 0);
}

__global__ void fifo_lead_kernel(FifoArgs a, int first, int count, int unit, int phases) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	const int s = first + i;
	const int n = unit * (int)fifo_phase_of((unsigned)s, (unsigned)phases);
	const int2 p = a.pos[s];
	if (n == 0) return;
	if (p.y + n > a.capacity) {
		atomicAdd(a.overflow, 1);
		return;
	}
	int16_t *r = a.ring + (size_t)s * a.capacity;
	unsigned k = (unsigned)p.x + (unsigned)p.y;
	if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
	for (int j = 0; j < n; ++j) {
		r[k] = 0;
		if (++k == (unsigned)a.capacity) k = 0;
	}
	a.pos[s] = make_int2(p.x, p.y + n);
}

// d_count[s] samples of silence appended to stream s (0 = nothing): the frames of zeros MSSpeexEC injects into its delay
// line when the far end runs short (speexec.c:261-272), the delay line's initial fill (:205-208)
__global__ void fifo_silence_kernel(FifoArgs a) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= a.nstreams) return;
	const int n = a.count[s];
	if (n <= 0) return;
	const int2 p = a.pos[s];
	if (p.y + n > a.capacity) {
		atomicAdd(a.overflow, 1);
		return;
	}
	int16_t *r = a.ring + (size_t)s * a.capacity;
	unsigned k = (unsigned)p.x + (unsigned)p.y;
	if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
	for (int j = 0; j < n; ++j) {
		r[k] = 0;
		if (++k == (unsigned)a.capacity) k = 0;
	}
	a.pos[s] = make_int2(p.x, p.y + n);
}

constexpr int FIFO_WAVES = 4; // independent wavefronts (streams) per workgroup: 4x fewer workgroups to dispatch

__global__ __launch_bounds__(64 * FIFO_WAVES) void fifo_push_kernel(FifoArgs a) {
	const int s = blockIdx.x * FIFO_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (s >= a.nstreams) return;
	if (a.gate && !a.gate[s]) return;
	const int n = a.nframes ? min((int)a.nframes[s], a.max_frames) * a.nsamples
	                        : (a.count ? min(max(a.count[s], 0), a.nsamples) : a.nsamples);
	if (n == 0) return;
	const int2 p = a.pos[s];
	if (p.y + n > a.capacity) { // would overwrite unread samples: refuse the block, count it
		if (lane == 0) atomicAdd(a.overflow, 1);
		return;
	}
	int16_t *r = a.ring + (size_t)s * a.capacity;
	const int16_t *src = a.in + (size_t)s * a.stride;
	unsigned base = (unsigned)p.x + (unsigned)p.y;
	if (base >= (unsigned)a.capacity) base -= (unsigned)a.capacity;
	if (a.vec && ((n | base) & 7) == 0) { // whole 16-byte groups, also across the wrap (capacity % 8 == 0)
		for (int i = lane; i < (n >> 3); i += 64) {
			unsigned k = base + 8u * (unsigned)i;
			if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
			*reinterpret_cast<uint4 *>(r + k) = *reinterpret_cast<const uint4 *>(src + 8 * i);
		}
	} else {
		for (int i = lane; i < n; i += 64) {
			unsigned k = base + (unsigned)i;
			if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
			r[k] = src[i];
		}
	}
	if (lane == 0) a.pos[s] = make_int2(p.x, p.y + n);
}

__global__ __launch_bounds__(64 * FIFO_WAVES) void fifo_pop_kernel(FifoArgs a) {
	const int s = blockIdx.x * FIFO_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (s >= a.nstreams) return;
	const int n = a.nsamples;
	const int2 p = a.pos[s];
	const bool take = (!a.gate || a.gate[s]) && (p.y >= n);
	int16_t *dst = a.out + (size_t)s * a.stride;
	if (take) {
		const int16_t *r = a.ring + (size_t)s * a.capacity;
		const unsigned base = (unsigned)p.x;
		if (a.vec && ((n | base) & 7) == 0) {
			for (int i = lane; i < (n >> 3); i += 64) {
				unsigned k = base + 8u * (unsigned)i;
				if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
				*reinterpret_cast<uint4 *>(dst + 8 * i) = *reinterpret_cast<const uint4 *>(r + k);
			}
		} else {
			for (int i = lane; i < n; i += 64) {
				unsigned k = base + (unsigned)i;
				if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
				dst[i] = r[k];
			}
		}
		if (lane == 0) {
			unsigned h = base + (unsigned)n;
			if (h >= (unsigned)a.capacity) h -= (unsigned)a.capacity;
			a.pos[s] = make_int2((int)h, p.y - n);
		}
	} else if (a.zero_fill) {
		if (a.vec && (n & 7) == 0) {
			for (int i = lane; i < (n >> 3); i += 64) *reinterpret_cast<uint4 *>(dst + 8 * i) = make_uint4(0, 0, 0, 0);
		} else {
			for (int i = lane; i < n; i += 64) dst[i] = 0;
		}
	}
	if (lane == 0 && a.ok) a.ok[s] = take ? 1 : 0;
}

// Up to max_frames whole frames per stream in one launch, back to back in the output row: the `while` of
// speex_ec_process (speexec.c:256) for a whole tick.  Without a gate a stream delivers as many whole frames as it holds
// (reported in nframes_out: this is the canceller's per-stream frame count); with a gate it is asked for gate[s] frames
// and every frame it cannot supply is zero-filled (the silence injected for a short far end, speexec.c:261-272).
__global__ __launch_bounds__(64 * FIFO_WAVES) void fifo_pop_frames_kernel(FifoArgs a) {
	const int s = blockIdx.x * FIFO_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (s >= a.nstreams) return;
	const int n = a.nsamples;
	const int2 p = a.pos[s];
	const int have = min(p.y / n, a.max_frames);
	const int want = a.nframes ? min((int)a.nframes[s], a.max_frames) : have;
	const int take = min(have, want);
	int16_t *dst = a.out + (size_t)s * a.stride;
	const int16_t *r = a.ring + (size_t)s * a.capacity;
	const int tn = take * n;
	if (a.vec && ((n | p.x) & 7) == 0) {
		for (int i = lane; i < (tn >> 3); i += 64) {
			unsigned k = (unsigned)p.x + 8u * (unsigned)i;
			if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
			*reinterpret_cast<uint4 *>(dst + 8 * i) = *reinterpret_cast<const uint4 *>(r + k);
		}
	} else {
		for (int i = lane; i < tn; i += 64) {
			unsigned k = (unsigned)p.x + (unsigned)i;
			if (k >= (unsigned)a.capacity) k -= (unsigned)a.capacity;
			dst[i] = r[k];
		}
	}
	if (a.zero_fill)
		for (int i = tn + lane; i < want * n; i += 64) dst[i] = 0;
	if (lane == 0) {
		if (take) {
			unsigned h = (unsigned)p.x + (unsigned)tn;
			if (h >= (unsigned)a.capacity) h -= (unsigned)a.capacity;
			a.pos[s] = make_int2((int)h, p.y - tn);
		}
		if (a.nframes_out) a.nframes_out[s] = (uint8_t)take;
	}
}

__global__ void fifo_level_kernel(FifoArgs a) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s < a.nstreams) a.levels[s] = a.pos[s].y;
}

} // namespace

// struct mi_fifo: common.hpp

extern "C" {

int mi_fifo_create(mi_ctx *ctx, int nstreams, int capacity_samples, mi_fifo **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && capacity_samples > 0);
	*out = nullptr;
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_fifo *f = new mi_fifo();
	f->ctx = ctx;
	f->nstreams = nstreams;
	f->capacity = capacity_samples;
	if (hipMalloc((void **)&f->d_ring, (size_t)nstreams * capacity_samples * sizeof(int16_t)) != hipSuccess ||
	    hipMalloc((void **)&f->d_pos, (size_t)nstreams * sizeof(int2)) != hipSuccess ||
	    hipMalloc((void **)&f->d_overflow, sizeof(int32_t)) != hipSuccess) {
		mi::set_error("hipMalloc failed for %d FIFOs of %d samples", nstreams, capacity_samples);
		mi_fifo_destroy(f);
		return MI_ENOMEM;
	}
	if (hipMemsetAsync(f->d_pos, 0, (size_t)nstreams * sizeof(int2), ctx->stream) != hipSuccess ||
	    hipMemsetAsync(f->d_overflow, 0, sizeof(int32_t), ctx->stream) != hipSuccess ||
	    hipStreamSynchronize(ctx->stream) != hipSuccess) {
		mi::set_error("FIFO state reset failed");
		mi_fifo_destroy(f);
		return MI_ENODEV;
	}
	*out = f;
	return MI_OK;
}

void mi_fifo_destroy(mi_fifo *f) {
	if (!f) return;
	(void)hipSetDevice(f->ctx->device);
	if (f->d_ring) (void)hipFree(f->d_ring);
	if (f->d_pos) (void)hipFree(f->d_pos);
	if (f->d_overflow) (void)hipFree(f->d_overflow);
	delete f;
}

static void fifo_args(mi_fifo *f, FifoArgs &a) {
	memset(&a, 0, sizeof(a));
	a.ring = f->d_ring;
	a.pos = f->d_pos;
	a.nstreams = f->nstreams;
	a.capacity = f->capacity;
	a.overflow = f->d_overflow;
}

int mi_fifo_push(mi_fifo *f, const int16_t *d_in, int nsamples, int stride, const int32_t *d_count) {
	MI_CHECK_ARG(f && d_in && nsamples > 0 && stride >= nsamples && nsamples <= f->capacity);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.in = d_in;
	a.count = d_count;
	a.nsamples = nsamples;
	a.stride = stride;
	a.vec = ((f->capacity | stride) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
	hipLaunchKernelGGL(fifo_push_kernel, dim3(mi::ceil_div(f->nstreams, FIFO_WAVES)), dim3(64 * FIFO_WAVES), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_push_gated(mi_fifo *f, const int16_t *d_in, int nsamples, int stride, const uint8_t *d_gate) {
	MI_CHECK_ARG(f && d_in && nsamples > 0 && stride >= nsamples && nsamples <= f->capacity);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.in = d_in;
	a.gate = d_gate;
	a.nsamples = nsamples;
	a.stride = stride;
	a.vec = ((f->capacity | stride) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
	hipLaunchKernelGGL(fifo_push_kernel, dim3(mi::ceil_div(f->nstreams, FIFO_WAVES)), dim3(64 * FIFO_WAVES), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_pop(mi_fifo *f, int frame, int16_t *d_out, int stride, uint8_t *d_ok, const uint8_t *d_gate, int zero_fill) {
	MI_CHECK_ARG(f && d_out && frame > 0 && stride >= frame);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.out = d_out;
	a.ok = d_ok;
	a.gate = d_gate;
	a.nsamples = frame;
	a.stride = stride;
	a.zero_fill = zero_fill;
	a.vec = ((f->capacity | stride) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
	hipLaunchKernelGGL(fifo_pop_kernel, dim3(mi::ceil_div(f->nstreams, FIFO_WAVES)), dim3(64 * FIFO_WAVES), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_push_frames(mi_fifo *f, const int16_t *d_in, int frame, int max_frames, int stride, const uint8_t *d_nframes) {
	MI_CHECK_ARG(f && d_in && d_nframes && frame > 0 && max_frames > 0 && stride >= frame * max_frames && frame * max_frames <= f->capacity);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.in = d_in;
	a.nframes = d_nframes;
	a.max_frames = max_frames;
	a.nsamples = frame;
	a.stride = stride;
	a.vec = ((f->capacity | stride | frame) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
	hipLaunchKernelGGL(fifo_push_kernel, dim3(mi::ceil_div(f->nstreams, FIFO_WAVES)), dim3(64 * FIFO_WAVES), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_pop_frames(mi_fifo *f, int frame, int max_frames, int16_t *d_out, int stride, uint8_t *d_nframes_out,
                       const uint8_t *d_nframes_wanted, int zero_fill) {
	MI_CHECK_ARG(f && d_out && frame > 0 && max_frames > 0 && stride >= frame * max_frames && (d_nframes_out || d_nframes_wanted));
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.out = d_out;
	a.nframes_out = d_nframes_out;
	a.nframes = d_nframes_wanted;
	a.max_frames = max_frames;
	a.nsamples = frame;
	a.stride = stride;
	a.zero_fill = zero_fill;
	a.vec = ((f->capacity | stride | frame) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
	hipLaunchKernelGGL(fifo_pop_frames_kernel, dim3(mi::ceil_div(f->nstreams, FIFO_WAVES)), dim3(64 * FIFO_WAVES), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_phase_of(int stream, int phases) { return phases > 0 ? (int)fifo_phase_of((unsigned)stream, (unsigned)phases) : 0; }

int mi_fifo_push_lead(mi_fifo *f, int first, int count, int unit, int phases) {
	MI_CHECK_ARG(f && first >= 0 && count >= 0 && first + count <= f->nstreams && unit > 0 && phases > 0 && unit * (phases - 1) <= f->capacity);
	if (count == 0) return MI_OK;
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	hipLaunchKernelGGL(fifo_lead_kernel, dim3(mi::ceil_div(count, 256)), dim3(256), 0, f->ctx->stream, a, first, count, unit, phases);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_push_silence(mi_fifo *f, const int32_t *d_count) {
	MI_CHECK_ARG(f && d_count);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.count = d_count;
	hipLaunchKernelGGL(fifo_silence_kernel, dim3(mi::ceil_div(f->nstreams, 256)), dim3(256), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_levels(mi_fifo *f, int32_t *d_levels) {
	MI_CHECK_ARG(f && d_levels);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	FifoArgs a;
	fifo_args(f, a);
	a.levels = d_levels;
	hipLaunchKernelGGL(fifo_level_kernel, dim3(mi::ceil_div(f->nstreams, 256)), dim3(256), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_fifo_snapshot(mi_fifo *f, int16_t *h_rings, int32_t *h_head, int32_t *h_level) {
	MI_CHECK_ARG(f != nullptr);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	if (h_rings) MI_HIP(hipMemcpy(h_rings, f->d_ring, (size_t)f->nstreams * f->capacity * sizeof(int16_t), hipMemcpyDeviceToHost));
	if (h_head || h_level) {
		std::vector<int2> pos((size_t)f->nstreams);
		MI_HIP(hipMemcpy(pos.data(), f->d_pos, pos.size() * sizeof(int2), hipMemcpyDeviceToHost));
		for (int s = 0; s < f->nstreams; ++s) {
			if (h_head) h_head[s] = pos[(size_t)s].x;
			if (h_level) h_level[s] = pos[(size_t)s].y;
		}
	}
	return MI_OK;
}

// The queues of streams [first, first + count) as a host sees MSBufferizers: stream k's h_level[k] samples, oldest first, at h_samples +
// k * stride -- and back.  A conference that is re-plumbed (audioconference.c:322-374) takes what its members' bufferizers hold out of the
// batch and back in: one round trip for all of them (mi_fifo_pop / push, all-or-nothing per piece length, took one per member and length).
int mi_fifo_export_range(mi_fifo *f, int first, int count, int16_t *h_samples, int stride, int32_t *h_level) {
	MI_CHECK_ARG(f && h_samples && h_level && first >= 0 && count >= 0 && first + count <= f->nstreams && stride >= f->capacity);
	if (count == 0) return MI_OK;
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	std::vector<int16_t> rings((size_t)count * f->capacity);
	std::vector<int2> pos((size_t)count);
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	MI_HIP(hipMemcpy(rings.data(), f->d_ring + (size_t)first * f->capacity, rings.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
	MI_HIP(hipMemcpy(pos.data(), f->d_pos + first, pos.size() * sizeof(int2), hipMemcpyDeviceToHost));
	for (int k = 0; k < count; ++k) {
		const int head = pos[(size_t)k].x, level = pos[(size_t)k].y;
		h_level[k] = level;
		for (int i = 0; i < level; ++i) h_samples[(size_t)k * stride + i] = rings[(size_t)k * f->capacity + (size_t)((head + i) % f->capacity)];
	}
	return MI_OK;
}
// tail_at_end: the queue ENDS on the ring's end (head = capacity - level: the canceller's launches append whole frames at a tail they take to
// be frame-aligned, mi_fifo_reset_range_at); levels then are multiples of 8.  Otherwise the queue starts at the ring's start.
int mi_fifo_import_range(mi_fifo *f, int first, int count, const int16_t *h_samples, int stride, const int32_t *h_level, int tail_at_end) {
	MI_CHECK_ARG(f && h_samples && h_level && first >= 0 && count >= 0 && first + count <= f->nstreams);
	if (count == 0) return MI_OK;
	for (int k = 0; k < count; ++k) MI_CHECK_ARG(h_level[k] >= 0 && h_level[k] <= f->capacity && h_level[k] <= stride && (!tail_at_end || (h_level[k] & 7) == 0));
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	std::vector<int16_t> rings((size_t)count * f->capacity, 0);
	std::vector<int2> pos((size_t)count);
	for (int k = 0; k < count; ++k) {
		const int level = h_level[k], head = (tail_at_end && level > 0) ? (f->capacity - level) % f->capacity : 0;
		pos[(size_t)k] = make_int2(head, level);
		for (int i = 0; i < level; ++i) rings[(size_t)k * f->capacity + (size_t)((head + i) % f->capacity)] = h_samples[(size_t)k * stride + i];
	}
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	MI_HIP(hipMemcpy(f->d_ring + (size_t)first * f->capacity, rings.data(), rings.size() * sizeof(int16_t), hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(f->d_pos + first, pos.data(), pos.size() * sizeof(int2), hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_fifo_overflows(mi_fifo *f, int32_t *h_count) {
	MI_CHECK_ARG(f && h_count);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemcpyAsync(h_count, f->d_overflow, sizeof(int32_t), hipMemcpyDeviceToHost, f->ctx->stream));
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	return MI_OK;
}

int mi_fifo_reset_range(mi_fifo *f, int first, int count) {
	MI_CHECK_ARG(f && first >= 0 && count >= 0 && first + count <= f->nstreams);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	if (count) MI_HIP(hipMemsetAsync(f->d_pos + first, 0, (size_t)count * sizeof(int2), f->ctx->stream));
	return MI_OK;
}

int mi_fifo_reset_range_at(mi_fifo *f, int first, int count, int head) {
	MI_CHECK_ARG(f && first >= 0 && count >= 0 && first + count <= f->nstreams && head >= 0 && head < f->capacity && (head & 7) == 0);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	if (count) {
		hipLaunchKernelGGL(fifo_rewind_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, f->ctx->stream, f->d_pos + first, count, head);
		MI_LAUNCH_CHECK();
	}
	return MI_OK;
}

int mi_fifo_reset(mi_fifo *f) {
	MI_CHECK_ARG(f != nullptr);
	if (f->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemsetAsync(f->d_pos, 0, (size_t)f->nstreams * sizeof(int2), f->ctx->stream));
	MI_HIP(hipMemsetAsync(f->d_overflow, 0, sizeof(int32_t), f->ctx->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_fifo(reinterpret_cast<const void *>(&fifo_silence_kernel));
