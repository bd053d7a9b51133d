// codec.hip -- the per-stream stages that sit either side of the hot path in every AudioStream graph
// (src/voip/audiostream.c:1798-1832), batched for gfx950: G.711 A-law / mu-law decode and encode, L16 byte
// order, MSChannelAdapter's mono <-> stereo loops, and MSAudioFlowControl's sample dropper.
//
//   MSAlawDec / MSUlawDec   alaw.c:208-221, ulaw.c (same)  -> g711_decode_kernel   Snack_Alaw2Lin / Snack_Mulaw2Lin  g711.c:147-166,:242-255
//   MSAlawEnc / MSUlawEnc   alaw.c:77-82,  ulaw.c:78-82    -> g711_encode_kernel   Snack_Lin2Alaw / Snack_Lin2Mulaw  g711.c:113-141,:200-231
//   MSL16Enc / MSL16Dec     l16.c:58-70                    -> l16_swap_kernel
//   MSChannelAdapter        chanadapt.c:87-90,:110-121     -> chan_adapt_kernel
//   MSAudioFlowControl      flowcontrol.c:56-152           -> flowctl_kernel
//
// All integer / byte work, bit-exact.  The conversions are pure streaming (1 B <-> 2 B per sample, HBM-bound):
// one lane owns 16 consecutive samples of one row = one 16-byte load (or two) and two (or one) 16-byte stores,
// consecutive lanes consecutive groups, rows x groups flattened over a grid-stride loop.  The segment search of
// the encoders (a table scan in the reference, g711.c:80-87) is a count-leading-zeros; the decoders work on two
// samples per 32-bit register with the packed 16-bit VALU ops (v_pk_lshlrev_b16 carries a per-half shift count).
// The flow controller is a per-stream state machine: one wavefront per stream, the block in LDS, a wave-wide
// arg-min per deleted sample.
#include "common.hpp"

namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ us2 splat(unsigned short v) { return us2{v, v}; }

// two A-law codes (one per 16-bit half, 0..255) -> two int16 samples.  g711.c:147-166
__device__ __forceinline__ uint32_t alaw2lin_x2(uint32_t codes) {
	const us2 a = as_us2(codes ^ 0x00550055u);
	const us2 seg = (a >> splat(4)) & splat(7);
	const us2 lin = __builtin_elementwise_min(seg, splat(1));      // 0 in the first (linear) segment, else 1
	const us2 mant = ((a & splat(15)) << splat(4)) + splat(8) + (lin << splat(8)); // +8, or +0x108
	const us2 mag = mant << (seg - lin);                            // seg 0,1: no shift; seg k: k-1
	const us2 neg = ((a >> splat(7)) & splat(1)) ^ splat(1);       // sign bit SET means positive
	const us2 m = splat(0) - neg;                                   // 0xFFFF where negative
	return as_u32((mag ^ m) + neg);
}

// two mu-law codes -> two int16 samples.  g711.c:242-255
__device__ __forceinline__ uint32_t ulaw2lin_x2(uint32_t codes) {
	const us2 u = as_us2(codes ^ 0x00ff00ffu);
	const us2 mag = ((((u & splat(15)) << splat(3)) + splat(0x84)) << ((u >> splat(4)) & splat(7))) - splat(0x84);
	const us2 neg = (u >> splat(7)) & splat(1);
	const us2 m = splat(0) - neg;
	return as_u32((mag ^ m) + neg);
}

// g711.c:113-141: 13-bit magnitude, segment = position of the leading one
__device__ __forceinline__ uint32_t lin2alaw(int pcm) {
	int v = pcm >> 3;
	const int sign = v >> 31; // -1 for negative input
	v ^= sign;                // -v - 1
	const int seg = max(27 - __clz(v), 0); // bit_length - 5; v <= 4095 so seg <= 7
	const int mant = (v >> max(seg, 1)) & 15;
	return (uint32_t)(((seg << 4) | mant) ^ (0xD5 ^ (sign & 0x80)));
}

// g711.c:200-231: 14-bit magnitude clipped at 8159, bias 33; the clipped maximum overflows the table (seg 8)
__device__ __forceinline__ uint32_t lin2ulaw(int pcm) {
	int v = pcm >> 2;
	const int sign = v >> 31;
	v = min((v ^ sign) - sign, 8159) + 33;
	const int seg = max(26 - __clz(v), 0); // bit_length - 6
	const int code = seg >= 8 ? 0x7F : ((seg << 4) | ((v >> (seg + 1)) & 15));
	return (uint32_t)(code ^ (0xFF ^ (sign & 0x80)));
}

struct MapArgs {
	const void *in;
	const void *in2;
	void *out;
	size_t in_stride, out_stride; // elements of the respective type
	const int32_t *len;           // per-row sample count, or null = `n` for every row
	int n, groups, vec;           // groups of 16 samples per row
	size_t rows;
};

constexpr int MAP_THREADS = 256;

template <int LAW>
__global__ __launch_bounds__(MAP_THREADS) void g711_decode_kernel(MapArgs a) {
	const size_t total = a.rows * (size_t)a.groups;
	for (size_t idx = (size_t)blockIdx.x * MAP_THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * MAP_THREADS) {
		const size_t row = idx / (size_t)a.groups;
		const int g = (int)(idx - row * (size_t)a.groups);
		const int n = a.len ? min(max(a.len[row], 0), a.n) : a.n;
		const int first = 16 * g;
		if (first >= n) continue;
		const uint8_t *src = (const uint8_t *)a.in + row * a.in_stride + first;
		int16_t *dst = (int16_t *)a.out + row * a.out_stride + first;
		if (a.vec && first + 16 <= n) {
			const uint4 c = *reinterpret_cast<const uint4 *>(src);
			const uint32_t w[4] = {c.x, c.y, c.z, c.w};
			uint32_t o[8];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t lo = __builtin_amdgcn_perm(0u, w[k], 0x0c010c00u); // bytes 0,1 -> the two halves
				const uint32_t hi = __builtin_amdgcn_perm(0u, w[k], 0x0c030c02u); // bytes 2,3
				o[2 * k] = LAW ? ulaw2lin_x2(lo) : alaw2lin_x2(lo);
				o[2 * k + 1] = LAW ? ulaw2lin_x2(hi) : alaw2lin_x2(hi);
			}
			reinterpret_cast<uint4 *>(dst)[0] = make_uint4(o[0], o[1], o[2], o[3]);
			reinterpret_cast<uint4 *>(dst)[1] = make_uint4(o[4], o[5], o[6], o[7]);
		} else {
			const int m = min(16, n - first);
			for (int i = 0; i < m; ++i) {
				const uint32_t r = LAW ? ulaw2lin_x2(src[i]) : alaw2lin_x2(src[i]);
				dst[i] = (int16_t)(r & 0xffffu);
			}
		}
	}
}

template <int LAW>
__global__ __launch_bounds__(MAP_THREADS) void g711_encode_kernel(MapArgs a) {
	const size_t total = a.rows * (size_t)a.groups;
	for (size_t idx = (size_t)blockIdx.x * MAP_THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * MAP_THREADS) {
		const size_t row = idx / (size_t)a.groups;
		const int g = (int)(idx - row * (size_t)a.groups);
		const int n = a.len ? min(max(a.len[row], 0), a.n) : a.n;
		const int first = 16 * g;
		if (first >= n) continue;
		const int16_t *src = (const int16_t *)a.in + row * a.in_stride + first;
		uint8_t *dst = (uint8_t *)a.out + row * a.out_stride + first;
		if (a.vec && first + 16 <= n) {
			const uint4 p0 = reinterpret_cast<const uint4 *>(src)[0], p1 = reinterpret_cast<const uint4 *>(src)[1];
			const uint32_t w[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
			uint32_t o[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				uint32_t acc = 0;
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint32_t pair = w[2 * k + (j >> 1)];
					const int s = (j & 1) ? (int)pair >> 16 : (int)(pair << 16) >> 16;
					acc |= (LAW ? lin2ulaw(s) : lin2alaw(s)) << (8 * j);
				}
				o[k] = acc;
			}
			*reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
		} else {
			const int m = min(16, n - first);
			for (int i = 0; i < m; ++i) dst[i] = (uint8_t)(LAW ? lin2ulaw(src[i]) : lin2alaw(src[i]));
		}
	}
}

// l16.c:58-70: htons / ntohs of every sample; 8 samples (16 bytes) per lane
__global__ __launch_bounds__(MAP_THREADS) void l16_swap_kernel(const int16_t *in, int16_t *out, size_t n, int vec) {
	const size_t groups = (n + 7) / 8;
	for (size_t g = (size_t)blockIdx.x * MAP_THREADS + threadIdx.x; g < groups; g += (size_t)gridDim.x * MAP_THREADS) {
		const size_t first = 8 * g;
		if (vec && first + 8 <= n) {
			uint4 v = *reinterpret_cast<const uint4 *>(in + first);
			v.x = __builtin_amdgcn_perm(0u, v.x, 0x02030001u);
			v.y = __builtin_amdgcn_perm(0u, v.y, 0x02030001u);
			v.z = __builtin_amdgcn_perm(0u, v.z, 0x02030001u);
			v.w = __builtin_amdgcn_perm(0u, v.w, 0x02030001u);
			*reinterpret_cast<uint4 *>(out + first) = v;
		} else {
			for (size_t i = first; i < min(first + 8, n); ++i) {
				const uint16_t s = (uint16_t)in[i];
				out[i] = (int16_t)(uint16_t)((s << 8) | (s >> 8));
			}
		}
	}
}

// chanadapt.c: MODE 0 mono -> stereo (:110-113), 1 stereo -> mono keeping the left sample (:118-121),
// 2 two mono rows -> interleaved stereo, a missing side is silence (:81-90).  8 frames per lane.
template <int MODE>
__global__ __launch_bounds__(MAP_THREADS) void chan_adapt_kernel(const int16_t *a, const int16_t *b, int16_t *out, size_t frames,
                                                                  int vec) {
	const size_t groups = (frames + 7) / 8;
	for (size_t g = (size_t)blockIdx.x * MAP_THREADS + threadIdx.x; g < groups; g += (size_t)gridDim.x * MAP_THREADS) {
		const size_t first = 8 * g;
		if (vec && first + 8 <= frames) {
			if (MODE == 1) {
				const uint4 p0 = reinterpret_cast<const uint4 *>(a + 2 * first)[0], p1 = reinterpret_cast<const uint4 *>(a + 2 * first)[1];
				uint4 o; // low halves of consecutive dwords
				o.x = __builtin_amdgcn_perm(p0.y, p0.x, 0x05040100u);
				o.y = __builtin_amdgcn_perm(p0.w, p0.z, 0x05040100u);
				o.z = __builtin_amdgcn_perm(p1.y, p1.x, 0x05040100u);
				o.w = __builtin_amdgcn_perm(p1.w, p1.z, 0x05040100u);
				*reinterpret_cast<uint4 *>(out + first) = o;
			} else {
				const uint4 l = *reinterpret_cast<const uint4 *>(a + first);
				const uint4 r = MODE == 0 ? l : (b ? *reinterpret_cast<const uint4 *>(b + first) : make_uint4(0, 0, 0, 0));
				const uint32_t lw[4] = {l.x, l.y, l.z, l.w}, rw[4] = {r.x, r.y, r.z, r.w};
				uint32_t o[8];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					o[2 * k] = __builtin_amdgcn_perm(rw[k], lw[k], 0x05040100u);     // (left lo, right lo)
					o[2 * k + 1] = __builtin_amdgcn_perm(rw[k], lw[k], 0x07060302u); // (left hi, right hi)
				}
				reinterpret_cast<uint4 *>(out + 2 * first)[0] = make_uint4(o[0], o[1], o[2], o[3]);
				reinterpret_cast<uint4 *>(out + 2 * first)[1] = make_uint4(o[4], o[5], o[6], o[7]);
			}
		} else {
			for (size_t i = first; i < min(first + 8, frames); ++i) {
				if (MODE == 1) out[i] = a[2 * i];
				else {
					out[2 * i] = a[i];
					out[2 * i + 1] = MODE == 0 ? a[i] : (b ? b[i] : (int16_t)0);
				}
			}
		}
	}
}

int map_blocks(const mi_ctx *c, size_t items) {
	const size_t need = (items + MAP_THREADS - 1) / MAP_THREADS;
	const size_t cap = (size_t)(c->cu_count > 0 ? c->cu_count : 256) * 16;
	return (int)std::max<size_t>(1, std::min(need, cap));
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// ------------------------------------------------------------------------------------ flow controller
struct FlowState { // MSAudioFlowController, include/mediastreamer2/flowcontrol.h:40-46
	uint32_t target, total, pos, dropped;
};
struct FlowCfg { // MSAudioFlowControlConfig :33-38
	int32_t strategy;
	float silent_threshold;
};

struct FlowArgs {
	FlowState *st;
	const FlowCfg *cfg;
	const uint2 *arm; // per stream (samples_to_drop, total_samples) requests, x = 0: none
	const int16_t *in;
	int16_t *out;
	const int32_t *len;
	int32_t *out_len;
	size_t in_stride, out_stride;
	int n, nstreams, cap;
};

constexpr int FLOW_WAVES = 4;
constexpr int FLOW_MAX_BLOCK = 2048; // samples per block: the arg-min key keeps 12 bits of index, a lane 32 samples in registers

__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ms_audio_flow_control_drop, flowcontrol.c:209-219: a request is taken only while no drop is in progress
__global__ void flowctl_arm_kernel(FlowArgs a) {
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= a.nstreams) return;
	const uint2 r = a.arm[s];
	if (r.x == 0 && r.y == 0) return;
	FlowState st = a.st[s];
	if (st.total > 0 && st.target > 0) return;
	a.st[s] = FlowState{r.x, r.y, 0u, 0u}; // ms_audio_flow_controller_set_target :49-54
}

// ms_audio_flow_controller_process, flowcontrol.c:107-152, one wavefront per stream
__global__ __launch_bounds__(64 * FLOW_WAVES) void flowctl_kernel(FlowArgs a) {
	extern __shared__ int16_t lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int s = blockIdx.x * FLOW_WAVES + wave;
	if (s >= a.nstreams) return;
	int16_t *blk = lds + (size_t)wave * (a.cap + 2);
	const int n = a.len ? min(max(a.len[s], 0), a.n) : a.n;
	const int16_t *src = a.in + (size_t)s * a.in_stride;
	int16_t *dst = a.out + (size_t)s * a.out_stride;
	if (n == 0) { // no block from this stream in this round
		if (lane == 0) a.out_len[s] = 0;
		return;
	}
	FlowState st = a.st[s];
	int left = n;
	const bool running = st.total > 0 && st.target > 0;
	bool edit = false;
	if (running) {
		const FlowCfg cfg = a.cfg[s];
		const uint32_t nsamples = (uint32_t)n;
		st.pos += nsamples;
		if (cfg.strategy == 0) { // MSAudioFlowControlBasic :115-121
			if (st.dropped + nsamples <= st.target) {
				st.dropped += nsamples;
				left = 0;
			}
		} else {
			const uint32_t th = (uint32_t)(((uint64_t)st.target * (uint64_t)st.pos) / (uint64_t)st.total);
			uint32_t todrop = th > st.dropped ? th - st.dropped : 0u;
			if (todrop > 0) {
				for (int i = lane; i < n; i += 64) blk[i] = src[i];
				wave_sync();
				bool silent = false;
				if (nsamples <= st.target) { // compute_frame_power :97-105: float sum in sample order (every lane the same)
					float acc = 0.f;
					for (int i = 0; i < n; ++i) {
						const int v = blk[i];
						acc += (float)(v * v);
					}
					const float p = sqrtf(acc / (float)nsamples) / (32768 * 0.7f); // correctly rounded sqrt and divides (hipcc default)
					silent = p < cfg.silent_threshold;
				}
				if (silent) {
					left = 0;
					todrop = nsamples;
				} else if (todrop * 8u < nsamples) { // discard_well_choosed_samples :56-89
					edit = true;
					for (uint32_t d = 0; d < todrop; ++d) {
						// the LAST i in [0, left-2) minimising |s[i]-s[i+1]| + |s[i+1]-s[i+2]|  (<=, start value 32768)
						uint32_t best = 0xffffffffu;
						for (int i = lane; i + 2 < left; i += 64) {
							const int x0 = blk[i], x1 = blk[i + 1], x2 = blk[i + 2];
							const uint32_t diff = (uint32_t)(abs(x0 - x1) + abs(x1 - x2));
							best = min(best, (diff << 12) | (uint32_t)(4095 - i));
						}
#pragma unroll
						for (int off = 32; off; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
						const int pos = (best >> 12) <= 32768u ? 4095 - (int)(best & 4095u) : 0;
						// delete blk[pos+1]: every lane reads its successors first, then writes
						int16_t tmp[FLOW_MAX_BLOCK / 64];
#pragma unroll
						for (int k = 0; k < FLOW_MAX_BLOCK / 64; ++k) {
							const int i = pos + 1 + lane + 64 * k;
							if (i + 1 < left) tmp[k] = blk[i + 1];
						}
						wave_sync();
#pragma unroll
						for (int k = 0; k < FLOW_MAX_BLOCK / 64; ++k) {
							const int i = pos + 1 + lane + 64 * k;
							if (i + 1 < left) blk[i] = tmp[k];
						}
						wave_sync();
						--left;
					}
				} else {
					left = 0; // :137-142
					todrop = nsamples;
				}
				st.dropped += todrop;
			}
		}
		if (st.pos >= st.total) st.target = 0; // :149
		if (lane == 0) a.st[s] = st;
	}
	if (edit) {
		for (int i = lane; i < left; i += 64) dst[i] = blk[i];
	} else if (left && dst != src) {
		for (int i = lane; i < left; i += 64) dst[i] = src[i];
	}
	if (lane == 0) a.out_len[s] = left;
}

} // namespace

struct mi_flowctl {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, cap = 0;
	FlowState *d_state = nullptr;
	FlowCfg *d_cfg = nullptr;
	uint2 *d_arm = nullptr;
	std::vector<FlowCfg> h_cfg;
};

extern "C" {

static int g711_args(mi_ctx *c, const void *in, size_t in_stride, void *out, size_t out_stride, const int32_t *d_len, int len,
                     size_t rows, MapArgs *a) {
	MI_CHECK_ARG(c && in && out);
	MI_CHECK_ARG(len >= 0 && (rows <= 1 || (in_stride >= (size_t)len && out_stride >= (size_t)len)));
	a->in = in, a->in2 = nullptr, a->out = out;
	a->in_stride = in_stride, a->out_stride = out_stride;
	a->len = d_len, a->n = len, a->groups = (len + 15) / 16, a->rows = rows;
	return MI_OK;
}

int mi_g711_decode(mi_ctx *c, int law, const uint8_t *d_codes, size_t codes_stride, int16_t *d_pcm, size_t pcm_stride,
                   const int32_t *d_len, int len, size_t rows) {
	MapArgs a;
	int rc = g711_args(c, d_codes, codes_stride, d_pcm, pcm_stride, d_len, len, rows, &a);
	if (rc != MI_OK) return rc;
	MI_CHECK_ARG(law == MI_LAW_PCMA || law == MI_LAW_PCMU);
	if (rows == 0 || len == 0) return MI_OK;
	if ((rc = c->activate()) != MI_OK) return rc;
	a.vec = aligned16(d_codes) && aligned16(d_pcm) && codes_stride % 16 == 0 && pcm_stride % 8 == 0;
	const int blocks = map_blocks(c, rows * (size_t)a.groups);
	if (law == MI_LAW_PCMA) hipLaunchKernelGGL(g711_decode_kernel<0>, dim3(blocks), dim3(MAP_THREADS), 0, c->stream, a);
	else hipLaunchKernelGGL(g711_decode_kernel<1>, dim3(blocks), dim3(MAP_THREADS), 0, c->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_g711_encode(mi_ctx *c, int law, const int16_t *d_pcm, size_t pcm_stride, uint8_t *d_codes, size_t codes_stride,
                   const int32_t *d_len, int len, size_t rows) {
	MapArgs a;
	int rc = g711_args(c, d_pcm, pcm_stride, d_codes, codes_stride, d_len, len, rows, &a);
	if (rc != MI_OK) return rc;
	MI_CHECK_ARG(law == MI_LAW_PCMA || law == MI_LAW_PCMU);
	if (rows == 0 || len == 0) return MI_OK;
	if ((rc = c->activate()) != MI_OK) return rc;
	a.vec = aligned16(d_codes) && aligned16(d_pcm) && codes_stride % 16 == 0 && pcm_stride % 8 == 0;
	const int blocks = map_blocks(c, rows * (size_t)a.groups);
	if (law == MI_LAW_PCMA) hipLaunchKernelGGL(g711_encode_kernel<0>, dim3(blocks), dim3(MAP_THREADS), 0, c->stream, a);
	else hipLaunchKernelGGL(g711_encode_kernel<1>, dim3(blocks), dim3(MAP_THREADS), 0, c->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_l16_swap(mi_ctx *c, const int16_t *d_in, int16_t *d_out, size_t nsamples) {
	MI_CHECK_ARG(c && d_in && d_out);
	if (nsamples == 0) return MI_OK;
	int rc;
	if ((rc = c->activate()) != MI_OK) return rc;
	const int vec = aligned16(d_in) && aligned16(d_out);
	hipLaunchKernelGGL(l16_swap_kernel, dim3(map_blocks(c, (nsamples + 7) / 8)), dim3(MAP_THREADS), 0, c->stream, d_in, d_out, nsamples, vec);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_chan_adapt(mi_ctx *c, int mode, const int16_t *d_a, const int16_t *d_b, int16_t *d_out, size_t frames) {
	MI_CHECK_ARG(c && d_a && d_out);
	MI_CHECK_ARG(mode == MI_CHAN_MONO_TO_STEREO || mode == MI_CHAN_STEREO_TO_MONO || mode == MI_CHAN_TWO_MONO_TO_STEREO);
	if (frames == 0) return MI_OK;
	int rc;
	if ((rc = c->activate()) != MI_OK) return rc;
	const int vec = aligned16(d_a) && aligned16(d_out) && (d_b == nullptr || aligned16(d_b));
	const dim3 grid(map_blocks(c, (frames + 7) / 8)), block(MAP_THREADS);
	if (mode == MI_CHAN_MONO_TO_STEREO) hipLaunchKernelGGL(chan_adapt_kernel<0>, grid, block, 0, c->stream, d_a, d_b, d_out, frames, vec);
	else if (mode == MI_CHAN_STEREO_TO_MONO) hipLaunchKernelGGL(chan_adapt_kernel<1>, grid, block, 0, c->stream, d_a, d_b, d_out, frames, vec);
	else hipLaunchKernelGGL(chan_adapt_kernel<2>, grid, block, 0, c->stream, d_a, d_b, d_out, frames, vec);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

void mi_flowctl_destroy(mi_flowctl *f);

int mi_flowctl_create(mi_ctx *c, int nstreams, int max_block, mi_flowctl **out) {
	MI_CHECK_ARG(c && out && nstreams > 0);
	MI_CHECK_ARG(max_block >= 3 && max_block <= FLOW_MAX_BLOCK);
	int rc;
	if ((rc = c->activate()) != MI_OK) return rc;
	mi_flowctl *f = new mi_flowctl();
	f->ctx = c, f->nstreams = nstreams, f->cap = max_block;
	f->h_cfg.assign((size_t)nstreams, FlowCfg{1, 0.02f}); // ms_audio_flow_controller_init :37-41
	const size_t n = (size_t)nstreams;
	if (hipMalloc(&f->d_state, sizeof(FlowState) * n) != hipSuccess || hipMalloc(&f->d_cfg, sizeof(FlowCfg) * n) != hipSuccess ||
	    hipMalloc(&f->d_arm, sizeof(uint2) * n) != hipSuccess ||
	    hipMemsetAsync(f->d_state, 0, sizeof(FlowState) * n, c->stream) != hipSuccess ||
	    hipMemcpyAsync(f->d_cfg, f->h_cfg.data(), sizeof(FlowCfg) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
	    hipStreamSynchronize(c->stream) != hipSuccess) {
		mi::set_error("mi_flowctl_create: device allocation / initialisation failed for %d streams", nstreams);
		mi_flowctl_destroy(f);
		return MI_ENOMEM;
	}
	*out = f;
	return MI_OK;
}

void mi_flowctl_destroy(mi_flowctl *f) {
	if (!f) return;
	if (f->ctx->activate() == MI_OK) {
		(void)hipStreamSynchronize(f->ctx->stream);
		(void)hipFree(f->d_state);
		(void)hipFree(f->d_cfg);
		(void)hipFree(f->d_arm);
	}
	delete f;
}

int mi_flowctl_set_config(mi_flowctl *f, int first, int count, int strategy, float silent_threshold) {
	MI_CHECK_ARG(f && first >= 0 && count >= 0 && first + count <= f->nstreams);
	MI_CHECK_ARG(strategy == MI_FLOWCTL_BASIC || strategy == MI_FLOWCTL_SOFT);
	if (count == 0) return MI_OK;
	int rc;
	if ((rc = f->ctx->activate()) != MI_OK) return rc;
	for (int i = 0; i < count; ++i) f->h_cfg[(size_t)(first + i)] = FlowCfg{strategy, silent_threshold};
	MI_HIP(hipMemcpyAsync(f->d_cfg + first, f->h_cfg.data() + first, sizeof(FlowCfg) * (size_t)count, hipMemcpyHostToDevice, f->ctx->stream));
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	return MI_OK;
}

int mi_flowctl_request_drop(mi_flowctl *f, const uint32_t *h_samples_to_drop, const uint32_t *h_total_samples) {
	MI_CHECK_ARG(f && h_samples_to_drop && h_total_samples);
	int rc;
	if ((rc = f->ctx->activate()) != MI_OK) return rc;
	std::vector<uint2> req((size_t)f->nstreams);
	for (int s = 0; s < f->nstreams; ++s) req[(size_t)s] = make_uint2(h_samples_to_drop[s], h_total_samples[s]);
	MI_HIP(hipMemcpyAsync(f->d_arm, req.data(), sizeof(uint2) * req.size(), hipMemcpyHostToDevice, f->ctx->stream));
	FlowArgs a{};
	a.st = f->d_state, a.arm = f->d_arm, a.nstreams = f->nstreams;
	hipLaunchKernelGGL(flowctl_arm_kernel, dim3((f->nstreams + 255) / 256), dim3(256), 0, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	MI_HIP(hipStreamSynchronize(f->ctx->stream)); // req is a stack-lifetime buffer
	return MI_OK;
}

int mi_flowctl_process(mi_flowctl *f, const int16_t *d_in, size_t in_stride, const int32_t *d_len, int len, int16_t *d_out,
                       size_t out_stride, int32_t *d_out_len) {
	MI_CHECK_ARG(f && d_in && d_out && d_out_len);
	MI_CHECK_ARG(len >= 0 && len <= f->cap && in_stride >= (size_t)len && out_stride >= (size_t)len);
	int rc;
	if ((rc = f->ctx->activate()) != MI_OK) return rc;
	FlowArgs a{};
	a.st = f->d_state, a.cfg = f->d_cfg, a.in = d_in, a.out = d_out, a.len = d_len, a.out_len = d_out_len;
	a.in_stride = in_stride, a.out_stride = out_stride, a.n = len, a.nstreams = f->nstreams, a.cap = f->cap;
	const size_t lds = (size_t)FLOW_WAVES * (size_t)(f->cap + 2) * sizeof(int16_t);
	hipLaunchKernelGGL(flowctl_kernel, dim3((f->nstreams + FLOW_WAVES - 1) / FLOW_WAVES), dim3(64 * FLOW_WAVES), lds, f->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_flowctl_get_state(mi_flowctl *f, int stream, uint32_t out4[4]) {
	MI_CHECK_ARG(f && out4 && stream >= 0 && stream < f->nstreams);
	int rc;
	if ((rc = f->ctx->activate()) != MI_OK) return rc;
	FlowState st;
	MI_HIP(hipMemcpyAsync(&st, f->d_state + stream, sizeof(st), hipMemcpyDeviceToHost, f->ctx->stream));
	MI_HIP(hipStreamSynchronize(f->ctx->stream));
	out4[0] = st.target, out4[1] = st.total, out4[2] = st.pos, out4[3] = st.dropped;
	return MI_OK;
}

int mi_flowctl_reset(mi_flowctl *f, int first, int count) { // ms_audio_flow_controller_reset :30-35 (preprocess)
	MI_CHECK_ARG(f && first >= 0 && count >= 0 && first + count <= f->nstreams);
	if (count == 0) return MI_OK;
	int rc;
	if ((rc = f->ctx->activate()) != MI_OK) return rc;
	MI_HIP(hipMemsetAsync(f->d_state + first, 0, sizeof(FlowState) * (size_t)count, f->ctx->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_codec(reinterpret_cast<const void *>(&flowctl_arm_kernel));
