// mixer.hip -- batched MSAudioMixer ("msconf") tick for gfx950.
//
// One launch mixes one 10 ms tick of `nconf` conferences.  It is the loop nest
// of mixer_process (src/audiofilters/audiomixer.c:301-344): per channel
// channel_process_in (:78-90: short read -> zeros, optional in-place gain
// :46-51, int32 accumulate :33-38) and, in conference mode, per output
// channel_process_out (:113-130: saturate(sum - own contribution), symmetric
// +-32767 clamp :40-44).  Pure integer; bit-exact by construction.
//
// Mapping: one lane owns 4 consecutive samples (one 8-byte load per member)
// of one conference and keeps every member's (gained) contribution in VGPRs,
// so each input byte is read from HBM once and each output byte written once:
// algorithmic traffic = 2 * members * nsamples * 2 B per conference-tick.
// Loads of one member by consecutive lanes are contiguous (512 B per wave).
//
// The split form (partial_sum / finalize) is for conferences whose members are
// sharded over several GPUs: int32 partial sums are all-reduced by the caller
// (RCCL) between the two kernels; integer addition is associative, so the
// result is bit-identical to the single-GPU kernel.
#include "common.hpp"

namespace {

__device__ __forceinline__ int sat16(int s) { return max(-32767, min(32767, s)); }

struct MixArgs {
	const int16_t *in;       // [nconf][mm][ns]
	const uint8_t *has_data; // [nconf][mm] or null
	const uint8_t *flags;    // [nconf][mm]
	const float *gain;       // [nconf][mm]
	int16_t *out;
	int32_t *sum_out;      // partial mode
	const int32_t *sum_in; // finalize mode
	const uint8_t *run;       // [nconf] or null
	const uint8_t *conf_modes; // [nconf] or null
	int nconf, mm, ns, quads, conf_mode, out_conf_stride;
};

__device__ __forceinline__ int4 widen(const short4 v) { return make_int4(v.x, v.y, v.z, v.w); }

// contribution of one channel as channel_process_in stores it
__device__ __forceinline__ int4 load_contrib(const MixArgs &a, int c, int m, int q, unsigned f, bool &summed) {
	summed = false;
	const int cm = c * a.mm + m;
	const bool present = (f & MI_MIX_LINKED) && (a.has_data == nullptr || a.has_data[cm] != 0);
	if (!present) return make_int4(0, 0, 0, 0);
	int4 v = widen(*reinterpret_cast<const short4 *>(a.in + ((size_t)cm * a.ns) + 4 * q));
	if (f & MI_MIX_ACTIVE) {
		const float g = a.gain[cm];
		if (g != 1.0f) {
			v.x = sat16((int)(g * (float)v.x));
			v.y = sat16((int)(g * (float)v.y));
			v.z = sat16((int)(g * (float)v.z));
			v.w = sat16((int)(g * (float)v.w));
		}
		summed = true;
	}
	return v;
}

__device__ __forceinline__ void store_sat(int16_t *dst, int4 s) {
	short4 o;
	o.x = (short)sat16(s.x);
	o.y = (short)sat16(s.y);
	o.z = (short)sat16(s.z);
	o.w = (short)sat16(s.w);
	*reinterpret_cast<short4 *>(dst) = o;
}

// MODE 0: fused tick. MODE 1: partial sums only. MODE 2: outputs from sum_in.
template <int NMAX, int MODE>
__global__ __launch_bounds__(256) void mixer_kernel(MixArgs a) {
	const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
	if (g >= (long long)a.nconf * a.quads) return;
	const int c = (int)(g / a.quads);
	const int q = (int)(g - (long long)c * a.quads);
	if (a.run && !a.run[c]) return;
	const int conf_mode = a.conf_modes ? a.conf_modes[c] : a.conf_mode;
	const uint8_t *fl = a.flags + (size_t)c * a.mm;

	int4 sum = make_int4(0, 0, 0, 0);
	int4 v[NMAX];
	if (MODE == 2) {
		sum = *reinterpret_cast<const int4 *>(a.sum_in + (size_t)c * a.ns + 4 * q);
	} else {
#pragma unroll
		for (int m = 0; m < NMAX; ++m) {
			v[m] = make_int4(0, 0, 0, 0);
			if (m < a.mm) {
				bool summed;
				const int4 x = load_contrib(a, c, m, q, fl[m], summed);
				if (summed) {
					v[m] = x;
					sum.x += x.x;
					sum.y += x.y;
					sum.z += x.z;
					sum.w += x.w;
				}
			}
		}
	}
	if (MODE == 1) {
		*reinterpret_cast<int4 *>(a.sum_out + (size_t)c * a.ns + 4 * q) = sum;
		return;
	}
	if (conf_mode == 0) {
		store_sat(a.out + (size_t)c * a.out_conf_stride + 4 * q, sum);
		return;
	}
	if (MODE == 2) {
		// outputs from the all-reduced sum: every local member's own (gained) contribution is re-derived
		for (int m = 0; m < a.mm; ++m) {
			const unsigned f = fl[m];
			if (f & MI_MIX_OUTPUT) {
				bool summed;
				int4 own = load_contrib(a, c, m, q, f, summed);
				if (!summed) own = make_int4(0, 0, 0, 0);
				store_sat(a.out + ((size_t)(c * a.mm + m) * a.ns) + 4 * q,
				          make_int4(sum.x - own.x, sum.y - own.y, sum.z - own.z, sum.w - own.w));
			}
		}
		return;
	}
#pragma unroll
	for (int m = 0; m < NMAX; ++m) {
		if (m < a.mm) {
			const unsigned f = fl[m];
			if (f & MI_MIX_OUTPUT) {
				const int4 own = v[m]; // zero unless the channel was summed (active)
				store_sat(a.out + ((size_t)(c * a.mm + m) * a.ns) + 4 * q,
				          make_int4(sum.x - own.x, sum.y - own.y, sum.z - own.z, sum.w - own.w));
			}
		}
	}
}


// Member-parallel form of the fused tick for conferences of 8 or more channels.  A block owns one conference
// and `gb` groups of 8 consecutive samples; thread (member, group) loads its channel's 16 bytes once, the
// block reduces the (gained) contributions over members through LDS in two integer steps, and the same
// thread stores saturate(sum - own) as 16 bytes.  128 conferences x 32 members x 480 samples become 1024
// blocks of 256 lanes instead of 60, with the same one-read-one-write HBM traffic.
struct MixMArgs {
	MixArgs a;
	int gb;    // sample groups per block
	int ngrp;  // groups per conference = ns / 8
	int cblocks; // blocks per conference
	int q;     // member ranges in the first reduction step
	int per;   // members per range
};

__global__ __launch_bounds__(256) void mixer_members_kernel(MixMArgs ma) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const MixArgs &a = ma.a;
	// the column blocks of a conference (and the conferences next to it) on one XCD: a member row of 960 bytes starts in
	// the middle of a cache line every other time, and the neighbouring column block needs the other half of that line
	const unsigned item = mi::xcd_item(blockIdx.x, gridDim.x);
	if (item >= (unsigned)(a.nconf * ma.cblocks)) return;
	const int c = item / ma.cblocks;
	const int g0 = (item - c * ma.cblocks) * ma.gb;
	if (a.run && !a.run[c]) return;
	const int conf_mode = a.conf_modes ? a.conf_modes[c] : a.conf_mode;
	const int ncol = ma.gb * 8;
	int16_t *tile = reinterpret_cast<int16_t *>(smem);                          // [mm][ncol] contributions
	int32_t *part = reinterpret_cast<int32_t *>(smem + (size_t)a.mm * ncol * 2); // [q][ncol]
	int32_t *total = part + ma.q * ncol;                                         // [ncol]

	const int t = threadIdx.x;
	const int m = t / ma.gb, gl = t - m * ma.gb;
	const int g = g0 + gl;
	const bool mine = m < a.mm && g < ma.ngrp;
	unsigned f = 0;
	int own[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	bool summed = false;
	if (m < a.mm) {
		const int cm = c * a.mm + m;
		f = a.flags[cm];
		const bool present = (f & MI_MIX_LINKED) && (a.has_data == nullptr || a.has_data[cm] != 0);
		if (mine && present && (f & MI_MIX_ACTIVE)) {
			const uint4 raw = *reinterpret_cast<const uint4 *>(a.in + (size_t)cm * a.ns + 8 * g);
			const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				own[2 * i] = (int)(short)(w[i] & 0xffffu);
				own[2 * i + 1] = (int)(short)(w[i] >> 16);
			}
			const float gn = a.gain[cm];
			if (gn != 1.0f) {
#pragma unroll
				for (int i = 0; i < 8; ++i) own[i] = sat16((int)(gn * (float)own[i]));
			}
			summed = true;
		}
		uint4 pk;
		pk.x = (unsigned)(own[0] & 0xffff) | ((unsigned)own[1] << 16);
		pk.y = (unsigned)(own[2] & 0xffff) | ((unsigned)own[3] << 16);
		pk.z = (unsigned)(own[4] & 0xffff) | ((unsigned)own[5] << 16);
		pk.w = (unsigned)(own[6] & 0xffff) | ((unsigned)own[7] << 16);
		*reinterpret_cast<uint4 *>(tile + (size_t)m * ncol + 8 * gl) = pk; // |own| <= 32767 also after the gain
	}
	(void)summed;
	__syncthreads();
	// step 1: thread (range r, column j) adds its `per` members
	if (t < ma.q * ncol) {
		const int r = t / ncol, j = t - r * ncol;
		const int m0 = r * ma.per, m1 = min(a.mm, m0 + ma.per);
		int acc = 0;
		for (int k = m0; k < m1; ++k) acc += tile[k * ncol + j];
		part[r * ncol + j] = acc;
	}
	__syncthreads();
	if (t < ncol) {
		int acc = 0;
		for (int r = 0; r < ma.q; ++r) acc += part[r * ncol + t];
		total[t] = acc;
	}
	__syncthreads();
	if (conf_mode == 0) { // one mixed row per conference
		if (t < ma.gb && g0 + t < ma.ngrp) {
			const int4 lo = *reinterpret_cast<const int4 *>(total + 8 * t);
			const int4 hi = *reinterpret_cast<const int4 *>(total + 8 * t + 4);
			uint4 o;
			o.x = (unsigned)(sat16(lo.x) & 0xffff) | ((unsigned)sat16(lo.y) << 16);
			o.y = (unsigned)(sat16(lo.z) & 0xffff) | ((unsigned)sat16(lo.w) << 16);
			o.z = (unsigned)(sat16(hi.x) & 0xffff) | ((unsigned)sat16(hi.y) << 16);
			o.w = (unsigned)(sat16(hi.z) & 0xffff) | ((unsigned)sat16(hi.w) << 16);
			*reinterpret_cast<uint4 *>(a.out + (size_t)c * a.out_conf_stride + 8 * (g0 + t)) = o;
		}
		return;
	}
	if (mine && (f & MI_MIX_OUTPUT)) {
		const int4 lo = *reinterpret_cast<const int4 *>(total + 8 * gl);
		const int4 hi = *reinterpret_cast<const int4 *>(total + 8 * gl + 4);
		uint4 o;
		o.x = (unsigned)(sat16(lo.x - own[0]) & 0xffff) | ((unsigned)sat16(lo.y - own[1]) << 16);
		o.y = (unsigned)(sat16(lo.z - own[2]) & 0xffff) | ((unsigned)sat16(lo.w - own[3]) << 16);
		o.z = (unsigned)(sat16(hi.x - own[4]) & 0xffff) | ((unsigned)sat16(hi.y - own[5]) << 16);
		o.w = (unsigned)(sat16(hi.z - own[6]) & 0xffff) | ((unsigned)sat16(hi.w - own[7]) << 16);
		*reinterpret_cast<uint4 *>(a.out + (size_t)(c * a.mm + m) * a.ns + 8 * g) = o;
	}
}


// Any tick length (ticks of 44.1 kHz audio are 441 samples): one lane per sample, 2-byte accesses.  Same arithmetic as
// mixer_kernel; only reached when nsamples is not a multiple of 4.
template <int MODE>
__global__ __launch_bounds__(256) void mixer_scalar_kernel(MixArgs a) {
	const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
	if (g >= (long long)a.nconf * a.ns) return;
	const int c = (int)(g / a.ns), i = (int)(g - (long long)c * a.ns);
	if (a.run && !a.run[c]) return;
	const int conf_mode = a.conf_modes ? a.conf_modes[c] : a.conf_mode;
	const uint8_t *fl = a.flags + (size_t)c * a.mm;
	auto contrib = [&](int m, bool &summed) -> int {
		summed = false;
		const int cm = c * a.mm + m;
		const unsigned f = fl[m];
		if (!(f & MI_MIX_LINKED) || (a.has_data && !a.has_data[cm])) return 0;
		int v = a.in[(size_t)cm * a.ns + i];
		if (f & MI_MIX_ACTIVE) {
			const float gn = a.gain[cm];
			if (gn != 1.0f) v = sat16((int)(gn * (float)v));
			summed = true;
		}
		return v;
	};
	int sum = 0;
	if (MODE == 2) {
		sum = a.sum_in[(size_t)c * a.ns + i];
	} else {
		for (int m = 0; m < a.mm; ++m) {
			bool summed;
			const int v = contrib(m, summed);
			if (summed) sum += v;
		}
	}
	if (MODE == 1) {
		a.sum_out[(size_t)c * a.ns + i] = sum;
		return;
	}
	if (conf_mode == 0) {
		a.out[(size_t)c * a.out_conf_stride + i] = (int16_t)sat16(sum);
		return;
	}
	for (int m = 0; m < a.mm; ++m)
		if (fl[m] & MI_MIX_OUTPUT) {
			bool summed;
			const int own = contrib(m, summed);
			a.out[(size_t)(c * a.mm + m) * a.ns + i] = (int16_t)sat16(sum - (summed ? own : 0));
		}
}

} // namespace

struct mi_mixer {
	mi_ctx *ctx = nullptr;
	int nconf = 0, mm = 0, ns = 0;
	uint8_t *d_flags = nullptr;
	float *d_gain = nullptr;
};

// fused tick, 8 or more channels, 16-byte rows: the member-parallel kernel
static bool launch_members(mi_mixer *m, const MixArgs &a) {
	if (a.mm < 8 || (a.ns & 7) != 0 || (a.out_conf_stride & 7) != 0) return false;
	if (((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) != 0) return false;
	MixMArgs ma;
	ma.a = a;
	ma.ngrp = a.ns / 8;
	ma.gb = std::min(ma.ngrp, 256 / a.mm);
	ma.cblocks = mi::ceil_div(ma.ngrp, ma.gb);
	const int ncol = ma.gb * 8;
	ma.q = std::max(1, std::min(256 / ncol, a.mm));
	ma.per = mi::ceil_div(a.mm, ma.q);
	const size_t lds = (size_t)a.mm * ncol * 2 + (size_t)(ma.q + 1) * ncol * 4;
	hipLaunchKernelGGL(mixer_members_kernel, dim3(mi::xcd_grid((unsigned)(a.nconf * ma.cblocks))), dim3(256), lds, m->ctx->stream, ma);
	return true;
}

template <int MODE>
static int launch_mixer(mi_mixer *m, MixArgs &a) {
	if (a.ns % 4) {
		const long long lanes = (long long)a.nconf * a.ns;
		hipLaunchKernelGGL(mixer_scalar_kernel<MODE>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, m->ctx->stream, a);
		MI_LAUNCH_CHECK();
		return MI_OK;
	}
	if (MODE == 0 && launch_members(m, a)) {
		MI_LAUNCH_CHECK();
		return MI_OK;
	}
	const long long work = (long long)a.nconf * a.quads;
	const int grid = (int)((work + 255) / 256);
	hipStream_t st = m->ctx->stream;
	if (MODE == 2 || a.mm <= 8) hipLaunchKernelGGL((mixer_kernel<8, MODE>), dim3(grid), dim3(256), 0, st, a);
	else if (a.mm <= 16) hipLaunchKernelGGL((mixer_kernel<16, MODE>), dim3(grid), dim3(256), 0, st, a);
	else if (a.mm <= 32) hipLaunchKernelGGL((mixer_kernel<32, MODE>), dim3(grid), dim3(256), 0, st, a);
	else hipLaunchKernelGGL((mixer_kernel<MI_MIXER_MAX_CHANNELS, MODE>), dim3(grid), dim3(256), 0, st, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

static void fill_args(mi_mixer *m, MixArgs &a, const int16_t *d_in, const uint8_t *d_has, int conf_mode) {
	a.in = d_in;
	a.has_data = d_has;
	a.flags = m->d_flags;
	a.gain = m->d_gain;
	a.out = nullptr;
	a.sum_out = nullptr;
	a.sum_in = nullptr;
	a.run = nullptr;
	a.conf_modes = nullptr;
	a.out_conf_stride = m->ns;
	a.nconf = m->nconf;
	a.mm = m->mm;
	a.ns = m->ns;
	a.quads = m->ns / 4;
	a.conf_mode = conf_mode;
}

void mi_mixer_view(const mi_mixer *m, MixerView *v) {
	*v = MixerView();
	if (!m) return;
	v->flags = m->d_flags;
	v->gain = m->d_gain;
	v->nconf = m->nconf, v->mm = m->mm, v->ns = m->ns;
	v->device = m->ctx->device;
}

extern "C" {

int mi_mixer_create(mi_ctx *ctx, int nconf, int max_members, int nsamples, mi_mixer **out) {
	MI_CHECK_ARG(ctx && out && nconf > 0 && max_members > 0 && max_members <= MI_MIXER_MAX_CHANNELS);
	MI_CHECK_ARG(nsamples > 0);
	*out = nullptr;
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_mixer *m = new mi_mixer();
	m->ctx = ctx;
	m->nconf = nconf;
	m->mm = max_members;
	m->ns = nsamples;
	const size_t n = (size_t)nconf * max_members;
	if (hipMalloc((void **)&m->d_flags, n) != hipSuccess ||
	    hipMalloc((void **)&m->d_gain, n * sizeof(float)) != hipSuccess) {
		mi::set_error("hipMalloc failed for mixer controls");
		mi_mixer_destroy(m);
		return MI_ENOMEM;
	}
	std::vector<uint8_t> f(n, (uint8_t)(MI_MIX_LINKED | MI_MIX_ACTIVE | MI_MIX_OUTPUT));
	std::vector<float> g(n, 1.0f);
	int rc = mi_mixer_set_controls(m, f.data(), g.data());
	if (rc != MI_OK) {
		mi_mixer_destroy(m);
		return rc;
	}
	*out = m;
	return MI_OK;
}

void mi_mixer_destroy(mi_mixer *m) {
	if (!m) return;
	(void)hipSetDevice(m->ctx->device);
	if (m->d_flags) (void)hipFree(m->d_flags);
	if (m->d_gain) (void)hipFree(m->d_gain);
	delete m;
}

int mi_mixer_set_controls(mi_mixer *m, const uint8_t *h_flags, const float *h_gain) {
	MI_CHECK_ARG(m != nullptr);
	if (m->ctx->activate() != MI_OK) return MI_ENODEV;
	const size_t n = (size_t)m->nconf * m->mm;
	// synchronous copies: the host arrays may be transient
	MI_HIP(hipStreamSynchronize(m->ctx->stream));
	if (h_flags) MI_HIP(hipMemcpy(m->d_flags, h_flags, n, hipMemcpyHostToDevice));
	if (h_gain) MI_HIP(hipMemcpy(m->d_gain, h_gain, n * sizeof(float), hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_mixer_process(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int conf_mode, int16_t *d_out) {
	MI_CHECK_ARG(m && d_in && d_out);
	if (m->ctx->activate() != MI_OK) return MI_ENODEV;
	MixArgs a;
	fill_args(m, a, d_in, d_has_data, conf_mode);
	a.out = d_out;
	return launch_mixer<0>(m, a);
}

int mi_mixer_process_masked(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int conf_mode,
                            const uint8_t *d_conf_mode, int16_t *d_out, const uint8_t *d_run) {
	MI_CHECK_ARG(m && d_in && d_out);
	if (m->ctx->activate() != MI_OK) return MI_ENODEV;
	MixArgs a;
	fill_args(m, a, d_in, d_has_data, conf_mode);
	a.out = d_out;
	a.run = d_run;
	a.conf_modes = d_conf_mode;
	a.out_conf_stride = m->mm * m->ns; // one full [members][nsamples] slab per conference
	return launch_mixer<0>(m, a);
}

int mi_mixer_partial_sum(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int32_t *d_sum) {
	MI_CHECK_ARG(m && d_in && d_sum);
	if (m->ctx->activate() != MI_OK) return MI_ENODEV;
	MixArgs a;
	fill_args(m, a, d_in, d_has_data, 1);
	a.sum_out = d_sum;
	return launch_mixer<1>(m, a);
}

int mi_mixer_finalize(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, const int32_t *d_sum,
                      int conf_mode, int16_t *d_out) {
	MI_CHECK_ARG(m && d_in && d_sum && d_out);
	if (m->ctx->activate() != MI_OK) return MI_ENODEV;
	MixArgs a;
	fill_args(m, a, d_in, d_has_data, conf_mode);
	a.sum_in = d_sum;
	a.out = d_out;
	return launch_mixer<2>(m, a);
}

int mi_mixer_process_host(mi_mixer *m, const int16_t *h_in, const uint8_t *h_has_data, int conf_mode,
                          int16_t *h_out) {
	MI_CHECK_ARG(m && h_in && h_out);
	mi_ctx *c = m->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t n = (size_t)m->nconf * m->mm;
	const size_t ib = n * m->ns * sizeof(int16_t);
	const size_t ob = conf_mode ? ib : (size_t)m->nconf * m->ns * sizeof(int16_t);
	void *din, *dout, *dhas = nullptr;
	int rc;
	if ((rc = c->ensure_scratch(0, ib, &din)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, ob, &dout)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(din, h_in, ib, hipMemcpyHostToDevice, c->stream));
	if (h_has_data) {
		if ((rc = c->ensure_scratch(2, n, &dhas)) != MI_OK) return rc;
		MI_HIP(hipMemcpyAsync(dhas, h_has_data, n, hipMemcpyHostToDevice, c->stream));
	}
	// rows the kernel leaves untouched (disabled outputs) keep the caller's bytes
	MI_HIP(hipMemcpyAsync(dout, h_out, ob, hipMemcpyHostToDevice, c->stream));
	rc = mi_mixer_process(m, (const int16_t *)din, (const uint8_t *)dhas, conf_mode, (int16_t *)dout);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_out, dout, ob, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_mixer(reinterpret_cast<const void *>(&mixer_scalar_kernel<0>));
