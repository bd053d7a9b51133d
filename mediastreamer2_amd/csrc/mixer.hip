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

} // namespace

struct mi_mixer {
	mi_ctx *ctx = nullptr;
	int nconf = 0, mm = 0, ns = 0;
	uint8_t *d_flags = nullptr;
	float *d_gain = nullptr;
};

template <int MODE>
static int launch_mixer(mi_mixer *m, MixArgs &a) {
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

extern "C" {

int mi_mixer_create(mi_ctx *ctx, int nconf, int max_members, int nsamples, mi_mixer **out) {
	MI_CHECK_ARG(ctx && out && nconf > 0 && max_members > 0 && max_members <= MI_MIXER_MAX_CHANNELS);
	MI_CHECK_ARG(nsamples > 0);
	*out = nullptr;
	if (nsamples % 4) {
		mi::set_error("nsamples per tick must be a multiple of 4 (got %d); every rate*10ms the filter uses is",
		              nsamples);
		return MI_ENOTSUP;
	}
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
