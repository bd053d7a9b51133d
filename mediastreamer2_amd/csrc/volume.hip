// volume.hip -- batched MSVolume chunk (meter, echo limiter, AGC, noise gate,
// Q12 gain) for gfx950.  Built with -ffp-contract=off: the control chain of
// src/audiofilters/msvolume.c is float32 evaluated unfused in source order on
// the reference's x86-64 build, and the integer gain (hence every output
// sample) depends on it bit for bit.
//
// Per stream and chunk this is volume_process's loop body (msvolume.c:480-513):
//   update_energy :388-407  sequential float32 sum of squares + integer peak
//   volume_echo_avoider_process :201-238 (peer energy), volume_agc_process :172-184,
//   volume_noise_gate_process :240-260, apply_gain :409-445 (ramp, Q12 integer
//   gain with C truncating division, symmetric +-32767 clamp, optional DC removal).
//
// Mapping: a 128-thread workgroup owns SPB=16 streams.  (A) all lanes stage the
// 16 chunks into LDS with 16-byte coalesced loads (row pitch odd in dwords, so
// the per-stream walks of phase B are bank-conflict free); (B) one lane per
// stream walks its chunk serially -- the float accumulation ORDER is part of
// the reference's result -- and runs the scalar control chain;
// (C) all lanes apply the integer gain and store 16 bytes per lane.  Streams
// whose gain is exactly 1 (and no DC removal) are not written back, as in the
// reference (msvolume.c:440).  HBM traffic: 2 B/sample read, <= 2 B/sample
// written, + ~100 B of per-stream state.
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int SPB = 16;      // streams per block
constexpr int VTHREADS = 128;

struct VolArgs {
	int16_t *samples;
	const int32_t *nsamples_per_stream; // or null
	const mi_volume_params *params;
	mi_volume_state *state;
	const float *energy_prev; // peers read last launch's energy
	float *energy_next;
	int nstreams, nsamples, stride, sample_rate, pitch_dw;
};

__device__ __forceinline__ int sat16(int v) { return (v > 32767) ? 32767 : ((v < -32767) ? -32767 : v); }

__global__ __launch_bounds__(VTHREADS) void volume_kernel(VolArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint32_t *rows = reinterpret_cast<uint32_t *>(smem); // [SPB][pitch_dw]
	__shared__ int s_intgain[SPB], s_dcoff[SPB], s_mode[SPB], s_n[SPB];

	const int tid = threadIdx.x;
	const int s0 = blockIdx.x * SPB;
	const int nloc = min(SPB, a.nstreams - s0);
	const float max_e = (32768 * 0.7f);

	if (tid < SPB) {
		int n = 0;
		if (tid < nloc) n = a.nsamples_per_stream ? a.nsamples_per_stream[s0 + tid] : a.nsamples;
		s_n[tid] = min(max(n, 0), a.nsamples);
	}
	__syncthreads();

	// ---- (A) stage chunks
	const bool vec = ((a.stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0);
	if (vec) {
		const int oct = (a.nsamples + 7) >> 3; // 16-byte groups per row
		for (int i = tid; i < nloc * oct; i += VTHREADS) {
			const int sl = i / oct, q = i - sl * oct;
			if (8 * q < s_n[sl]) {
				const uint4 v = *reinterpret_cast<const uint4 *>(a.samples + (size_t)(s0 + sl) * a.stride + 8 * q);
				uint32_t *r = rows + sl * a.pitch_dw + 4 * q;
				r[0] = v.x;
				r[1] = v.y;
				r[2] = v.z;
				r[3] = v.w;
			}
		}
	} else {
		int16_t *rows16 = reinterpret_cast<int16_t *>(rows);
		for (int i = tid; i < nloc * a.nsamples; i += VTHREADS) {
			const int sl = i / a.nsamples, q = i - sl * a.nsamples;
			if (q < s_n[sl]) rows16[sl * a.pitch_dw * 2 + q] = a.samples[(size_t)(s0 + sl) * a.stride + q];
		}
	}
	__syncthreads();

	// ---- (B) one lane per stream: meter + control chain
	if (tid < nloc && s_n[tid] > 0) {
		const int s = s0 + tid;
		const int n = s_n[tid];
		const mi_volume_params p = a.params[s];
		mi_volume_state st = a.state[s];
		const int16_t *x = reinterpret_cast<const int16_t *>(rows + tid * a.pitch_dw);

		float acc = 0;
		int pk = 0, dcsum = 0;
		for (int i = 0; i < n; ++i) {
			const int v = x[i];
			acc += (float)(v * v);
			const int av = v < 0 ? -v : v;
			if (av > pk) pk = av;
			dcsum += v;
		}
		const float en = (float)((sqrt((double)(acc / n)) + 1) / (double)max_e);
		st.energy = (en * 0.2f) + st.energy * (1.0f - 0.2f);
		st.level_pk = (float)pk / max_e;
		st.instant_energy = en;

		float target = p.static_gain;
		if (p.peer >= 0) { // echo limiter
			const float peer_e = a.energy_prev[p.peer], peer_pk = peer_e;
			if (peer_pk > st.lt_speaker_en) st.lt_speaker_en = peer_pk;
			else st.lt_speaker_en = (0.005f * peer_pk) + (0.995f * st.lt_speaker_en);
			const float ratio = (st.energy / (st.lt_speaker_en + p.ea_thres));
			if (peer_e > p.ea_thres) {
				if (ratio > p.ea_transmit_thres) {
					st.target_gain = p.static_gain;
					st.fast_upramp = 1;
				} else {
					st.target_gain = p.static_gain / (1 + (peer_e * p.force));
					st.sustain_dur = p.sustain_time;
				}
			} else {
				if (st.sustain_dur > 0) {
					st.sustain_dur -= (n * 1000) / a.sample_rate;
				} else {
					st.target_gain = p.static_gain;
					st.fast_upramp = 1;
				}
			}
			target = st.target_gain;
		}
		if (p.agc_enabled) target /= (0.5f + st.level_pk) / 1;
		if (p.noise_gate_enabled) {
			float tgain = p.ng_floorgain;
			if (st.instant_energy > p.ng_threshold) {
				st.ng_noise_dur = p.ng_cut_time;
				tgain = 1.0f;
			} else if (st.ng_noise_dur > 0) {
				st.ng_noise_dur -= (n * 1000) / a.sample_rate;
				tgain = 1.0f;
			}
			st.ng_gain = st.ng_gain * 0.75f + tgain * 0.25f;
		}
		// apply_gain: multiplicative ramp toward target
		if (st.gain < target) {
			if (st.gain < p.ng_floorgain) st.gain = p.ng_floorgain;
			st.gain *= 1 + (st.fast_upramp ? p.vol_fast_upramp : p.vol_upramp);
			if (st.gain > target) st.gain = target;
		} else if (st.gain > target) {
			st.gain *= 1 - p.vol_downramp;
			if (st.gain < target) st.gain = target;
			st.fast_upramp = 0;
		}
		const float gain = st.gain * st.ng_gain;
		s_intgain[tid] = (int32_t)(gain * 4096);
		s_dcoff[tid] = st.dc_offset;
		if (p.remove_dc) {
			s_mode[tid] = 2;
			st.dc_offset = (st.dc_offset * 7 + dcsum * 2 / (2 * n)) / 8;
		} else {
			s_mode[tid] = (gain != 1) ? 1 : 0;
		}
		a.state[s] = st;
		a.energy_next[s] = st.energy;
	} else if (tid < SPB) {
		s_mode[tid] = 0;
		if (tid < nloc) a.energy_next[s0 + tid] = a.state[s0 + tid].energy;
	}
	__syncthreads();

	// ---- (C) integer gain, coalesced write-back
	if (vec) {
		const int oct = (a.nsamples + 7) >> 3;
		for (int i = tid; i < nloc * oct; i += VTHREADS) {
			const int sl = i / oct, q = i - sl * oct;
			const int mode = s_mode[sl], n = s_n[sl];
			if (mode == 0 || 8 * q >= n) continue;
			const int ig = s_intgain[sl], dc = (mode == 2) ? s_dcoff[sl] : 0;
			const int16_t *r = reinterpret_cast<const int16_t *>(rows + sl * a.pitch_dw) + 8 * q;
			int16_t *dst = a.samples + (size_t)(s0 + sl) * a.stride + 8 * q;
			if (8 * q + 8 <= n) {
				union {
					uint4 v;
					int16_t h[8];
				} o;
#pragma unroll
				for (int k = 0; k < 8; ++k) o.h[k] = (int16_t)sat16(((r[k] - dc) * ig) / 4096);
				*reinterpret_cast<uint4 *>(dst) = o.v;
			} else {
				for (int k = 0; 8 * q + k < n; ++k) dst[k] = (int16_t)sat16(((r[k] - dc) * ig) / 4096);
			}
		}
	} else {
		const int16_t *rows16 = reinterpret_cast<const int16_t *>(rows);
		for (int i = tid; i < nloc * a.nsamples; i += VTHREADS) {
			const int sl = i / a.nsamples, q = i - sl * a.nsamples;
			const int mode = s_mode[sl];
			if (mode == 0 || q >= s_n[sl]) continue;
			const int dc = (mode == 2) ? s_dcoff[sl] : 0;
			a.samples[(size_t)(s0 + sl) * a.stride + q] =
			    (int16_t)sat16(((rows16[sl * a.pitch_dw * 2 + q] - dc) * s_intgain[sl]) / 4096);
		}
	}
}

} // namespace

struct mi_volume {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, sample_rate = 0;
	mi_volume_params *d_params = nullptr;
	mi_volume_state *d_state = nullptr;
	float *d_energy[2] = {nullptr, nullptr};
	int cur = 0;
};

extern "C" {

void mi_volume_default_params(mi_volume_params *p) {
	if (!p) return;
	memset(p, 0, sizeof(*p));
	p->static_gain = 1;
	p->vol_upramp = 0.4f;
	p->vol_fast_upramp = 0.4f * 3;
	p->vol_downramp = 0.4f;
	p->ea_thres = 0.1f;
	p->ea_transmit_thres = 4;
	p->force = 4.0f;
	p->sustain_time = 200;
	p->ng_cut_time = 400;
	p->ng_threshold = 0.1f;
	p->ng_floorgain = 0.005f;
	p->peer = -1;
}

int mi_volume_create(mi_ctx *ctx, int nstreams, int sample_rate, mi_volume **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && sample_rate > 0);
	*out = nullptr;
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_volume *v = new mi_volume();
	v->ctx = ctx;
	v->nstreams = nstreams;
	v->sample_rate = sample_rate;
	if (hipMalloc((void **)&v->d_params, sizeof(mi_volume_params) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_state, sizeof(mi_volume_state) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_energy[0], sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_energy[1], sizeof(float) * (size_t)nstreams) != hipSuccess) {
		mi::set_error("hipMalloc failed for volume state");
		mi_volume_destroy(v);
		return MI_ENOMEM;
	}
	mi_volume_params dp;
	mi_volume_default_params(&dp);
	std::vector<mi_volume_params> hp((size_t)nstreams, dp);
	mi_volume_state ds;
	memset(&ds, 0, sizeof(ds));
	ds.gain = ds.target_gain = 1; // volume_init msvolume.c:92
	ds.ng_gain = 1;               // :112
	std::vector<mi_volume_state> hs((size_t)nstreams, ds);
	if (hipMemcpy(v->d_params, hp.data(), sizeof(dp) * hp.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(v->d_state, hs.data(), sizeof(ds) * hs.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemset(v->d_energy[0], 0, sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMemset(v->d_energy[1], 0, sizeof(float) * (size_t)nstreams) != hipSuccess) {
		mi::set_error("volume state upload failed");
		mi_volume_destroy(v);
		return MI_ENODEV;
	}
	*out = v;
	return MI_OK;
}

void mi_volume_destroy(mi_volume *v) {
	if (!v) return;
	(void)hipSetDevice(v->ctx->device);
	if (v->d_params) (void)hipFree(v->d_params);
	if (v->d_state) (void)hipFree(v->d_state);
	if (v->d_energy[0]) (void)hipFree(v->d_energy[0]);
	if (v->d_energy[1]) (void)hipFree(v->d_energy[1]);
	delete v;
}

int mi_volume_set_params(mi_volume *v, int first, int count, const mi_volume_params *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	for (int i = 0; i < count; ++i) MI_CHECK_ARG(h[i].peer < v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(v->d_params + first, h, sizeof(*h) * (size_t)count, hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_volume_get_state(mi_volume *v, int first, int count, mi_volume_state *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(h, v->d_state + first, sizeof(*h) * (size_t)count, hipMemcpyDeviceToHost));
	return MI_OK;
}

int mi_volume_set_state(mi_volume *v, int first, int count, const mi_volume_state *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(v->d_state + first, h, sizeof(*h) * (size_t)count, hipMemcpyHostToDevice));
	std::vector<float> e((size_t)count);
	for (int i = 0; i < count; ++i) e[(size_t)i] = h[i].energy;
	MI_HIP(hipMemcpy(v->d_energy[v->cur] + first, e.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_volume_process(mi_volume *v, int16_t *d_samples, int nsamples, int stride, const int32_t *d_nsamples) {
	MI_CHECK_ARG(v && d_samples && nsamples > 0 && stride >= nsamples);
	if (nsamples > 3840) {
		mi::set_error("chunk of %d samples exceeds the volume kernel's LDS staging (max 3840)", nsamples);
		return MI_ENOTSUP;
	}
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	VolArgs a;
	a.samples = d_samples;
	a.nsamples_per_stream = d_nsamples;
	a.params = v->d_params;
	a.state = v->d_state;
	a.energy_prev = v->d_energy[v->cur];
	a.energy_next = v->d_energy[v->cur ^ 1];
	a.nstreams = v->nstreams;
	a.nsamples = nsamples;
	a.stride = stride;
	a.sample_rate = v->sample_rate;
	// row pitch: whole 16-byte groups, then made odd in dwords (bank-conflict-free walks)
	int pitch = ((nsamples + 7) >> 3) * 4;
	pitch |= 1;
	a.pitch_dw = pitch;
	const size_t lds = (size_t)SPB * pitch * sizeof(uint32_t);
	hipLaunchKernelGGL(volume_kernel, dim3(mi::ceil_div(v->nstreams, SPB)), dim3(VTHREADS), lds, v->ctx->stream, a);
	MI_LAUNCH_CHECK();
	v->cur ^= 1;
	return MI_OK;
}

int mi_volume_process_host(mi_volume *v, int16_t *h_samples, int nsamples, int stride, const int32_t *h_nsamples) {
	MI_CHECK_ARG(v && h_samples);
	mi_ctx *c = v->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t b = (size_t)v->nstreams * stride * sizeof(int16_t);
	void *d, *dn = nullptr;
	int rc;
	if ((rc = c->ensure_scratch(0, b, &d)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(d, h_samples, b, hipMemcpyHostToDevice, c->stream));
	if (h_nsamples) {
		if ((rc = c->ensure_scratch(2, sizeof(int32_t) * (size_t)v->nstreams, &dn)) != MI_OK) return rc;
		MI_HIP(hipMemcpyAsync(dn, h_nsamples, sizeof(int32_t) * (size_t)v->nstreams, hipMemcpyHostToDevice, c->stream));
	}
	rc = mi_volume_process(v, (int16_t *)d, nsamples, stride, (const int32_t *)dn);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_samples, d, b, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"
