// equalizer.hip -- batched MSEqualizer for gfx950.  Built with -ffp-contract=off.
//
// Run path (per tick, on the GPU): equalizer_state_run -> ms_fir_mem16
// (src/audiofilters/equalizer.c:263-269, src/utils/dsptools.c:253-268): an
// nfft-tap (128/256/512) direct-form FIR per stream,
//     y[n] = ((x[n-ord+1]*h[ord-1] + h[ord-2]*x[n-ord+2]) + ...) + h[0]*x[n]
// evaluated in exactly that order with separate float32 multiply and add, so
// the int16 output is bit-identical to the reference build.  The delay line
// is not shifted: the last ord-1 inputs live in HBM as int16 (exact), are
// staged with the new block into LDS as float, and every lane slides a
// register window over it (1 LDS read feeds R outputs); the stream's taps are
// wave-uniform and come through the scalar cache.
//
// Design path (on gain changes only, on the host): MS_EQUALIZER_SET_GAIN
// (equalizer.c:128-172), then impulse response = packed inverse FFT, half
// swap, Hamming window (equalizer.c:184-237), uploaded as the stream's taps.
#include "common.hpp"
#include "host_fft.hpp"

#pragma clang fp contract(off)

namespace {

struct EqArgs {
	int16_t *samples;
	int16_t *hist;      // [nstreams][ord] int16 (ord-1 used)
	const float *taps;  // [nstreams][ord]
	const uint8_t *active;
	const int32_t *nper; // per-stream block length or null
	int nstreams, nsamples, stride, ord;
};

// One wavefront per stream, 8 outputs per lane held as four register pairs.  v_pk_mul_f32 + v_pk_add_f32 round exactly like the scalar
// multiply and add (the product is NOT fused into the sum, as in the reference's x86 build), but issue two outputs per slot; the tap is
// wave-uniform, window pairs at even offsets are register pairs as they come from 16-byte LDS reads, pairs at odd offsets cost one
// v_pk_mov_b32 each (one per two taps).  Per-output accumulation order is the reference's (oldest sample first).  (The one-output-per-slot
// form this replaced in round 3 left the source in round 6; scripts/micro/ keeps the idiom's own check.)
typedef float eq_f2 __attribute__((ext_vector_type(2)));
constexpr int EQ_PR = 8; // outputs per lane
template <int ORD>
__global__ __launch_bounds__(64) void equalizer_pk_kernel(EqArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *buf = reinterpret_cast<float *>(smem); // [ORD-1 + nsamples + 16]
	const int s = blockIdx.x;
	if (!a.active[s]) return;
	const int nsamples = a.nper ? min(max(a.nper[s], 0), a.nsamples) : a.nsamples;
	if (nsamples == 0) return;
	const int lane = threadIdx.x;
	int16_t *xs = a.samples + (size_t)s * a.stride;
	int16_t *hs = a.hist + (size_t)s * ORD;
	const float *__restrict__ h = a.taps + (size_t)s * ORD;

	for (int i = lane; i < ORD - 1; i += 64) buf[i] = (float)hs[i];
	for (int i = lane; i < nsamples; i += 64) buf[ORD - 1 + i] = (float)xs[i];
	for (int i = lane; i < 16; i += 64) buf[ORD - 1 + nsamples + i] = 0.f;
	__syncthreads();

	for (int n0 = lane * EQ_PR; n0 < nsamples; n0 += 64 * EQ_PR) {
		const float4 *wp = reinterpret_cast<const float4 *>(buf + n0);
		float4 c0 = wp[0], c1 = wp[1], c2 = wp[2], c3 = wp[3];
		eq_f2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 2
		for (int c = 0; c < ORD / 8; ++c) {
			const eq_f2 ev[8] = {{c0.x, c0.y}, {c0.z, c0.w}, {c1.x, c1.y}, {c1.z, c1.w},
			                     {c2.x, c2.y}, {c2.z, c2.w}, {c3.x, c3.y}, {c3.z, c3.w}};
			eq_f2 od[7];
#pragma unroll
			for (int i = 0; i < 7; ++i) od[i] = __builtin_shufflevector(ev[i], ev[i + 1], 1, 2);
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const float t = h[ORD - 1 - (8 * c + u)]; // wave-uniform: scalar load
				const eq_f2 ts = {t, t};
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int k = u + 2 * q;
					const eq_f2 wk = (k & 1) ? od[k / 2] : ev[k / 2];
					const eq_f2 p = ts * wk;
					acc[q] = acc[q] + p;
				}
			}
			c0 = c2, c1 = c3;
			c2 = wp[2 * c + 4], c3 = wp[2 * c + 5]; // the last trip reads the zero slack
		}
#pragma unroll
		for (int r = 0; r < EQ_PR; ++r) {
			if (n0 + r < nsamples) {
				const float v = (r & 1) ? acc[r / 2].y : acc[r / 2].x;
				// (int16_t)float of the reference is UB out of range (equalizer.c:251-255); saturate
				const int q = v >= 32767.f ? 32767 : (v <= -32768.f ? -32768 : (int)v);
				xs[n0 + r] = (int16_t)q;
			}
		}
	}
	for (int i = lane; i < ORD - 1; i += 64) hs[i] = (int16_t)buf[nsamples + i];
}

struct HostEq { // EqualizerState equalizer.c:37-46 (design-side fields)
	std::vector<float> spectrum; // fft_cpx, packed real
	bool stale = true;
	bool active = true;
};

} // namespace

struct mi_equalizer {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, rate = 0, nfft = 0;
	std::vector<HostEq> st;
	bool any_stale = true, active_dirty = true;
	float *d_taps = nullptr;
	int16_t *d_hist = nullptr;
	uint8_t *d_active = nullptr;
	std::mutex mu;
};

namespace {

void flatten(mi_equalizer *e, HostEq &s) { // equalizer.c:49-55
	s.spectrum.assign((size_t)e->nfft, 0.f);
	const float val = (float)(1.0f / e->nfft);
	s.spectrum[0] = val;
	for (int i = 1; i < e->nfft; i += 2) s.spectrum[(size_t)i] = val;
	s.stale = true;
}

int hz_to_index(const mi_equalizer *e, int hz) { // equalizer.c:95-108
	if (hz < 0) return -1;
	if (hz > (e->rate / 2)) hz = (e->rate / 2);
	int ret = ((hz * e->nfft) + (e->rate / 2)) / e->rate;
	if (ret == e->nfft / 2) ret = (e->nfft / 2) - 1;
	return ret;
}

int index2hz(const mi_equalizer *e, int index) { return (index * e->rate + e->nfft / 2) / e->nfft; }

float gainpoint(int f, int freq_0, float sqrt_gain, int freq_bw) { // equalizer.c:128-135
	float k1 = ((float)(f * f) - (float)(freq_0 * freq_0));
	k1 *= k1;
	float k2 = (float)(f * freq_bw);
	k2 *= k2;
	return (k1 + k2 * sqrt_gain) / (k1 + k2 / sqrt_gain);
}

void point_set(mi_equalizer *e, HostEq &s, int i, float gain) { // equalizer.c:137-145
	const int index = 1 + ((i - 1) * 2);
	if (index >= 0 && index < e->nfft)
		s.spectrum[(size_t)index] = (s.spectrum[(size_t)index] * (int)(gain * 32768)) / 32768;
}

void design(const mi_equalizer *e, const HostEq &s, float *fir) { // equalizer.c:215-237
	const int n = e->nfft, half = n / 2;
	mi::packed_real_ifft(n, s.spectrum.data(), fir);
	for (int i = 0; i < half; ++i) std::swap(fir[i], fir[i + half]); // time_shift :184-193
	for (int i = 0; i < n; ++i) {                                    // norm_and_apodize :203-213
		const float x = (float)((float)i * 2 * M_PI / (float)n);
		const float w = (float)(0.54 - (0.46 * cos(x)));
		fir[i] = w * (float)fir[i];
	}
}

int upload_stale(mi_equalizer *e) {
	if (e->any_stale) {
		// every stale stream's taps are designed into one staging array and go up run by run of neighbours, ONE wait at the end (a copy and a wait
		// per stream made the first tick of 2 048 legs with a mic_equalizer 110 ms)
		std::vector<float> fir;
		int run0 = -1;
		auto send = [&](int end) -> int { // streams [run0, end)
			if (run0 < 0) return MI_OK;
			MI_HIP(hipMemcpyAsync(e->d_taps + (size_t)run0 * e->nfft, fir.data() + (size_t)run0 * e->nfft, sizeof(float) * (size_t)(end - run0) * (size_t)e->nfft,
			                      hipMemcpyHostToDevice, e->ctx->stream));
			run0 = -1;
			return MI_OK;
		};
		for (int s = 0; s < e->nstreams; ++s) {
			HostEq &h = e->st[(size_t)s];
			if (!h.stale) {
				if (send(s) != MI_OK) return MI_ENODEV;
				continue;
			}
			if (fir.empty()) fir.resize((size_t)e->nstreams * (size_t)e->nfft);
			design(e, h, fir.data() + (size_t)s * e->nfft);
			if (run0 < 0) run0 = s;
			h.stale = false;
		}
		if (send(e->nstreams) != MI_OK) return MI_ENODEV;
		if (!fir.empty()) MI_HIP(hipStreamSynchronize(e->ctx->stream)); // (the staging array goes out of scope)
		e->any_stale = false;
	}
	if (e->active_dirty) {
		std::vector<uint8_t> act((size_t)e->nstreams);
		for (int s = 0; s < e->nstreams; ++s) act[(size_t)s] = e->st[(size_t)s].active ? 1 : 0;
		MI_HIP(hipMemcpyAsync(e->d_active, act.data(), act.size(), hipMemcpyHostToDevice, e->ctx->stream));
		MI_HIP(hipStreamSynchronize(e->ctx->stream));
		e->active_dirty = false;
	}
	return MI_OK;
}

} // namespace

extern "C" {

int mi_equalizer_create(mi_ctx *ctx, int nstreams, int sample_rate, mi_equalizer **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && sample_rate > 0);
	*out = nullptr;
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_equalizer *e = new mi_equalizer();
	e->ctx = ctx;
	e->nstreams = nstreams;
	e->rate = sample_rate;
	e->nfft = sample_rate < 16000 ? 128 : (sample_rate < 32000 ? 256 : 512); // equalizer.c:60-66
	e->st.resize((size_t)nstreams);
	// one flat design shared by every stream at start
	flatten(e, e->st[0]);
	std::vector<float> fir((size_t)e->nfft);
	design(e, e->st[0], fir.data());
	std::vector<float> all((size_t)nstreams * e->nfft);
	for (int s = 0; s < nstreams; ++s) {
		if (s) e->st[(size_t)s].spectrum = e->st[0].spectrum;
		e->st[(size_t)s].stale = false;
		memcpy(all.data() + (size_t)s * e->nfft, fir.data(), sizeof(float) * (size_t)e->nfft);
	}
	e->any_stale = false;
	const size_t hb = (size_t)nstreams * e->nfft * sizeof(int16_t);
	if (hipMalloc((void **)&e->d_taps, all.size() * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&e->d_hist, hb) != hipSuccess ||
	    hipMalloc((void **)&e->d_active, (size_t)nstreams) != hipSuccess) {
		mi::set_error("hipMalloc failed for equalizer state");
		mi_equalizer_destroy(e);
		return MI_ENOMEM;
	}
	if (hipMemcpy(e->d_taps, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemset(e->d_hist, 0, hb) != hipSuccess) {
		mi::set_error("equalizer state upload failed");
		mi_equalizer_destroy(e);
		return MI_ENODEV;
	}
	*out = e;
	return MI_OK;
}

void mi_equalizer_destroy(mi_equalizer *e) {
	if (!e) return;
	(void)hipSetDevice(e->ctx->device);
	if (e->d_taps) (void)hipFree(e->d_taps);
	if (e->d_hist) (void)hipFree(e->d_hist);
	if (e->d_active) (void)hipFree(e->d_active);
	delete e;
}

int mi_equalizer_fir_len(const mi_equalizer *e) { return e ? e->nfft : MI_EINVAL; }

int mi_equalizer_set_gain(mi_equalizer *e, int stream, float freq_hz, float gain, float width_hz) {
	MI_CHECK_ARG(e && stream >= 0 && stream < e->nstreams);
	std::lock_guard<std::mutex> lk(e->mu);
	HostEq &s = e->st[(size_t)stream];
	// equalizer_state_set equalizer.c:147-172
	const int freq_0 = (int)freq_hz;
	int freq_bw = (int)width_hz;
	const int delta_f = index2hz(e, 1);
	const float sqrt_gain = (float)sqrt(gain);
	const int mid = hz_to_index(e, freq_0);
	freq_bw -= delta_f / 2;
	if (freq_bw < delta_f / 2) freq_bw = delta_f / 2;
	int i = mid, f;
	point_set(e, s, i, gain);
	do {
		i++;
		f = index2hz(e, i);
		gain = gainpoint(f - delta_f, freq_0, sqrt_gain, freq_bw);
		point_set(e, s, i, gain);
	} while (i < e->nfft / 2 && (gain > 1.1 || gain < 0.9));
	i = mid;
	do {
		i--;
		f = index2hz(e, i);
		gain = gainpoint(f + delta_f, freq_0, sqrt_gain, freq_bw);
		point_set(e, s, i, gain);
	} while (i >= 0 && (gain > 1.1 || gain < 0.9));
	s.stale = true;
	e->any_stale = true;
	return MI_OK;
}

int mi_equalizer_flatten(mi_equalizer *e, int stream) {
	MI_CHECK_ARG(e && stream >= 0 && stream < e->nstreams);
	std::lock_guard<std::mutex> lk(e->mu);
	flatten(e, e->st[(size_t)stream]);
	e->any_stale = true;
	return MI_OK;
}

int mi_equalizer_prepare(mi_equalizer *e) {
	MI_CHECK_ARG(e);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	std::lock_guard<std::mutex> lk(e->mu);
	return upload_stale(e);
}

int mi_equalizer_set_active(mi_equalizer *e, int stream, int active) {
	MI_CHECK_ARG(e && stream >= 0 && stream < e->nstreams);
	std::lock_guard<std::mutex> lk(e->mu);
	e->st[(size_t)stream].active = active != 0;
	e->active_dirty = true;
	return MI_OK;
}

int mi_equalizer_dump(mi_equalizer *e, int stream, float *h_dst, int cap) { // equalizer.c:317-328
	MI_CHECK_ARG(e && h_dst && stream >= 0 && stream < e->nstreams && cap >= e->nfft / 2);
	std::lock_guard<std::mutex> lk(e->mu);
	const HostEq &s = e->st[(size_t)stream];
	float *t = h_dst;
	*t++ = s.spectrum[0];
	for (int i = 1; i < e->nfft && (t - h_dst) < cap; i += 2) *t++ = ((float)s.spectrum[(size_t)i] * (float)e->nfft) / 1.0f;
	return MI_OK;
}

int mi_equalizer_get_taps(mi_equalizer *e, int stream, float *h_dst, int cap) {
	MI_CHECK_ARG(e && h_dst && stream >= 0 && stream < e->nstreams && cap >= e->nfft);
	std::lock_guard<std::mutex> lk(e->mu);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	int rc = upload_stale(e);
	if (rc != MI_OK) return rc;
	MI_HIP(hipStreamSynchronize(e->ctx->stream));
	MI_HIP(hipMemcpy(h_dst, e->d_taps + (size_t)stream * e->nfft, sizeof(float) * (size_t)e->nfft,
	                 hipMemcpyDeviceToHost));
	return MI_OK;
}

int mi_equalizer_set_taps(mi_equalizer *e, int stream, const float *h_taps, int n) {
	MI_CHECK_ARG(e && h_taps && stream >= 0 && stream < e->nstreams && n == e->nfft);
	std::lock_guard<std::mutex> lk(e->mu);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(e->ctx->stream));
	MI_HIP(hipMemcpy(e->d_taps + (size_t)stream * e->nfft, h_taps, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
	e->st[(size_t)stream].stale = false;
	return MI_OK;
}

int mi_equalizer_get_history(mi_equalizer *e, int stream, int16_t *h_hist, int n) {
	MI_CHECK_ARG(e && h_hist && stream >= 0 && stream < e->nstreams && n == e->nfft);
	std::lock_guard<std::mutex> lk(e->mu);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(e->ctx->stream));
	MI_HIP(hipMemcpy(h_hist, e->d_hist + (size_t)stream * e->nfft, sizeof(int16_t) * (size_t)n, hipMemcpyDeviceToHost));
	return MI_OK;
}

int mi_equalizer_set_history(mi_equalizer *e, int stream, const int16_t *h_hist, int n) {
	MI_CHECK_ARG(e && stream >= 0 && stream < e->nstreams && n == e->nfft);
	std::lock_guard<std::mutex> lk(e->mu);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(e->ctx->stream));
	if (h_hist) MI_HIP(hipMemcpy(e->d_hist + (size_t)stream * e->nfft, h_hist, sizeof(int16_t) * (size_t)n, hipMemcpyHostToDevice));
	else MI_HIP(hipMemset(e->d_hist + (size_t)stream * e->nfft, 0, sizeof(int16_t) * (size_t)n));
	return MI_OK;
}

int mi_equalizer_process(mi_equalizer *e, int16_t *d_samples, int nsamples, int stride) {
	return mi_equalizer_process_masked(e, d_samples, nsamples, stride, nullptr);
}

int mi_equalizer_process_masked(mi_equalizer *e, int16_t *d_samples, int nsamples, int stride,
                                const int32_t *d_nsamples) {
	MI_CHECK_ARG(e && d_samples && nsamples > 0 && stride >= nsamples);
	if (nsamples > 8192) {
		mi::set_error("block of %d samples exceeds the equalizer kernel's LDS staging (max 8192)", nsamples);
		return MI_ENOTSUP;
	}
	std::lock_guard<std::mutex> lk(e->mu);
	if (e->ctx->activate() != MI_OK) return MI_ENODEV;
	int rc = upload_stale(e);
	if (rc != MI_OK) return rc;
	EqArgs a;
	a.samples = d_samples;
	a.hist = e->d_hist;
	a.taps = e->d_taps;
	a.active = e->d_active;
	a.nper = d_nsamples;
	a.nstreams = e->nstreams;
	a.nsamples = nsamples;
	a.stride = stride;
	a.ord = e->nfft;
	hipStream_t st = e->ctx->stream;
	const size_t lds_pk = (size_t)(e->nfft - 1 + nsamples + 16 + 3) * sizeof(float);
	switch (e->nfft) {
		case 128: hipLaunchKernelGGL(equalizer_pk_kernel<128>, dim3(e->nstreams), dim3(64), lds_pk, st, a); break;
		case 256: hipLaunchKernelGGL(equalizer_pk_kernel<256>, dim3(e->nstreams), dim3(64), lds_pk, st, a); break;
		default: hipLaunchKernelGGL(equalizer_pk_kernel<512>, dim3(e->nstreams), dim3(64), lds_pk, st, a); break;
	}
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_equalizer_process_host(mi_equalizer *e, int16_t *h_samples, int nsamples, int stride) {
	MI_CHECK_ARG(e && h_samples);
	mi_ctx *c = e->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t b = (size_t)e->nstreams * stride * sizeof(int16_t);
	void *d;
	int rc;
	if ((rc = c->ensure_scratch(0, b, &d)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(d, h_samples, b, hipMemcpyHostToDevice, c->stream));
	rc = mi_equalizer_process(e, (int16_t *)d, nsamples, stride);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_samples, d, b, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_equalizer(reinterpret_cast<const void *>(&equalizer_pk_kernel<512>));
