// aec.hip -- batched MSSpeexEC core for gfx950: MDF echo canceller + residual
// echo / noise post-filter, one frame of every stream per launch.
// Built with -ffp-contract=off.
//
// Replaces, per frame, speex_echo_cancellation + speex_preprocess_run as called
// from src/audiofilters/speexec.c:297-298 (libspeexdsp mdf.c / preprocess.c,
// un-vendored): multidelay block frequency-domain adaptive filter with M
// blocks of N = 2*frame points, a foreground/background filter pair with
// two-path control, AUMDF constraint on block 0 + one round-robin block,
// adaptive per-bin step size; then the Ephraim-Malah style post-filter with the
// residual-echo estimate of the echo state.
//
// Mapping (MI355X-first, not how the CPU code is laid out; kernels in aec_wave.hpp):
//   * one 64-lane wavefront per stream, each lane owning F/64 consecutive bins and time
//     samples in registers (no workgroup barriers; the FFT is the only cross-lane exchange);
//   * the far-end spectral history X is a RING in HBM (no (M+1)*N memmove per
//     frame) and every spectrum is stored bin-interleaved ([DC,Nyq],[re1,im1],..)
//     so each lane does one aligned 8-byte access per block, 2 KB contiguous
//     per workgroup;
//   * X, W and foreground are streamed ONCE per frame in a single fused pass
//     (foreground response, background gradient+update, background response):
//     (M+1)+M+M block reads and M block writes = the algorithmic
//     ~202 KB/frame at 48 kHz / 128 ms tail -- this pass is what the HBM
//     roofline prices; the blocks the AUMDF constraint touches are finished
//     first so the accumulation order over blocks stays the library's;
//   * FFTs (12 per frame) run in LDS with the same radix-4/2 decomposition,
//     twiddles and operation order as the kiss_fft float build, and the
//     decision scalars (Sff, See, Dbf, ...) are accumulated in the library's
//     serial order by single lanes, so the two-path decisions match the CPU
//     restatement; only the per-block weight norms behind the proportional
//     step use a tree reduction.
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>

#pragma clang fp contract(off)


namespace {

constexpr int NB_BANDS = 24;
constexpr int MAX_STAGES = 8;

struct FftPlan {
	int nstages;
	int p[MAX_STAGES], m[MAX_STAGES], fs[MAX_STAGES]; // execution order (deepest first)
};

// device-resident constant tables shared by all streams
struct AecTables {
	const float2 *tw;      // [F] forward twiddles of the F-point complex FFT
	const float2 *super;   // [F] forward super-twiddles of the real transform
	const uint16_t *perm;  // [F] digit permutation
	const float *hann;     // [N] MDF window
	const float *pwin;     // [N] post-filter analysis/synthesis window
	const int16_t *bleft;  // [F] filterbank: left band of each bin
	const float *bfl;      // [F] left weight
	const float *bfr;      // [F] right weight
	const int16_t *brange; // [24][4]: first/last(+1) bin with right==b, first/last(+1) bin with left==b
};

// per-stream scalar state (kept as one 128-byte record)
struct AecScalars {
	float Davg1, Davg2, Dvar1, Dvar2;
	float Pey, Pyy, sum_adapt, leak_estimate;
	float memX, memD, memE, notch0, notch1;
	int adapted, saturated, screwed_up, cancel_count, xhead;
	int nb_adapt, min_count;
	// event counters since the last reset of the stream (diagnostics: how often the data-dependent paths run)
	int fg_updates;   // foreground := background (two-path control)
	int bg_resets;    // background := foreground
	int state_resets; // speex_echo_state_reset after persistent divergence
	int frames;       // frames cancelled
	// the last frame asked for foreground := background and no pass over the filter has run since: the foreground array
	// in HBM is stale, the background array is what it has to become (aec_tick.hpp: pendingFG)
	int fg_pending;
	int bg_pending; // the same the other way round: background := foreground waits for the next pass
	int wsel;       // which half of the stream's filter memory holds the background (the other one the foreground)
	int pad_[1];
};
static_assert(sizeof(AecScalars) == 112, "scalar record");

struct AecArgs {
	const int16_t *mic, *ref;
	int16_t *out;
	const uint8_t *run;   // per-frame entry: 0 = the stream has no frame this call
	const uint8_t *count; // per-tick entry: frames ready for the stream (0 .. max_frames), rows hold them back to back
	int max_frames;
	// FIFO entry (fmic.ring != NULL): the tick's new blocks are appended to the microphone / far-end rings, every whole
	// frame they then hold is processed, the cleaned frames are appended to the output ring -- all inside the kernel
	FifoView fmic, fref, fout;
	const int16_t *mic_tick, *ref_tick; // [nstreams][*_tick_stride], tick_len new samples per stream
	int tick_len, mic_tick_stride, ref_tick_stride;
	const int32_t *ref_len; // nullable: per-stream length of the far-end block (0 .. tick_len)
	// MSResample folded in (rs_in != NULL): the tick's microphone block is NOT read from mic_tick; the wave up-samples the
	// leg's rs_in_len input samples by rs_den itself (the resampler's own tile FIR, history and table) and queues the result
	const int16_t *rs_in;
	int16_t *rs_hist;
	const float *rs_table;
	int rs_in_len, rs_in_stride, rs_hist_stride, rs_den;
	uint8_t *count_out; // nullable: frames each stream ran
	int stride, nstreams, M, flags;
	int first;             // first stream of this launch (a launch may cover a chunk of the batch)
	// FIFO entry: leg order of this launch / of the next one, [2][8][cap8] (aec_tick.hpp: TickOrder); null = block b serves leg b
	int *order, *ctl;
	int cap8;
	float *X;              // [nstreams][M+1][N] far-end spectra (ring)
	float *WF;             // [nstreams][2][M][N]: background filter, then foreground filter, of each stream back to back
	float *small;          // [nstreams][small_stride]
	AecScalars *scal;      // [nstreams]
	int small_stride;
	float spec_average, beta0, beta_max, notch_radius, ss, ss_1;
	int sampling_rate;
	AecTables t;
};

typedef float v2f __attribute__((ext_vector_type(2))); // a register pair for v_pk_*_f32

// Complex arithmetic on register pairs, three packed instructions per product.  The halves a packed instruction reads and
// the signs it applies are operand modifiers (op_sel / op_sel_hi / neg_lo / neg_hi), so the scalar form's four products,
// their rounding and the order of the two additions are kept exactly:
//   p = (a.x b.x, a.y b.x)   q = (a.y b.y, a.x b.y)   a b = (p.x - q.x, p.y + q.y)
// (left to itself the compiler forms both p + q and p - q and moves one half over: five instructions and a wait state)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b) {
	v2f p, q, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));
	asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));
	return r;
}
// conj(a) b = (a.x b.x + a.y b.y, (-a.y) b.x + a.x b.y)
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f b) {
	v2f p, q, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));
	asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(p), "v"(q));
	return r;
}
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
	const v2f r = pk_cmul((v2f){a.x, a.y}, (v2f){b.x, b.y}); // (a.x b.x - a.y b.y, a.y b.x + a.x b.y)
	return make_float2(r.x, r.y);
}

// kiss_fft factorisation (4s first, then 2), stages listed deepest first:
// F=256: radix 4,4,4,4 with m = 1,4,16,64;  F=128: radix 2 (m=1) then 4,4,4 with m = 2,8,32;
// F=64 (8 kHz): radix 4,4,4 with m = 1,4,16.
__host__ __device__ constexpr int plan_n(int F) { return F == 64 ? 3 : 4; }
__host__ __device__ constexpr int plan_p(int F, int s) { return (F == 128 && s == 0) ? 2 : 4; }
__host__ __device__ constexpr int plan_m(int F, int s) {
	return F == 128 ? (s == 0 ? 1 : (s == 1 ? 2 : (s == 2 ? 8 : 32))) : (s == 0 ? 1 : (s == 1 ? 4 : (s == 2 ? 16 : 64)));
}
__host__ __device__ constexpr int plan_fs(int F, int s) {
	return F == 64 ? (s == 0 ? 16 : (s == 1 ? 4 : 1)) : (s == 0 ? 64 : (s == 1 ? 16 : (s == 2 ? 4 : 1)));
}

__device__ __forceinline__ float rdlane(float v, int l) {
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ int16_t word2int(float x) {
	if (x < -32767.5f) return (int16_t)-32768;
	if (x > 32766.5f) return (int16_t)32767;
	return (int16_t)(int)floor(.5 + (double)x);
}

__device__ __forceinline__ float qcurve(float x) { return 1.f / (1.f + .15f / x); }

// (float)sqrt((double)x), which is what the library's spx_sqrt gives for a float argument, equals the correctly
// rounded single-precision square root bit for bit: rounding a square root twice is innocuous once the wider
// format has at least 2 p + 2 digits (53 >= 50).  The float form is a third of the instructions and none of them f64.
__device__ __forceinline__ float sqrt_via_double(float x) { return sqrtf(x); } // correctly rounded (hipcc default); __fsqrt_rn is the native approximation

__constant__ float kHypergeom[21] = {0.82157f, 1.02017f, 1.20461f, 1.37534f, 1.53363f, 1.68092f, 1.81865f,
                                     1.94811f, 2.07038f, 2.18638f, 2.29688f, 2.40255f, 2.50391f, 2.60144f,
                                     2.69551f, 2.78647f, 2.87458f, 2.96015f, 3.04333f, 3.12431f, 3.20326f};

__device__ float hypergeom_gain(float xx) {
	const float *table = kHypergeom;
	const float x = xx;
	const float integer = (float)floor(2 * x);
	const int ind = (int)integer;
	if (ind < 0) return 1.f;
	if (ind > 19) return (float)(1.f * (1 + .1296 / x));
	const float frac = 2 * x - integer;
	return (float)(((1 - frac) * table[ind] + frac * table[ind + 1]) / sqrt((double)(x + .0001f)));
}

// filterbank_compute_bank32 in the library's accumulation order: one lane per band.
// pl/pr hold the per-bin products filter_left*ps / filter_right*ps (formed lane-parallel).
template <int F>
__device__ __forceinline__ float band_sum(const AecTables &t, int b, const float *pl, const float *pr) {
	float mel = 0;
	const int r0 = t.brange[4 * b + 0], r1 = t.brange[4 * b + 1];
	const int l0 = t.brange[4 * b + 2], l1 = t.brange[4 * b + 3];
	int i = r0;
	for (; i + 4 <= r1; i += 4) {
		const float a0 = pr[i], a1 = pr[i + 1], a2 = pr[i + 2], a3 = pr[i + 3];
		mel += a0;
		mel += a1;
		mel += a2;
		mel += a3;
	}
	for (; i < r1; ++i) mel += pr[i];
	i = l0;
	for (; i + 4 <= l1; i += 4) {
		const float a0 = pl[i], a1 = pl[i + 1], a2 = pl[i + 2], a3 = pl[i + 3];
		mel += a0;
		mel += a1;
		mel += a2;
		mel += a3;
	}
	for (; i < l1; ++i) mel += pl[i];
	return mel;
}
// Three spectra through the filterbank at once (the post-filter needs the echo estimate's, the frame's and the noise
// estimate's band energies): each band's sum is a serial chain of up to ~180 additions fed by LDS reads -- latency, not
// work -- so the three independent chains run interleaved in one loop and cost what one of them costs.  sp[j] = products
// of spectrum j, left halves at [0, F), right halves at [F, 2F).  Same additions in the same order as band_sum.
template <int F>
__device__ __forceinline__ void band_sum3(const AecTables &t, int b, const float *s0, const float *s1, const float *s2, float &m0, float &m1,
                                          float &m2) {
	float a = 0, c = 0, d = 0;
	const int r0 = t.brange[4 * b + 0], r1 = t.brange[4 * b + 1];
	const int l0 = t.brange[4 * b + 2], l1 = t.brange[4 * b + 3];
	auto range = [&](int off, int i, int e) {
		for (; i + 4 <= e; i += 4) {
			const float a0 = s0[off + i], a1 = s0[off + i + 1], a2 = s0[off + i + 2], a3 = s0[off + i + 3];
			const float c0 = s1[off + i], c1 = s1[off + i + 1], c2 = s1[off + i + 2], c3 = s1[off + i + 3];
			const float d0 = s2[off + i], d1 = s2[off + i + 1], d2 = s2[off + i + 2], d3 = s2[off + i + 3];
			a += a0, c += c0, d += d0;
			a += a1, c += c1, d += d1;
			a += a2, c += c2, d += d2;
			a += a3, c += c3, d += d3;
		}
		for (; i < e; ++i) a += s0[off + i], c += s1[off + i], d += s2[off + i];
	};
	range(F, r0, r1); // the bins this band is the RIGHT neighbour of come first (lower bins), then those it is the left one of
	range(0, l0, l1);
	m0 = a, m1 = c, m2 = d;
}

#include "resample_tile.hpp"
#include "aec_wave.hpp"
#include "aec_tick.hpp"
#include "aec_group.hpp"

// ---- debug: forward/inverse transform of one 2F-point frame per block (parity of the FFT itself)
template <int F>
__global__ __launch_bounds__(64) void fft_debug_kernel(const float *in, float *out, int inverse, AecTables t) {
	__shared__ WLds<F> L;
	constexpr int K = F / 64;
	const int lane = threadIdx.x, e0 = lane * K;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.tw[e0 + k] = t.tw[e0 + k];
		L.super[e0 + k] = t.super[e0 + k];
		L.perm[e0 + k] = t.perm[e0 + k];
	}
	const float *src = in + (size_t)blockIdx.x * 2 * F;
	float *dst = out + (size_t)blockIdx.x * 2 * F;
	if (!inverse) {
		float lo[K], hi[K];
		load_vec<K>(src + e0, lo);
		load_vec<K>(src + F + e0, hi);
		store_vec<K>(L.tbuf + e0, lo);
		store_vec<K>(L.tbuf + F + e0, hi);
		float2 r[K];
		w_rfft_forward<F>(L, t, r);
		store_bins<K>(reinterpret_cast<float2 *>(dst) + e0, r);
	} else {
		float2 r[K];
		load_bins<K>(reinterpret_cast<const float2 *>(src) + e0, r);
		w_rfft_inverse<F>(L, t, r);
		float lo[K], hi[K];
		load_vec<K>(w_time(L) + e0, lo);
		load_vec<K>(w_time(L) + F + e0, hi);
		store_vec<K>(dst + e0, lo);
		store_vec<K>(dst + F + e0, hi);
	}
}

// the same through the transforms of the several-legs-per-wavefront form (aec_group.hpp): frame blockIdx.x * LPW + leg
template <int F>
__global__ __launch_bounds__(64) void fft_debug_group_kernel(const float *in, float *out, int inverse, int nframes, AecTables t) {
	using GP = Grp<F>;
	constexpr int K = GP::K, LPW = GP::LPW;
	__shared__ GLds<F> LW;
	const int lane = GP::lane(), e0 = lane * K;
	auto &L = LW.leg[GP::index()];
	for (int i = threadIdx.x; i < F; i += 64) LW.tw[i] = t.tw[i], LW.super[i] = t.super[i], LW.perm[i] = t.perm[i];
	WSYNC();
	int fr = (int)blockIdx.x * LPW + GP::index();
	if (fr >= nframes) fr = nframes - 1; // (every lane takes part in the transforms; the spare legs redo the last frame)
	const float *src = in + (size_t)fr * 2 * F;
	float *dst = out + (size_t)fr * 2 * F;
	if (!inverse) {
		float lo[K], hi[K];
		load_vec<K>(src + e0, lo);
		load_vec<K>(src + F + e0, hi);
		store_vec<K>(L.tbuf + e0, lo);
		store_vec<K>(L.tbuf + F + e0, hi);
		float2 r[K];
		g_rfft_forward<F>(L, LW, r);
		store_bins<K>(reinterpret_cast<float2 *>(dst) + e0, r);
	} else {
		float2 r[K];
		load_bins<K>(reinterpret_cast<const float2 *>(src) + e0, r);
		g_rfft_inverse<F>(L, LW, r);
		float lo[K], hi[K];
		load_vec<K>(w_time(L) + e0, lo);
		load_vec<K>(w_time(L) + F + e0, hi);
		store_vec<K>(dst + e0, lo);
		store_vec<K>(dst + F + e0, hi);
	}
}

} // namespace

struct mi_aec {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, rate = 0, F = 0, N = 0, M = 0;
	float *d_X = nullptr, *d_WF = nullptr, *d_small = nullptr;
	// stream s owns two halves of M N floats at d_WF + 2 s M N: one holds the background filter, the other the foreground;
	// AecScalars::wsel says which (the halves swap roles on update_foreground, aec_tick.hpp)
	size_t wf_stride() const { return (size_t)2 * M * N; }
	float *half(int s, int h) const { return d_WF + wf_stride() * s + (size_t)h * M * N; }
	AecScalars *d_scal = nullptr;
	void *d_tables = nullptr;
	AecTables t;
	FftPlan plan;
	int small_stride = 0;
	int *d_order = nullptr, *d_ctl = nullptr; // leg order of the FIFO entry's launches (aec_tick.hpp: TickOrder)
	int cap8 = 0;
	std::vector<float> h_prop0;
	float spec_average, beta0, beta_max, notch_radius, ss, ss_1;
};

namespace {

float to_bark(float n) { return (float)(13.1f * atan(.00074f * n) + 2.24f * atan(n * n * 1.85e-8f) + 1e-4f * n); }

void conj_window(float *w, int len) {
	for (int i = 0; i < len; i++) {
		float tmp, x = (4.f * i) / len;
		int inv = 0;
		if (x < 1.f) {
		} else if (x < 2.f) {
			x = 2.f - x;
			inv = 1;
		} else if (x < 3.f) {
			x = x - 2.f;
			inv = 1;
		} else {
			x = 2.f - x + 2.f;
		}
		x = 1.271903f * x;
		tmp = (float)(.5f - .5f * cos(.5f * M_PI * x));
		tmp = tmp * tmp;
		if (inv) tmp = 1.0f - tmp;
		w[i] = (float)sqrt(tmp);
	}
}

template <typename T>
size_t put(std::vector<uint8_t> &blob, const std::vector<T> &v) {
	size_t off = mi::round_up(blob.size(), 16);
	blob.resize(off + v.size() * sizeof(T));
	memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
	return off;
}

int build_tables(mi_aec *a) {
	const int F = a->F, N = a->N;
	// kiss factorisation of F: 4s then 2 (F is a power of two)
	std::vector<int> radix, rest;
	int left = F;
	while (left > 1) {
		const int p = (left % 4 == 0) ? 4 : 2;
		left /= p;
		radix.push_back(p);
		rest.push_back(left);
	}
	std::vector<int> stride(radix.size());
	int f = 1;
	for (size_t L = 0; L < radix.size(); ++L) {
		stride[L] = f;
		f *= radix[L];
	}
	if ((int)radix.size() > MAX_STAGES) return MI_ENOTSUP;
	a->plan.nstages = (int)radix.size();
	for (int s = 0; s < a->plan.nstages; ++s) { // deepest stage first
		const int L = a->plan.nstages - 1 - s;
		a->plan.p[s] = radix[(size_t)L];
		a->plan.m[s] = rest[(size_t)L];
		a->plan.fs[s] = stride[(size_t)L];
	}
	// the kernels carry this plan as compile-time constants; make sure both agree
	if (a->plan.nstages != plan_n(F)) return MI_ENOTSUP;
	for (int s = 0; s < plan_n(F); ++s)
		if (a->plan.p[s] != plan_p(F, s) || a->plan.m[s] != plan_m(F, s) || a->plan.fs[s] != plan_fs(F, s)) {
			mi::set_error("FFT plan mismatch for F=%d stage %d", F, s);
			return MI_ENOTSUP;
		}
	std::vector<uint16_t> perm((size_t)F);
	for (int o = 0; o < F; ++o) {
		int rem = o, src = 0;
		for (size_t L = 0; L < radix.size(); ++L) {
			const int j = rem / rest[L];
			rem -= j * rest[L];
			src += j * stride[L];
		}
		perm[(size_t)o] = (uint16_t)src;
	}
	std::vector<float2> tw((size_t)F), super((size_t)F);
	const double pi = 3.14159265358979323846264338327;
	for (int k = 0; k < F; ++k) {
		const double ph = (-2 * pi / F) * k;
		tw[(size_t)k] = make_float2((float)cos(ph), (float)sin(ph));
		const double ps = -(pi * (((double)k) / F + .5));
		super[(size_t)k] = make_float2((float)cos(ps), (float)sin(ps));
	}
	std::vector<float> hann((size_t)N), pwin((size_t)N);
	for (int i = 0; i < N; i++) hann[(size_t)i] = (float)(.5 - .5 * cos(2 * M_PI * i / N));
	conj_window(pwin.data(), N);
	// filterbank (filterbank_new, Bark scale, 24 bands over F bins)
	std::vector<int16_t> bleft((size_t)F, 0);
	std::vector<float> bfl((size_t)F, 0.f), bfr((size_t)F, 0.f);
	{
		const float sampling = (float)a->rate;
		const float df = sampling / (float)(2 * F);
		const float max_mel = to_bark(sampling / 2);
		const float mel_interval = max_mel / (float)(NB_BANDS - 1);
		for (int i = 0; i < F; i++) {
			const float curr_freq = (float)i * df;
			const float mel = to_bark(curr_freq);
			if (mel > max_mel) break;
			int id1 = (int)(floor(mel / mel_interval));
			float val;
			if (id1 > NB_BANDS - 2) {
				id1 = NB_BANDS - 2;
				val = 1.0f;
			} else {
				val = (mel - id1 * mel_interval) / mel_interval;
			}
			bleft[(size_t)i] = (int16_t)id1;
			bfl[(size_t)i] = 1.0f - val;
			bfr[(size_t)i] = val;
		}
	}
	std::vector<int16_t> brange((size_t)NB_BANDS * 4, 0);
	for (int b = 0; b < NB_BANDS; ++b) {
		int r0 = F, r1 = 0, l0 = F, l1 = 0;
		for (int i = 0; i < F; ++i) {
			if (bleft[(size_t)i] + 1 == b) {
				r0 = std::min(r0, i);
				r1 = std::max(r1, i + 1);
			}
			if (bleft[(size_t)i] == b) {
				l0 = std::min(l0, i);
				l1 = std::max(l1, i + 1);
			}
		}
		if (r1 == 0) r0 = 0;
		if (l1 == 0) l0 = 0;
		brange[(size_t)4 * b + 0] = (int16_t)r0;
		brange[(size_t)4 * b + 1] = (int16_t)r1;
		brange[(size_t)4 * b + 2] = (int16_t)l0;
		brange[(size_t)4 * b + 3] = (int16_t)l1;
	}
	// the bank's bins are monotone in band index, so every band's contributors are two
	// contiguous runs and "right" contributions (lower bins) precede "left" ones
	for (int i = 1; i < F; ++i)
		if (bleft[(size_t)i] < bleft[(size_t)i - 1] && bfl[(size_t)i] + bfr[(size_t)i] != 0.f) return MI_ENOTSUP;

	std::vector<uint8_t> blob;
	const size_t o_tw = put(blob, tw), o_su = put(blob, super), o_pe = put(blob, perm), o_ha = put(blob, hann),
	             o_pw = put(blob, pwin), o_bl = put(blob, bleft), o_fl = put(blob, bfl), o_fr = put(blob, bfr),
	             o_br = put(blob, brange);
	MI_HIP(hipMalloc(&a->d_tables, blob.size()));
	MI_HIP(hipMemcpy(a->d_tables, blob.data(), blob.size(), hipMemcpyHostToDevice));
	uint8_t *base = (uint8_t *)a->d_tables;
	a->t.tw = (const float2 *)(base + o_tw);
	a->t.super = (const float2 *)(base + o_su);
	a->t.perm = (const uint16_t *)(base + o_pe);
	a->t.hann = (const float *)(base + o_ha);
	a->t.pwin = (const float *)(base + o_pw);
	a->t.bleft = (const int16_t *)(base + o_bl);
	a->t.bfl = (const float *)(base + o_fl);
	a->t.bfr = (const float *)(base + o_fr);
	a->t.brange = (const int16_t *)(base + o_br);
	return MI_OK;
}

template <int F>
int init_state(mi_aec *a, int first, int count) {
	using SL = TickLayout<F>;
	std::vector<float> small((size_t)a->small_stride, 0.f);
	for (int i = 0; i < F; ++i) small[(size_t)SL::POWER1 + i] = 1.0f;
	small[(size_t)SL::TAIL + 1] = 1.0f; // power_1[F]
	for (int i = 0; i < a->M; ++i) small[(size_t)SL::PROP + i] = a->h_prop0[(size_t)i];
	for (int i = 0; i < F; ++i) {
		small[(size_t)SL::NOISE + i] = 1.f;
		small[(size_t)SL::OLDPS + i] = 1.f;
	}
	for (int i = 0; i < NB_BANDS; ++i) small[(size_t)SL::OLDPS_B + i] = 1.f;
	AecScalars sc;
	memset(&sc, 0, sizeof(sc));
	sc.Pey = sc.Pyy = 1.0f;
	std::vector<float> all((size_t)count * a->small_stride);
	std::vector<AecScalars> scs((size_t)count, sc);
	for (int i = 0; i < count; ++i) memcpy(all.data() + (size_t)i * a->small_stride, small.data(), small.size() * sizeof(float));
	MI_HIP(hipStreamSynchronize(a->ctx->stream));
	MI_HIP(hipMemcpy(a->d_small + (size_t)first * a->small_stride, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(a->d_scal + first, scs.data(), scs.size() * sizeof(AecScalars), hipMemcpyHostToDevice));
	const size_t xn = (size_t)(a->M + 1) * a->N;
	MI_HIP(hipMemset(a->half(first, 0), 0, (size_t)count * a->wf_stride() * sizeof(float)));
	MI_HIP(hipMemset(a->d_X + first * xn, 0, (size_t)count * xn * sizeof(float)));
	return MI_OK;
}

} // namespace

extern "C" {

int mi_aec_framesize(int framesize_at_8000, int sample_rate) { // speexec.c:171-180
	const int newsize = (framesize_at_8000 * sample_rate) / 8000;
	int n = 1, next;
	while ((next = n << 1) <= newsize) n = next;
	return n;
}

static bool aec_group_form_on();
int mi_aec_create(mi_ctx *ctx, int nstreams, int sample_rate, int frame_size, int filter_length, mi_aec **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && sample_rate > 0 && filter_length > 0);
	*out = nullptr;
	if (frame_size != 64 && frame_size != 128 && frame_size != 256) {
		mi::set_error("frame size %d not supported: the filter's 2^k sizing (speexec.c:171-180) gives 64 at 8 kHz, "
		              "128 at 16 kHz and 256 at 32-48 kHz",
		              frame_size);
		return MI_ENOTSUP;
	}
	const int M = (filter_length + frame_size - 1) / frame_size;
	if (M > 64) {
		mi::set_error("filter of %d blocks exceeds the kernel's 64-block limit", M);
		return MI_ENOTSUP;
	}
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_aec *a = new mi_aec();
	a->ctx = ctx;
	a->nstreams = nstreams;
	a->rate = sample_rate;
	a->F = frame_size;
	a->N = 2 * frame_size;
	a->M = M;
	a->spec_average = (float)frame_size / (float)sample_rate;
	a->beta0 = (2.0f * frame_size) / sample_rate;
	a->beta_max = (.5f * frame_size) / sample_rate;
	a->notch_radius = sample_rate < 12000 ? .9f : (sample_rate < 24000 ? .982f : .992f);
	a->ss = (float)(.35 / M);
	a->ss_1 = 1 - a->ss;
	a->h_prop0.resize((size_t)M);
	{
		float sum, decay = (float)exp(-(2.4f / M));
		a->h_prop0[0] = .7f;
		sum = a->h_prop0[0];
		for (int i = 1; i < M; i++) {
			a->h_prop0[(size_t)i] = a->h_prop0[(size_t)i - 1] * decay;
			sum = sum + a->h_prop0[(size_t)i];
		}
		for (int i = M - 1; i >= 0; i--) a->h_prop0[(size_t)i] = (.8f * a->h_prop0[(size_t)i]) / sum;
	}
	a->small_stride = 19 * frame_size + 256; // TickLayout<F>::TOTAL
	const size_t xn = (size_t)(M + 1) * a->N;
	int rc = build_tables(a);
	if (rc != MI_OK) {
		mi_aec_destroy(a);
		return rc;
	}
	if (hipMalloc((void **)&a->d_X, (size_t)nstreams * xn * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&a->d_WF, (size_t)nstreams * a->wf_stride() * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&a->d_small, (size_t)nstreams * a->small_stride * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&a->d_scal, (size_t)nstreams * sizeof(AecScalars)) != hipSuccess) {
		mi::set_error("hipMalloc failed for AEC state (%zu bytes per stream)", mi_aec_state_bytes(a));
		mi_aec_destroy(a);
		return MI_ENOMEM;
	}
	rc = mi_aec_reset(a, 0, nstreams);
	if (rc != MI_OK) {
		mi_aec_destroy(a);
		return rc;
	}
	{ // the FIFO entry's leg lists start as the identity: class c = legs c, c + 8, .. (all entered from the front)
		a->cap8 = (nstreams + 7) / 8 + TickOrder::SLACK;
		std::vector<int> ord((size_t)2 * 8 * a->cap8, 0), ctl(TickOrder::WORDS, 0);
		for (int s = 0; s < nstreams; ++s) ord[(size_t)(s & 7) * a->cap8 + (size_t)(s >> 3)] = s;
		for (int c = 0; c < 8; ++c) ctl[(size_t)c * TickOrder::STRIDE + TickOrder::PLACED] = (nstreams - c + 7) / 8; // parity 0: front = all, back = 0
		if (hipMalloc((void **)&a->d_order, ord.size() * sizeof(int)) != hipSuccess || hipMalloc((void **)&a->d_ctl, ctl.size() * sizeof(int)) != hipSuccess ||
		    hipMemcpy(a->d_order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(a->d_ctl, ctl.data(), ctl.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
			mi::set_error("hipMalloc failed for the canceller's leg lists");
			mi_aec_destroy(a);
			return MI_ENOMEM;
		}
	}
	*out = a;
	return MI_OK;
}

void mi_aec_destroy(mi_aec *a) {
	if (!a) return;
	(void)hipSetDevice(a->ctx->device);
	if (a->d_X) (void)hipFree(a->d_X);
	if (a->d_WF) (void)hipFree(a->d_WF);
	if (a->d_small) (void)hipFree(a->d_small);
	if (a->d_scal) (void)hipFree(a->d_scal);
	if (a->d_tables) (void)hipFree(a->d_tables);
	if (a->d_order) (void)hipFree(a->d_order);
	if (a->d_ctl) (void)hipFree(a->d_ctl);
	delete a;
}

int mi_aec_reset(mi_aec *a, int first, int count) {
	MI_CHECK_ARG(a && first >= 0 && count >= 0 && first + count <= a->nstreams);
	if (count == 0) return MI_OK;
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	return a->F == 256 ? init_state<256>(a, first, count) : (a->F == 128 ? init_state<128>(a, first, count) : init_state<64>(a, first, count));
}

size_t mi_aec_state_bytes(const mi_aec *a) {
	if (!a) return 0;
	return ((size_t)(a->M + 1) * a->N + 2 * (size_t)a->M * a->N + (size_t)a->small_stride) * sizeof(float) +
	       sizeof(AecScalars);
}

static std::atomic<int> g_group_form{-1}; // -1: ask the environment on first use (MSMI355X_AEC_GROUP); mi_debug_aec_group_form sets it

struct AecFifoCall {
	ResamplerView rs; // rs.ok: the microphone block is up-sampled inside the launch from rs_in
	const int16_t *rs_in = nullptr;
	int rs_in_len = 0, rs_in_stride = 0;
	mi_fifo *f_mic, *f_ref, *f_out;
	const int16_t *d_mic_tick, *d_ref_tick;
	int tick_len, mic_stride, ref_stride;
	const int32_t *d_ref_len;
	uint8_t *d_count_out;
	const uint8_t *d_mic_gate = nullptr; // the *_masked forms: 0 = no microphone block for the leg in this launch
};

static int aec_launch(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride, const uint8_t *d_run,
                      const uint8_t *d_count, int max_frames, unsigned flags, const AecFifoCall *fifo) {
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	AecArgs g;
	g.mic = d_mic;
	g.ref = d_ref;
	g.out = d_out;
	g.run = d_run;
	g.count = d_count;
	g.max_frames = max_frames;
	g.fmic = g.fref = g.fout = fifo_view(nullptr);
	g.mic_tick = g.ref_tick = nullptr;
	g.tick_len = g.mic_tick_stride = g.ref_tick_stride = 0;
	g.count_out = nullptr;
	g.ref_len = nullptr;
	g.order = g.ctl = nullptr;
	g.cap8 = a->cap8;
	g.rs_in = nullptr;
	g.rs_hist = nullptr;
	g.rs_table = nullptr;
	g.rs_in_len = g.rs_in_stride = g.rs_hist_stride = g.rs_den = 0;
	if (fifo) {
		g.ref_len = fifo->d_ref_len;
		g.fmic = fifo_view(fifo->f_mic);
		g.fref = fifo_view(fifo->f_ref);
		g.fout = fifo_view(fifo->f_out);
		g.mic_tick = fifo->d_mic_tick;
		g.ref_tick = fifo->d_ref_tick;
		g.tick_len = fifo->tick_len;
		g.mic_tick_stride = fifo->mic_stride;
		g.ref_tick_stride = fifo->ref_stride;
		g.count_out = fifo->d_count_out;
		g.run = fifo->d_mic_gate;
		g.order = a->d_order;
		g.ctl = a->d_ctl;
		if (fifo->rs.ok) {
			g.rs_in = fifo->rs_in;
			g.rs_in_len = fifo->rs_in_len;
			g.rs_in_stride = fifo->rs_in_stride;
			g.rs_hist = fifo->rs.hist;
			g.rs_hist_stride = fifo->rs.hist_stride;
			g.rs_table = fifo->rs.table;
			g.rs_den = fifo->rs.den;
		}
	}
	g.stride = stride;
	g.nstreams = a->nstreams;
	g.M = a->M;
	g.flags = (int)flags;
	g.X = a->d_X;
	g.WF = a->d_WF;
	g.small = a->d_small;
	g.scal = a->d_scal;
	g.small_stride = a->small_stride;
	g.spec_average = a->spec_average;
	g.beta0 = a->beta0;
	g.beta_max = a->beta_max;
	g.notch_radius = a->notch_radius;
	g.ss = a->ss;
	g.ss_1 = a->ss_1;
	g.sampling_rate = a->rate;
	g.t = a->t;
	// One wavefront per stream and TICK: the canceller for every frame the stream has ready and, with MI_AEC_POSTFILTER, the
	// post-filter of the same frames as the wave's tail phase (aec_tick.hpp).  One launch, on the context's stream.
	g.first = 0;
	// the FIFO entry: one workgroup per list SLOT (8 classes x cap8; the few empty slots leave at once), else one per stream
	const dim3 grid(fifo ? 8 * a->cap8 : a->nstreams);
	const int mode = !fifo ? TICK_ROWS : (fifo->rs.ok ? TICK_FIFO_RS : TICK_FIFO);
	// The small frame sizes handed in as rows: several legs per wavefront, one launch per frame of the tick (aec_group.hpp).
	// MSMI355X_AEC_GROUP=0: the one-leg-per-wavefront tick form for them too (A/B; the state in HBM is the same).
	aec_group_form_on();
	// (its lanes take their four samples of a row as one 8-byte access: rows that start off that grid stay on the tick form)
	const bool rows8 = !fifo && (stride & 3) == 0 &&
	                   ((reinterpret_cast<uintptr_t>(d_mic) | reinterpret_cast<uintptr_t>(d_ref) | reinterpret_cast<uintptr_t>(d_out)) & 7) == 0;
	if (rows8 && a->F != 256 && g_group_form.load(std::memory_order_relaxed) > 0) {
		for (int frame = 0; frame < max_frames; ++frame) {
			if (a->F == 128) hipLaunchKernelGGL(aec_group_kernel<128>, dim3((a->nstreams + 1) / 2), dim3(64), 0, a->ctx->stream, g, frame);
			else hipLaunchKernelGGL(aec_group_kernel<64>, dim3((a->nstreams + 3) / 4), dim3(64), 0, a->ctx->stream, g, frame);
			MI_LAUNCH_CHECK();
		}
		return MI_OK;
	}
#define MI_TICK_LAUNCH(FR)                                                                                          \
	do {                                                                                                            \
		if (mode == TICK_FIFO_RS) hipLaunchKernelGGL((aec_tick_kernel<FR, TICK_FIFO_RS>), grid, dim3(64), 0, a->ctx->stream, g); \
		else if (mode == TICK_FIFO) hipLaunchKernelGGL((aec_tick_kernel<FR, TICK_FIFO>), grid, dim3(64), 0, a->ctx->stream, g);  \
		else hipLaunchKernelGGL((aec_tick_kernel<FR, TICK_ROWS>), grid, dim3(64), 0, a->ctx->stream, g);                        \
	} while (0)
	if (a->F == 256) MI_TICK_LAUNCH(256);
	else if (a->F == 128) MI_TICK_LAUNCH(128);
	else MI_TICK_LAUNCH(64);
#undef MI_TICK_LAUNCH
	MI_LAUNCH_CHECK();
	if (fifo) { // the leg lists this launch filled become the ones the next launch serves (aec_tick.hpp: TickOrder)
		hipLaunchKernelGGL(aec_tick_advance_kernel, dim3(1), dim3(64), 0, a->ctx->stream, a->d_ctl);
		MI_LAUNCH_CHECK();
	}
	return MI_OK;
}

int mi_aec_process(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride,
                   const uint8_t *d_run, unsigned flags) {
	MI_CHECK_ARG(a && d_mic && d_ref && d_out && stride >= a->F);
	return aec_launch(a, d_mic, d_ref, d_out, stride, d_run, nullptr, 1, flags, nullptr);
}

int mi_aec_process_frames(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride,
                          const uint8_t *d_count, int max_frames, unsigned flags) {
	MI_CHECK_ARG(a && d_mic && d_ref && d_out && d_count && max_frames >= 1 && max_frames <= MI_AEC_MAX_TICK_FRAMES &&
	             stride >= max_frames * a->F);
	return aec_launch(a, d_mic, d_ref, d_out, stride, nullptr, d_count, max_frames, flags, nullptr);
}

// (The FIFO entries at the small frame sizes -- 8 kHz: F = 64, 16 kHz: F = 128 -- stay on the tick form, one leg per wavefront with the
// FIFOs inside the launch.  Round 5 built the entry out of the group kernel too -- queue, pop frames into rows, one aec_group_kernel launch
// per frame index, push -- and measured it 5 % slower per whole tick (1 010 / 1 559 us against 960 / 1 482 at 65 536 legs: a tick is 1.25
// frames per leg and the second launch serves a quarter of the legs at the price of all): profiles/r05_small_frame_fifo_entry.txt.  That
// path is no longer in the source; rows handed in directly -- mi_aec_process / mi_aec_process_frames -- run the group kernel.)
static bool aec_group_form_on() {
	if (g_group_form.load(std::memory_order_relaxed) < 0) {
		const char *e = getenv("MSMI355X_AEC_GROUP");
		g_group_form.store(e && e[0] == '0' ? 0 : 1, std::memory_order_relaxed);
	}
	return g_group_form.load(std::memory_order_relaxed) > 0;
}
static int aec_launch(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride, const uint8_t *d_run,
                      const uint8_t *d_count, int max_frames, unsigned flags, const AecFifoCall *fifo);

int mi_aec_process_fifos_masked(mi_aec *a, mi_fifo *f_mic, const int16_t *d_mic_tick, int mic_stride, mi_fifo *f_ref,
                                const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, int tick_len, mi_fifo *f_out,
                                int max_frames, unsigned flags, uint8_t *d_count_out, const uint8_t *d_mic_gate) {
	MI_CHECK_ARG(a && f_mic && f_ref && f_out && d_mic_tick && d_ref_tick && tick_len > 0 && mic_stride >= tick_len &&
	             ref_stride >= tick_len && max_frames >= 1 && max_frames <= MI_AEC_MAX_TICK_FRAMES);
	MI_CHECK_ARG(f_mic->nstreams == a->nstreams && f_ref->nstreams == a->nstreams && f_out->nstreams == a->nstreams);
	for (const mi_fifo *f : {f_mic, f_ref, f_out})
		if (f->capacity % a->F || f->capacity < max_frames * a->F) {
			mi::set_error("mi_aec_process_fifos: FIFO capacities must be multiples of the frame size %d (got %d)", a->F, f->capacity);
			return MI_EINVAL;
		}
	AecFifoCall fc;
	fc.f_mic = f_mic, fc.f_ref = f_ref, fc.f_out = f_out;
	fc.d_mic_tick = d_mic_tick, fc.d_ref_tick = d_ref_tick;
	fc.tick_len = tick_len, fc.mic_stride = mic_stride, fc.ref_stride = ref_stride;
	fc.d_ref_len = d_ref_len, fc.d_count_out = d_count_out;
	fc.d_mic_gate = d_mic_gate;
	return aec_launch(a, nullptr, nullptr, nullptr, 0, nullptr, nullptr, max_frames, flags, &fc);
}

int mi_aec_process_fifos(mi_aec *a, mi_fifo *f_mic, const int16_t *d_mic_tick, int mic_stride, mi_fifo *f_ref,
                         const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, int tick_len, mi_fifo *f_out,
                         int max_frames, unsigned flags, uint8_t *d_count_out) {
	return mi_aec_process_fifos_masked(a, f_mic, d_mic_tick, mic_stride, f_ref, d_ref_tick, ref_stride, d_ref_len, tick_len, f_out, max_frames, flags,
	                                   d_count_out, nullptr);
}

int mi_aec_process_fifos_resampled_masked(mi_aec *a, mi_resampler *rs, const int16_t *d_mic_in, int in_len, int in_stride, mi_fifo *f_mic,
                                          mi_fifo *f_ref, const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, mi_fifo *f_out,
                                          int max_frames, unsigned flags, uint8_t *d_count_out, const uint8_t *d_mic_gate) {
	MI_CHECK_ARG(a && rs && d_mic_in && f_mic && f_ref && f_out && d_ref_tick && in_len > 0 && in_stride >= in_len && max_frames >= 1 &&
	             max_frames <= MI_AEC_MAX_TICK_FRAMES);
	MI_CHECK_ARG(f_mic->nstreams == a->nstreams && f_ref->nstreams == a->nstreams && f_out->nstreams == a->nstreams);
	AecFifoCall fc;
	mi_resampler_view(rs, &fc.rs);
	const int tiles = (in_len + 7) / 8;
	// what the in-launch up-sampler handles: an integer ratio with the 48-tap direct table, one (tile, phase) per lane, rows
	// of whole 8-byte groups, every stream on a whole output period; anything else: call the resampler, then mi_aec_process_fifos
	if (!fc.rs.ok || fc.rs.nstreams != a->nstreams || fc.rs.device != a->ctx->device || fc.rs.den * tiles > 64 ||
	    ((in_len | in_stride) & 3) != 0 || (fc.rs.hist_stride >> 2) + (in_len >> 2) > 64 || (reinterpret_cast<uintptr_t>(d_mic_in) & 7) != 0 ||
	    ((in_len * fc.rs.den) & 7) != 0 || (size_t)(48 + in_len + 12) * 4 + (size_t)in_len * fc.rs.den * 2 + (size_t)fc.rs.den * 48 * 4 > (size_t)6 * a->F * 4) {
		mi::set_error("mi_aec_process_fifos_resampled: this resampler / block shape is not one the canceller's launch can up-sample itself "
		              "(integer ratio, 48-tap table, <= 64 (tile, phase) lanes, blocks of whole 8-byte groups)");
		return MI_ENOTSUP;
	}
	const int tick_len = in_len * fc.rs.den;
	MI_CHECK_ARG(ref_stride >= tick_len);
	for (const mi_fifo *f : {f_mic, f_ref, f_out})
		if (f->capacity % a->F || f->capacity < max_frames * a->F || (f->capacity & 7)) {
			mi::set_error("mi_aec_process_fifos_resampled: FIFO capacities must be multiples of the frame size %d (got %d)", a->F, f->capacity);
			return MI_EINVAL;
		}
	fc.rs_in = d_mic_in, fc.rs_in_len = in_len, fc.rs_in_stride = in_stride;
	fc.f_mic = f_mic, fc.f_ref = f_ref, fc.f_out = f_out;
	fc.d_mic_tick = nullptr, fc.d_ref_tick = d_ref_tick;
	fc.tick_len = tick_len, fc.mic_stride = 0, fc.ref_stride = ref_stride;
	fc.d_ref_len = d_ref_len, fc.d_count_out = d_count_out;
	fc.d_mic_gate = d_mic_gate;
	return aec_launch(a, nullptr, nullptr, nullptr, 0, nullptr, nullptr, max_frames, flags, &fc);
}

int mi_aec_process_fifos_resampled(mi_aec *a, mi_resampler *rs, const int16_t *d_mic_in, int in_len, int in_stride, mi_fifo *f_mic,
                                   mi_fifo *f_ref, const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, mi_fifo *f_out,
                                   int max_frames, unsigned flags, uint8_t *d_count_out) {
	return mi_aec_process_fifos_resampled_masked(a, rs, d_mic_in, in_len, in_stride, f_mic, f_ref, d_ref_tick, ref_stride, d_ref_len, f_out,
	                                             max_frames, flags, d_count_out, nullptr);
}

// The re-framing phase of a leg: ticks of tick_len samples against frames of F leave a leg's microphone FIFO at a level
// that cycles through the multiples of g = gcd(tick_len, F) -- at 48 kHz (480 / 256) eight levels, and in the tick a leg
// passes level 0 it has one frame to cancel instead of two.  Legs that all start empty pass it together: seven heavy
// ticks and a light one.  A lead of unit * phase(s) samples of silence in BOTH queues (the echo path between them is
// unchanged) spreads the light ticks evenly: every tick then carries 15/8 frames per leg.
int mi_aec_stagger_info(const mi_aec *a, int tick_len, int *unit, int *phases) {
	MI_CHECK_ARG(a && tick_len > 0);
	int g = a->F, t = tick_len % a->F;
	while (t) { // gcd(F, tick_len): a power of two, F is one
		const int r = g % t;
		g = t;
		t = r;
	}
	const int p = std::min(8, a->F / g);
	if (unit) *unit = a->F / p;
	if (phases) *phases = p;
	return MI_OK;
}

int mi_aec_stagger_fifos(mi_aec *a, mi_fifo *f_mic, mi_fifo *f_ref, int tick_len, int first, int count) {
	MI_CHECK_ARG(a && f_mic && f_ref && first >= 0 && count >= 0 && first + count <= a->nstreams && f_mic->nstreams == a->nstreams &&
	             f_ref->nstreams == a->nstreams);
	int unit = 0, phases = 0;
	int rc = mi_aec_stagger_info(a, tick_len, &unit, &phases);
	if (rc != MI_OK) return rc;
	if (phases < 2) return MI_OK; // ticks of whole frames: nothing to spread
	if ((rc = mi_fifo_push_lead(f_mic, first, count, unit, phases)) != MI_OK) return rc;
	return mi_fifo_push_lead(f_ref, first, count, unit, phases);
}

int mi_aec_process_host(mi_aec *a, const int16_t *h_mic, const int16_t *h_ref, int16_t *h_out, int stride,
                        const uint8_t *h_run, unsigned flags) {
	MI_CHECK_ARG(a && h_mic && h_ref && h_out);
	mi_ctx *c = a->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t b = (size_t)a->nstreams * stride * sizeof(int16_t);
	void *dm, *dr, *dout, *drun = nullptr;
	int rc;
	if ((rc = c->ensure_scratch(0, b, &dm)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, b, &dr)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(2, b, &dout)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(dm, h_mic, b, hipMemcpyHostToDevice, c->stream));
	MI_HIP(hipMemcpyAsync(dr, h_ref, b, hipMemcpyHostToDevice, c->stream));
	MI_HIP(hipMemcpyAsync(dout, h_out, b, hipMemcpyHostToDevice, c->stream));
	if (h_run) {
		if ((rc = c->ensure_scratch(3, (size_t)a->nstreams, &drun)) != MI_OK) return rc;
		MI_HIP(hipMemcpyAsync(drun, h_run, (size_t)a->nstreams, hipMemcpyHostToDevice, c->stream));
	}
	rc = mi_aec_process(a, (const int16_t *)dm, (const int16_t *)dr, (int16_t *)dout, stride, (const uint8_t *)drun,
	                    flags);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_out, dout, b, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

// read-back in the library's packed order [DC, re1, im1, ..., Nyq] for parity tests
int mi_aec_get(mi_aec *a, int stream, const char *what, float *h_dst, int cap) {
	MI_CHECK_ARG(a && what && h_dst && stream >= 0 && stream < a->nstreams);
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(a->ctx->stream));
	const int F = a->F, N = a->N, M = a->M;
	auto unpack = [&](const float *src, float *dst) { // [DC,Nyq,re1,im1,..] -> [DC,re1,im1,..,Nyq]
		dst[0] = src[0];
		dst[N - 1] = src[1];
		for (int k = 1; k < F; ++k) {
			dst[2 * k - 1] = src[2 * k];
			dst[2 * k] = src[2 * k + 1];
		}
	};
	AecScalars sc;
	MI_HIP(hipMemcpy(&sc, a->d_scal + stream, sizeof(sc), hipMemcpyDeviceToHost));
	std::vector<float> small((size_t)a->small_stride);
	MI_HIP(hipMemcpy(small.data(), a->d_small + (size_t)stream * a->small_stride, small.size() * sizeof(float),
	                 hipMemcpyDeviceToHost));
	const int o_e = F, o_pw = 3 * F, o_p1 = 4 * F, o_eh = 5 * F, o_yh = 6 * F, o_ly = 7 * F, o_misc = 19 * F, o_prop = o_misc, o_tail = o_misc + 128;
	auto with_tail = [&](int off, int t) { // the per-bin array plus its Nyquist entry
		std::vector<float> v(small.begin() + off, small.begin() + off + F);
		v.push_back(small[(size_t)o_tail + t]);
		return v;
	};
	std::vector<float> res;
	if (!strcmp(what, "W") || !strcmp(what, "foreground")) {
		std::vector<float> raw((size_t)M * N);
		// a foreground update that waits for the next pass over the filter: the foreground IS the background by then
		const bool want_w = !strcmp(what, "W");
		// the half that holds the background / the foreground; a copy still waiting for its pass is looked through
		const float *src = a->half(stream, ((want_w && !sc.bg_pending) || (!want_w && sc.fg_pending)) ? sc.wsel : sc.wsel ^ 1);
		MI_HIP(hipMemcpy(raw.data(), src, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
		res.resize(raw.size());
		for (int j = 0; j < M; ++j) unpack(raw.data() + (size_t)j * N, res.data() + (size_t)j * N);
	} else if (!strcmp(what, "X")) {
		std::vector<float> raw((size_t)(M + 1) * N);
		MI_HIP(hipMemcpy(raw.data(), a->d_X + (size_t)stream * (M + 1) * N, raw.size() * sizeof(float),
		                 hipMemcpyDeviceToHost));
		res.resize(raw.size());
		for (int j = 0; j <= M; ++j) // logical block j = ring slot (head + j) % (M+1)
			unpack(raw.data() + (size_t)((sc.xhead + j) % (M + 1)) * N, res.data() + (size_t)j * N);
	} else if (!strcmp(what, "E")) {
		res.resize((size_t)N);
		unpack(small.data() + o_e, res.data());
	} else if (!strcmp(what, "power")) res = with_tail(o_pw, 0);
	else if (!strcmp(what, "power_1")) res = with_tail(o_p1, 1);
	else if (!strcmp(what, "Eh")) res = with_tail(o_eh, 2);
	else if (!strcmp(what, "Yh")) res = with_tail(o_yh, 3);
	else if (!strcmp(what, "last_y")) res.assign(small.begin() + o_ly, small.begin() + o_ly + N); // [older | newest]
	else if (!strcmp(what, "prop")) res.assign(small.begin() + o_prop, small.begin() + o_prop + M);
	else if (!strcmp(what, "order")) { // the list the next launch of the FIFO entry will serve the legs in, class after class
		std::vector<int> ctl(TickOrder::WORDS), ord((size_t)2 * 8 * a->cap8);
		MI_HIP(hipMemcpy(ctl.data(), a->d_ctl, ctl.size() * sizeof(int), hipMemcpyDeviceToHost));
		MI_HIP(hipMemcpy(ord.data(), a->d_order, ord.size() * sizeof(int), hipMemcpyDeviceToHost));
		// class after class: the entries from the front, then those from the back, each class closed by a -1
		const size_t par = (size_t)ctl[TickOrder::GLOBAL + TickOrder::PARITY];
		for (int c = 0; c < 8; ++c) {
			const int front = ctl[(size_t)c * TickOrder::STRIDE + TickOrder::PLACED + 2 * par], back = ctl[(size_t)c * TickOrder::STRIDE + TickOrder::PLACED + 2 * par + 1];
			for (int i = 0; i < front; ++i) res.push_back((float)ord[(par * 8 + (size_t)c) * a->cap8 + (size_t)i]);
			for (int i = a->cap8 - back; i < a->cap8; ++i) res.push_back((float)ord[(par * 8 + (size_t)c) * a->cap8 + (size_t)i]);
			res.push_back(-1.f);
		}
	} else if (!strcmp(what, "counters")) res = {(float)sc.fg_updates, (float)sc.bg_resets, (float)sc.state_resets, (float)sc.frames};
	else if (!strcmp(what, "scalars")) {
		res = {sc.Davg1, sc.Davg2, sc.Dvar1, sc.Dvar2, sc.Pey, sc.Pyy, sc.sum_adapt, sc.leak_estimate,
		       (float)sc.adapted, (float)sc.saturated, (float)sc.screwed_up, (float)sc.cancel_count,
		       sc.memX, sc.memD, sc.memE, sc.notch0};
	} else {
		mi::set_error("mi_aec_get: unknown array '%s'", what);
		return MI_EINVAL;
	}
	int n = (int)res.size();
	if (n > cap) n = cap;
	memcpy(h_dst, res.data(), sizeof(float) * (size_t)n);
	return n;
}

// ---- state blob: what SPEEX_ECHO_GET_BLOB / SET_BLOB of the reference's speex fork serve in fetch_config / apply_config
// (src/audiofilters/speexec.c:119-167): a converged canceller survives the end of a call.  The blob is the stream's whole
// state (history ring, both filters, the per-bin arrays, the scalars), so a restored stream continues bit for bit.
namespace {
struct BlobHeader {
	char magic[4];
	uint32_t version, rate, F, M, N, small_stride, scal_bytes;
};
} // namespace

size_t mi_aec_blob_bytes(const mi_aec *a) { return a ? sizeof(BlobHeader) + mi_aec_state_bytes(a) : 0; }

int mi_aec_export_state(mi_aec *a, int stream, void *h_blob, size_t cap) {
	MI_CHECK_ARG(a && h_blob && stream >= 0 && stream < a->nstreams && cap >= mi_aec_blob_bytes(a));
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(a->ctx->stream));
	const size_t wn = (size_t)a->M * a->N, xn = (size_t)(a->M + 1) * a->N, sn = (size_t)a->small_stride;
	BlobHeader h = {{'M', 'I', 'E', 'C'}, MI_AEC_BLOB_VERSION, (uint32_t)a->rate, (uint32_t)a->F, (uint32_t)a->M, (uint32_t)a->N, (uint32_t)sn, (uint32_t)sizeof(AecScalars)};
	uint8_t *p = (uint8_t *)h_blob;
	memcpy(p, &h, sizeof(h));
	p += sizeof(h);
	MI_HIP(hipMemcpy(p, a->d_X + (size_t)stream * xn, xn * 4, hipMemcpyDeviceToHost));
	p += xn * 4;
	AecScalars sc;
	MI_HIP(hipMemcpy(&sc, a->d_scal + stream, sizeof(sc), hipMemcpyDeviceToHost));
	// a filter copy still waiting for its pass is carried out in the blob: the blob holds the filters as they are meant
	MI_HIP(hipMemcpy(p, a->half(stream, sc.bg_pending ? sc.wsel ^ 1 : sc.wsel), wn * 4, hipMemcpyDeviceToHost));
	p += wn * 4;
	MI_HIP(hipMemcpy(p, a->half(stream, sc.fg_pending ? sc.wsel : sc.wsel ^ 1), wn * 4, hipMemcpyDeviceToHost));
	sc.fg_pending = sc.bg_pending = sc.wsel = 0; // in the blob: background first, foreground second, nothing waiting
	p += wn * 4;
	MI_HIP(hipMemcpy(p, a->d_small + (size_t)stream * sn, sn * 4, hipMemcpyDeviceToHost));
	p += sn * 4;
	memcpy(p, &sc, sizeof(sc));
	return MI_OK;
}

int mi_aec_import_state(mi_aec *a, int stream, const void *h_blob, size_t size) {
	MI_CHECK_ARG(a && h_blob && stream >= 0 && stream < a->nstreams);
	BlobHeader h;
	if (size < sizeof(h)) {
		mi::set_error("mi_aec_import_state: blob of %zu bytes is too short", size);
		return MI_EINVAL;
	}
	memcpy(&h, h_blob, sizeof(h));
	if (memcmp(h.magic, "MIEC", 4) != 0) { // e.g. a SPEEX_ECHO_GET_BLOB blob saved by the reference's own MSSpeexEC
		mi::set_error("mi_aec_import_state: not a blob of this library (no 'MIEC' tag): a state saved by another echo canceller "
		              "cannot be loaded, the canceller keeps its current state");
		return MI_EINVAL;
	}
	if (h.version != MI_AEC_BLOB_VERSION) {
		mi::set_error("mi_aec_import_state: blob format version %u, this library reads version %u", h.version, (unsigned)MI_AEC_BLOB_VERSION);
		return MI_EINVAL;
	}
	if (h.rate != (uint32_t)a->rate || h.F != (uint32_t)a->F || h.M != (uint32_t)a->M || h.N != (uint32_t)a->N ||
	    h.small_stride != (uint32_t)a->small_stride || h.scal_bytes != sizeof(AecScalars) || size != mi_aec_blob_bytes(a)) {
		mi::set_error("mi_aec_import_state: the blob was taken from a canceller of another shape (rate %u, frame %u, %u blocks; this one: "
		              "rate %d, frame %d, %d blocks) or is damaged",
		              h.rate, h.F, h.M, a->rate, a->F, a->M);
		return MI_EINVAL;
	}
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(a->ctx->stream));
	const size_t wn = (size_t)a->M * a->N, xn = (size_t)(a->M + 1) * a->N, sn = (size_t)a->small_stride;
	const uint8_t *p = (const uint8_t *)h_blob + sizeof(h);
	MI_HIP(hipMemcpy(a->d_X + (size_t)stream * xn, p, xn * 4, hipMemcpyHostToDevice));
	p += xn * 4;
	MI_HIP(hipMemcpy(a->half(stream, 0), p, wn * 4, hipMemcpyHostToDevice));
	p += wn * 4;
	MI_HIP(hipMemcpy(a->half(stream, 1), p, wn * 4, hipMemcpyHostToDevice));
	p += wn * 4;
	MI_HIP(hipMemcpy(a->d_small + (size_t)stream * sn, p, sn * 4, hipMemcpyHostToDevice));
	p += sn * 4;
	MI_HIP(hipMemcpy(a->d_scal + stream, p, sizeof(AecScalars), hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_aec_copy_state(mi_aec *dst, int dst_first, const mi_aec *src, int src_first, int count) {
	MI_CHECK_ARG(dst && src && count >= 0 && dst_first >= 0 && src_first >= 0 && dst_first + count <= dst->nstreams && src_first + count <= src->nstreams);
	if (dst->rate != src->rate || dst->F != src->F || dst->M != src->M || dst->ctx->device != src->ctx->device) {
		mi::set_error("mi_aec_copy_state: the two batches differ in rate / frame / tail or live on different devices");
		return MI_EINVAL;
	}
	if (count == 0) return MI_OK;
	if (dst->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(src->ctx->stream)); // the source's state as of everything enqueued on its stream so far
	const size_t xn = (size_t)(dst->M + 1) * dst->N, sn = (size_t)dst->small_stride, c = (size_t)count;
	hipStream_t st = dst->ctx->stream;
	MI_HIP(hipMemcpyAsync(dst->d_X + dst_first * xn, src->d_X + src_first * xn, c * xn * 4, hipMemcpyDeviceToDevice, st));
	MI_HIP(hipMemcpyAsync(dst->half(dst_first, 0), src->half(src_first, 0), c * dst->wf_stride() * 4, hipMemcpyDeviceToDevice, st));
	MI_HIP(hipMemcpyAsync(dst->d_small + dst_first * sn, src->d_small + src_first * sn, c * sn * 4, hipMemcpyDeviceToDevice, st));
	MI_HIP(hipMemcpyAsync(dst->d_scal + dst_first, src->d_scal + src_first, c * sizeof(AecScalars), hipMemcpyDeviceToDevice, st));
	return MI_OK;
}

// debug entry (not in the public header): raw transform parity
int mi_debug_fft(mi_aec *a, const float *d_in, float *d_out, int nframes, int inverse) {
	MI_CHECK_ARG(a && d_in && d_out && nframes > 0);
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	if ((inverse & 2) && a->F != 256) { // bit 1: the several-legs-per-wavefront transforms
		if (a->F == 64) hipLaunchKernelGGL(fft_debug_group_kernel<64>, dim3((nframes + 3) / 4), dim3(64), 0, a->ctx->stream, d_in, d_out, inverse & 1, nframes, a->t);
		else hipLaunchKernelGGL(fft_debug_group_kernel<128>, dim3((nframes + 1) / 2), dim3(64), 0, a->ctx->stream, d_in, d_out, inverse & 1, nframes, a->t);
		MI_LAUNCH_CHECK();
		return MI_OK;
	}
	inverse &= 1;
	if (a->F == 64)
		hipLaunchKernelGGL(fft_debug_kernel<64>, dim3(nframes), dim3(64), 0, a->ctx->stream, d_in, d_out, inverse, a->t);
	else if (a->F == 256)
		hipLaunchKernelGGL(fft_debug_kernel<256>, dim3(nframes), dim3(64), 0, a->ctx->stream, d_in, d_out, inverse, a->t);
	else
		hipLaunchKernelGGL(fft_debug_kernel<128>, dim3(nframes), dim3(64), 0, a->ctx->stream, d_in, d_out, inverse, a->t);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

// debug entry: 1 / 0 = the small frame sizes handed in as rows run several legs per wavefront (aec_group.hpp) / one leg per
// wavefront (aec_tick.hpp); the state in HBM is the same, a batch may change between launches
void mi_debug_aec_group_form(int on) { g_group_form.store(on ? 1 : 0, std::memory_order_relaxed); }

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_aec(reinterpret_cast<const void *>(&aec_tick_advance_kernel));
