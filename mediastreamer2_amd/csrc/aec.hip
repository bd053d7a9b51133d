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
// Mapping (MI355X-first, not how the CPU code is laid out):
//   * one workgroup per stream, one lane per complex bin (frame = 256 lanes at
//     48 kHz); all per-bin state lives in registers for the frame;
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
};
static_assert(sizeof(AecScalars) == 80, "scalar record");

struct AecArgs {
	const int16_t *mic, *ref;
	int16_t *out;
	const uint8_t *run;
	int stride, nstreams, M, flags;
	float *X, *W, *FG;     // [nstreams][(M+1) or M][N]
	float *small;          // [nstreams][small_stride]
	AecScalars *scal;      // [nstreams]
	int small_stride;
	float spec_average, beta0, beta_max, notch_radius, ss, ss_1;
	int sampling_rate;
	AecTables t;
};

// offsets (in floats) inside the per-stream small-state block, as multiples of F
// xprev F | E 2F | power F+1.. (padded to 2F each) ...
template <int F>
struct SmallLayout {
	static constexpr int XPREV = 0;
	static constexpr int E = F;                 // 2F
	static constexpr int POWER = 3 * F;         // F+1 (uses 2 slots of F: [0..F) + extra at +F)
	static constexpr int POWER1 = 5 * F;
	static constexpr int EH = 7 * F;
	static constexpr int YH = 9 * F;
	static constexpr int LASTY = 11 * F;        // 2F
	static constexpr int PROP = 13 * F;         // M (<= F)
	static constexpr int WNORM = 14 * F;        // M
	static constexpr int INBUF = 15 * F;
	static constexpr int OUTBUF = 16 * F;
	static constexpr int NOISE = 17 * F;        // F + 24 (2 slots)
	static constexpr int ECHON = 19 * F;
	static constexpr int OLDPS = 21 * F;
	static constexpr int ZETA = 23 * F;
	static constexpr int S_ = 25 * F;
	static constexpr int SMIN = 26 * F;
	static constexpr int STMP = 27 * F;
	static constexpr int TOTAL = 28 * F;
};

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int F>
struct Lds {
	float fin[F];      // raw mic as float
	float v[F];        // notch output
	float input[F];    // pre-emphasised mic
	float tbuf[2 * F]; // time-domain exchange
	float2 zbuf[F];    // complex FFT work
	float spec[2 * F]; // bin-interleaved spectrum exchange
	float efg[F], ybg[F], e1[F], e2[F];
	float2 tw[F], super[F];
	uint16_t perm[F];
	float prop[64];
	float red[64];
	float band[4 * NB_BANDS + 8]; // band-domain scratch (post-filter)
	float wn[64 * 4];             // per-block, per-wave partial weight norms
	int flag[4];
};

// kiss_fft factorisation (4s first, then 2), stages listed deepest first:
// F=256: radix 4,4,4,4 with m = 1,4,16,64;  F=128: radix 2 (m=1) then 4,4,4 with m = 2,8,32.
__host__ __device__ constexpr int plan_p(int F, int s) { return (F == 128 && s == 0) ? 2 : 4; }
__host__ __device__ constexpr int plan_m(int F, int s) {
	return F == 128 ? (s == 0 ? 1 : (s == 1 ? 2 : (s == 2 ? 8 : 32))) : (s == 0 ? 1 : (s == 1 ? 4 : (s == 2 ? 16 : 64)));
}
__host__ __device__ constexpr int plan_fs(int F, int s) { return s == 0 ? 64 : (s == 1 ? 16 : (s == 2 ? 4 : 1)); }

// ---- complex FFT of F points, in place on z (kiss order). All F lanes call it.
template <int F>
__device__ void cfft(Lds<F> &L, const float2 *src, bool inverse) {
	const int tid = threadIdx.x;
	float2 val = src[L.perm[tid]];
	__syncthreads();
	L.zbuf[tid] = val;
	__syncthreads();
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		constexpr int FF = F;
		const int p = plan_p(FF, s), m = plan_m(FF, s), fs = plan_fs(FF, s);
		if (tid < F / p) {
			const int i = tid / m, j = tid - i * m;
			float2 *Fo = L.zbuf + i * (p * m) + j;
			if (p == 2) {
				float2 w = L.tw[j * fs];
				if (inverse) w.y = -w.y;
				const float2 t = cmulf(Fo[m], w);
				const float2 a = Fo[0];
				Fo[m] = make_float2(a.x - t.x, a.y - t.y);
				Fo[0] = make_float2(a.x + t.x, a.y + t.y);
			} else {
				float2 w1 = L.tw[j * fs], w2 = L.tw[j * fs * 2], w3 = L.tw[j * fs * 3];
				if (inverse) {
					w1.y = -w1.y;
					w2.y = -w2.y;
					w3.y = -w3.y;
				}
				const float2 s0 = cmulf(Fo[m], w1);
				const float2 s1 = cmulf(Fo[2 * m], w2);
				const float2 s2 = cmulf(Fo[3 * m], w3);
				float2 f0 = Fo[0];
				const float2 s5 = make_float2(f0.x - s1.x, f0.y - s1.y);
				f0.x += s1.x;
				f0.y += s1.y;
				const float2 s3 = make_float2(s0.x + s2.x, s0.y + s2.y);
				const float2 s4 = make_float2(s0.x - s2.x, s0.y - s2.y);
				Fo[2 * m] = make_float2(f0.x - s3.x, f0.y - s3.y);
				f0.x += s3.x;
				f0.y += s3.y;
				Fo[0] = f0;
				if (inverse) {
					Fo[m] = make_float2(s5.x - s4.y, s5.y + s4.x);
					Fo[3 * m] = make_float2(s5.x + s4.y, s5.y - s4.x);
				} else {
					Fo[m] = make_float2(s5.x + s4.y, s5.y - s4.x);
					Fo[3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
				}
			}
		}
		__syncthreads();
	}
}

// time L.tbuf[2F] -> this lane's bin (scaled by 1/N like spx_fft / ms_fft).
// bin 0 returns (DC, Nyquist).  Also leaves the interleaved spectrum in L.spec.
template <int F>
__device__ float2 rfft_forward(Lds<F> &L) {
	const int tid = threadIdx.x;
	cfft<F>(L, reinterpret_cast<const float2 *>(L.tbuf), false);
	const float scale = 1.f / (2 * F);
	if (tid == 0) {
		const float2 t0 = L.zbuf[0];
		L.spec[0] = (t0.x + t0.y) * scale;
		L.spec[1] = (t0.x - t0.y) * scale;
	} else if (tid <= F / 2) {
		const int k = tid;
		const float2 a = L.zbuf[k], b = L.zbuf[F - k];
		const float2 sw = L.super[k];
		const float f2r = a.x - b.x, f2i = a.y + b.y;
		const float f1r = a.x + b.x, f1i = a.y - b.y;
		const float twr = f2r * sw.x - f2i * sw.y;
		const float twi = f2i * sw.x + f2r * sw.y;
		if (k != F - k) {
			L.spec[2 * k] = (.5f * (f1r + twr)) * scale;
			L.spec[2 * k + 1] = (.5f * (f1i + twi)) * scale;
		}
		L.spec[2 * (F - k)] = (.5f * (f1r - twr)) * scale;
		L.spec[2 * (F - k) + 1] = (.5f * (twi - f1i)) * scale;
	}
	__syncthreads();
	const float2 r = make_float2(L.spec[2 * tid], L.spec[2 * tid + 1]);
	return r;
}

// L.spec (interleaved spectrum, every lane has written its bin) -> time in L.tbuf, unscaled.
template <int F>
__device__ void rfft_inverse(Lds<F> &L) {
	const int tid = threadIdx.x;
	float2 *tmp = reinterpret_cast<float2 *>(L.tbuf); // staging for the pre-processed bins
	__syncthreads();
	if (tid == 0) {
		tmp[0] = make_float2(L.spec[0] + L.spec[1], L.spec[0] - L.spec[1]);
	} else if (tid <= F / 2) {
		const int k = tid;
		const float2 fk = make_float2(L.spec[2 * k], L.spec[2 * k + 1]);
		const float2 fnkc = make_float2(L.spec[2 * (F - k)], -L.spec[2 * (F - k) + 1]);
		float2 sw = L.super[k];
		sw.y = -sw.y; // inverse super-twiddle = conjugate
		const float2 fek = make_float2(fk.x + fnkc.x, fk.y + fnkc.y);
		const float2 d = make_float2(fk.x - fnkc.x, fk.y - fnkc.y);
		const float2 fok = cmulf(d, sw);
		if (k != F - k) tmp[k] = make_float2(fek.x + fok.x, fek.y + fok.y);
		float2 c = make_float2(fek.x - fok.x, fek.y - fok.y);
		c.y *= -1;
		tmp[F - k] = c;
	}
	__syncthreads();
	cfft<F>(L, tmp, true);
	// zbuf[n] = (t[2n], t[2n+1])
	const float2 r = L.zbuf[tid];
	L.tbuf[2 * tid] = r.x;
	L.tbuf[2 * tid + 1] = r.y;
	__syncthreads();
}

__device__ __forceinline__ float rdlane(float v, int l) {
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Serial-order reductions executed by ONE wave: every lane owns K = F/64 consecutive
// elements in registers, the products are formed lane-parallel (same values the
// library's loop forms), and only the running sum walks the lanes in order via
// v_readlane, so the dependent chain never waits on LDS.
template <int F>
struct WaveSeq {
	static constexpr int K = F / 64;
	// mdf_inner_prod: sum += (x0*y0 + x1*y1) over consecutive pairs
	__device__ static float inner_prod(const float *x, const float *y, int lane) {
		float part[K / 2];
#pragma unroll
		for (int k = 0; k < K; k += 2) {
			float p = 0;
			p = p + x[lane * K + k] * y[lane * K + k];
			p = p + x[lane * K + k + 1] * y[lane * K + k + 1];
			part[k / 2] = p;
		}
		float sum = 0;
#pragma unroll
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int k = 0; k < K / 2; ++k) sum = sum + rdlane(part[k], l);
		}
		return sum;
	}
	// acc = init; for j = F-1 .. 0: acc = acc + a[j]*b[j]   (descending)
	__device__ static float dot_desc(float init, const float *a, const float *b, int lane) {
		float p[K];
#pragma unroll
		for (int k = 0; k < K; ++k) p[k] = a[lane * K + k] * b[lane * K + k];
		float acc = init;
#pragma unroll
		for (int l = 63; l >= 0; --l) {
#pragma unroll
			for (int k = K - 1; k >= 0; --k) acc = acc + rdlane(p[k], l);
		}
		return acc;
	}
};

__device__ __forceinline__ int16_t word2int(float x) {
	if (x < -32767.5f) return (int16_t)-32768;
	if (x > 32766.5f) return (int16_t)32767;
	return (int16_t)(int)floor(.5 + (double)x);
}

__device__ __forceinline__ float qcurve(float x) { return 1.f / (1.f + .15f / x); }

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

template <int F>
__global__ __launch_bounds__(F) void aec_kernel(AecArgs a) {
	__shared__ Lds<F> L;
	using SL = SmallLayout<F>;
	constexpr int N = 2 * F;
	const int s = blockIdx.x;
	if (a.run && !a.run[s]) return;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int M = a.M;
	float *sm = a.small + (size_t)s * a.small_stride;
	float2 *Xs = reinterpret_cast<float2 *>(a.X + (size_t)s * (M + 1) * N);
	float2 *Ws = reinterpret_cast<float2 *>(a.W + (size_t)s * M * N);
	float2 *FGs = reinterpret_cast<float2 *>(a.FG + (size_t)s * M * N);
	AecScalars sc = a.scal[s];

	// ---------------------------------------------------------------- tables + inputs
	L.tw[tid] = a.t.tw[tid];
	L.super[tid] = a.t.super[tid];
	L.perm[tid] = a.t.perm[tid];
	if (tid < M) L.prop[tid] = sm[SL::PROP + tid];
	const int16_t mic_i = a.mic[(size_t)s * a.stride + tid];
	const int16_t far_i = a.ref[(size_t)s * a.stride + tid];
	const float far_prev = (tid == 0) ? sc.memX : (float)a.ref[(size_t)s * a.stride + tid - 1];
	L.fin[tid] = (float)mic_i;
	L.tbuf[tid] = sm[SL::XPREV + tid];
	const float xnew = (float)far_i - .9f * far_prev;
	L.tbuf[F + tid] = xnew;
	sm[SL::XPREV + tid] = xnew;
	const int any_sat = __syncthreads_or(mic_i <= -32000 || mic_i >= 32000);
	sc.cancel_count++;

	// ---------------------------------------------------------------- serial: DC notch (wave 0), Sxx (wave 1 or lane 1)
	float Sxx = 0;
	constexpr int K = F / 64;
	if (wave == 0) {
		// filter_dc_notch16: a 2-state IIR, inherently serial; samples come out of registers
		const float radius = a.notch_radius;
		const float den2 = (float)(radius * radius + .7 * (1 - radius) * (1 - radius));
		float m0 = sc.notch0, m1 = sc.notch1;
		float xin[K], yo[K];
#pragma unroll
		for (int k = 0; k < K; ++k) xin[k] = L.fin[lane * K + k];
#pragma unroll
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const float vin = rdlane(xin[k], l);
				const float vout = m0 + vin;
				m0 = m1 + 2 * (-vin + radius * vout);
				m1 = vin - den2 * vout;
				const float y = radius * vout;
				if (lane == l) yo[k] = y;
			}
		}
#pragma unroll
		for (int k = 0; k < K; ++k) L.v[lane * K + k] = yo[k];
		if (lane == 0) {
			L.red[0] = m0;
			L.red[1] = m1;
		}
	} else if (wave == 1) {
		const float r = WaveSeq<F>::inner_prod(L.tbuf + F, L.tbuf + F, lane);
		if (lane == 0) L.red[2] = r;
	}
	__syncthreads();
	sc.notch0 = L.red[0];
	sc.notch1 = L.red[1];
	Sxx = L.red[2];
	{
		const float vprev = (tid == 0) ? sc.memD : L.v[tid - 1];
		L.input[tid] = L.v[tid] - .9f * vprev;
	}
	sc.memD = L.v[F - 1];
	sc.memX = (float)a.ref[(size_t)s * a.stride + F - 1];

	// ---------------------------------------------------------------- X0 = FFT(x), into the ring
	const float2 X0 = rfft_forward<F>(L);
	const int head = (sc.xhead + M) % (M + 1); // the slot of the oldest block becomes the newest
	sc.xhead = head;
	Xs[(size_t)head * F + tid] = X0;
	auto xslot = [&](int j) { return (size_t)((head + j) % (M + 1)) * F + tid; };

	// per-bin state
	const float2 Eprev = reinterpret_cast<const float2 *>(sm + SL::E)[tid];
	const float p1_k = sm[SL::POWER1 + tid];
	const float p1_F = sm[SL::POWER1 + F]; // Nyquist weight (used by lane 0)

	// ---------------------------------------------------------------- proportional step (mdf_adjust_prop)
	if (sc.adapted) {
		if (tid == 0) {
			float max_sum = 1, prop_sum = 1;
			for (int i = 0; i < M; ++i) {
				const float p = (float)sqrt((double)(1.0f + sm[SL::WNORM + i]));
				L.prop[i] = p;
				if (p > max_sum) max_sum = p;
			}
			for (int i = 0; i < M; ++i) {
				L.prop[i] += .1f * max_sum;
				prop_sum += L.prop[i];
			}
			for (int i = 0; i < M; ++i) L.prop[i] = (.99f * L.prop[i]) / prop_sum;
		}
		__syncthreads();
		if (tid < M) sm[SL::PROP + tid] = L.prop[tid];
	}
	const bool do_grad = (sc.saturated == 0);
	if (!do_grad) sc.saturated--;

	// gradient of one bin: W += p*w * conj-product(X, E)   (weighted_spectral_mul_conj)
	auto grad = [&](float2 w, float2 x, float prop) -> float2 {
		if (tid == 0) { // DC and Nyquist are real
			const float W0 = prop * p1_k, WN = prop * p1_F;
			w.x += W0 * (x.x * Eprev.x);
			w.y += WN * (x.y * Eprev.y);
		} else {
			const float Wt = prop * p1_k;
			w.x += Wt * ((x.x * Eprev.x) + x.y * Eprev.y);
			w.y += Wt * (((-x.y) * Eprev.x) + x.x * Eprev.y);
		}
		return w;
	};

	// ---------------------------------------------------------------- AUMDF blocks first: j = 0 and the round-robin one
	const int jc = (M > 1) ? (sc.cancel_count % (M - 1)) + 1 : -1;
	float2 wsp0 = make_float2(0, 0), wspc = make_float2(0, 0);
	for (int pass = 0; pass < 2; ++pass) {
		const int jb = pass == 0 ? 0 : jc;
		if (jb < 0) break;
		float2 w = Ws[(size_t)jb * F + tid];
		if (do_grad) w = grad(w, Xs[xslot(jb + 1)], L.prop[jb]);
		L.spec[2 * tid] = w.x;
		L.spec[2 * tid + 1] = w.y;
		rfft_inverse<F>(L);
		L.tbuf[F + tid] = 0.f;
		__syncthreads();
		w = rfft_forward<F>(L);
		Ws[(size_t)jb * F + tid] = w;
		if (pass == 0) wsp0 = w;
		else wspc = w;
	}

	// ---------------------------------------------------------------- the one streaming pass over X, FG, W
	float2 yfg = make_float2(0, 0), ybgs = make_float2(0, 0);
	{
		// software-pipelined: block j+1's three loads are in flight while block j is consumed
		float2 xj = X0;
		float2 xn = Xs[xslot(1)];
		float2 fg = FGs[tid];
		float2 wl = Ws[tid];
		float nn_acc[64 / 4]; // weight norms, 4 blocks per register group (reduced after the loop)
		for (int j = 0; j < M; ++j) {
			float2 xn2 = xn, fg2 = fg, wl2 = wl;
			if (j + 1 < M) {
				xn2 = Xs[xslot(j + 2)];
				fg2 = FGs[(size_t)(j + 1) * F + tid];
				wl2 = Ws[(size_t)(j + 1) * F + tid];
			}
			float2 w;
			if (j == 0) w = wsp0;
			else if (j == jc) w = wspc;
			else {
				w = wl;
				if (do_grad) {
					w = grad(w, xn, L.prop[j]);
					Ws[(size_t)j * F + tid] = w;
				}
			}
			if (tid == 0) {
				yfg.x += xj.x * fg.x;
				yfg.y += xj.y * fg.y;
				ybgs.x += xj.x * w.x;
				ybgs.y += xj.y * w.y;
			} else {
				yfg.x += (xj.x * fg.x - xj.y * fg.y);
				yfg.y += (xj.y * fg.x + xj.x * fg.y);
				ybgs.x += (xj.x * w.x - xj.y * w.y);
				ybgs.y += (xj.y * w.x + xj.x * w.y);
			}
			// per-block weight norm (tree reduction; only feeds the proportional step)
			if (!(a.flags & 0x100)) {
				float nn = w.x * w.x + w.y * w.y;
				for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o);
				if (lane == 0) L.wn[j * 4 + wave] = nn;
			}
			xj = xn;
			xn = xn2;
			fg = fg2;
			wl = wl2;
		}
		(void)nn_acc;
	}

	// ---------------------------------------------------------------- time-domain responses
	L.spec[2 * tid] = yfg.x;
	L.spec[2 * tid + 1] = yfg.y;
	rfft_inverse<F>(L);
	L.efg[tid] = L.tbuf[F + tid];
	L.e1[tid] = L.input[tid] - L.tbuf[F + tid]; // foreground error
	__syncthreads();
	L.spec[2 * tid] = ybgs.x;
	L.spec[2 * tid + 1] = ybgs.y;
	rfft_inverse<F>(L);
	L.ybg[tid] = L.tbuf[F + tid];
	L.e2[tid] = L.input[tid] - L.tbuf[F + tid]; // background error
	L.v[tid] = L.efg[tid] - L.tbuf[F + tid];    // difference of the two responses
	__syncthreads();
	if (wave < 3 && wave < F / 64) {
		const float *p = wave == 0 ? L.e1 : (wave == 1 ? L.v : L.e2);
		const float r = WaveSeq<F>::inner_prod(p, p, lane);
		if (lane == 0) L.red[wave] = r;
	}
	if (F / 64 < 3 && wave == 0) { // 2-wave workgroups: wave 0 takes the third sum
		const float r = WaveSeq<F>::inner_prod(L.e2, L.e2, lane);
		if (lane == 0) L.red[2] = r;
	}
	__syncthreads();
	const float Sff = L.red[0];
	const float Dbf = 10 + L.red[1];
	float See = L.red[2];
	__syncthreads();

	// ---------------------------------------------------------------- two-path control (uniform)
	sc.Davg1 = .6f * sc.Davg1 + .4f * (Sff - See);
	sc.Davg2 = .85f * sc.Davg2 + .15f * (Sff - See);
	sc.Dvar1 = .36f * sc.Dvar1 + (.4f * Sff) * (.4f * Dbf);
	sc.Dvar2 = .7225f * sc.Dvar2 + (.15f * Sff) * (.15f * Dbf);
	bool update_foreground = false;
	if ((Sff - See) * fabsf(Sff - See) > Sff * Dbf) update_foreground = true;
	else if (sc.Davg1 * fabsf(sc.Davg1) > .5f * sc.Dvar1) update_foreground = true;
	else if (sc.Davg2 * fabsf(sc.Davg2) > .25f * sc.Dvar2) update_foreground = true;
	bool wnorm_from_fg = false;
	if (update_foreground) {
		sc.Davg1 = sc.Davg2 = 0;
		sc.Dvar1 = sc.Dvar2 = 0;
		for (int j = 0; j < M; ++j) FGs[(size_t)j * F + tid] = Ws[(size_t)j * F + tid];
		L.efg[tid] = a.t.hann[tid + F] * L.efg[tid] + a.t.hann[tid] * L.ybg[tid];
	} else {
		bool reset_background = false;
		if ((-(Sff - See)) * fabsf(Sff - See) > 4.f * (Sff * Dbf)) reset_background = true;
		if ((-sc.Davg1) * fabsf(sc.Davg1) > 4.f * sc.Dvar1) reset_background = true;
		if ((-sc.Davg2) * fabsf(sc.Davg2) > 4.f * sc.Dvar2) reset_background = true;
		if (reset_background) {
			for (int j = 0; j < M; ++j) {
				const float2 w = FGs[(size_t)j * F + tid];
				Ws[(size_t)j * F + tid] = w;
				float nn = w.x * w.x + w.y * w.y;
				for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o);
				if (lane == 0) L.wn[j * 4 + wave] = nn;
			}
			wnorm_from_fg = true;
			L.ybg[tid] = L.efg[tid];
			L.e2[tid] = L.input[tid] - L.efg[tid];
			See = Sff;
			sc.Davg1 = sc.Davg2 = 0;
			sc.Dvar1 = sc.Dvar2 = 0;
		}
	}
	(void)wnorm_from_fg;
	__syncthreads();
	if (tid < M) {
		float t = 0;
		for (int w = 0; w < F / 64; ++w) t += L.wn[tid * 4 + w];
		sm[SL::WNORM + tid] = t;
	}

	// ---------------------------------------------------------------- output (serial de-emphasis) + correlations
	L.v[tid] = L.input[tid] - L.efg[tid];
	__syncthreads();
	if (wave == 0) {
		float memE = sc.memE;
		float xin[K], yo[K];
#pragma unroll
		for (int k = 0; k < K; ++k) xin[k] = L.v[lane * K + k];
#pragma unroll
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int k = 0; k < K; ++k) {
				float t = rdlane(xin[k], l);
				t = t + .9f * memE;
				memE = t;
				if (lane == l) yo[k] = t;
			}
		}
#pragma unroll
		for (int k = 0; k < K; ++k) L.tbuf[lane * K + k] = yo[k]; // tmp_out
		if (lane == 0) L.red[3] = memE;
		if (F / 64 < 3) { // 2-wave workgroups: Sdd here
			const float r = WaveSeq<F>::inner_prod(L.input, L.input, lane);
			if (lane == 0) L.red[2] = r;
		}
	} else if (wave == 1) {
		const float r = WaveSeq<F>::inner_prod(L.e2, L.ybg, lane);
		if (lane == 0) L.red[0] = r;
		if (F / 64 < 3) {
			const float r2 = WaveSeq<F>::inner_prod(L.ybg, L.ybg, lane);
			if (lane == 0) L.red[1] = r2;
		}
	} else if (wave == 2) {
		const float r = WaveSeq<F>::inner_prod(L.ybg, L.ybg, lane);
		if (lane == 0) L.red[1] = r;
	} else if (wave == 3) {
		const float r = WaveSeq<F>::inner_prod(L.input, L.input, lane);
		if (lane == 0) L.red[2] = r;
	}
	__syncthreads();
	sc.memE = L.red[3];
	const float Sey = L.red[0], Syy = L.red[1], Sdd = L.red[2];
	if (any_sat && sc.saturated == 0) sc.saturated = 1;
	int16_t out_i = word2int(L.tbuf[tid]);
	__syncthreads();

	// ---------------------------------------------------------------- error / response spectra
	L.tbuf[tid] = 0.f;
	L.tbuf[F + tid] = L.e2[tid];
	__syncthreads();
	const float2 Ecur = rfft_forward<F>(L);
	__syncthreads();
	L.tbuf[tid] = 0.f;
	L.tbuf[F + tid] = L.ybg[tid];
	__syncthreads();
	const float2 Ycur = rfft_forward<F>(L);
	reinterpret_cast<float2 *>(sm + SL::E)[tid] = Ecur;
	float Rf_k, Yf_k, Xf_k, Rf_F = 0, Yf_F = 0, Xf_F = 0;
	if (tid == 0) {
		Rf_k = Ecur.x * Ecur.x;
		Rf_F = Ecur.y * Ecur.y;
		Yf_k = Ycur.x * Ycur.x;
		Yf_F = Ycur.y * Ycur.y;
		Xf_k = X0.x * X0.x;
		Xf_F = X0.y * X0.y;
	} else {
		Rf_k = Ecur.x * Ecur.x + Ecur.y * Ecur.y;
		Yf_k = Ycur.x * Ycur.x + Ycur.y * Ycur.y;
		Xf_k = X0.x * X0.x + X0.y * X0.y;
	}

	// ---------------------------------------------------------------- sanity checks
	bool zero_out = false;
	if (!(Syy >= 0 && Sxx >= 0 && See >= 0) || !(Sff < N * 1e9 && Syy < N * 1e9 && Sxx < N * 1e9)) {
		sc.screwed_up += 50;
		zero_out = true;
	} else if (Sff > Sdd + (float)(N * 10000)) {
		sc.screwed_up++;
	} else {
		sc.screwed_up = 0;
	}
	if (zero_out) out_i = 0;
	if (sc.screwed_up >= 50) {
		// speex_echo_state_reset: everything back to the initial state
		for (int j = 0; j < M; ++j) {
			Ws[(size_t)j * F + tid] = make_float2(0, 0);
			FGs[(size_t)j * F + tid] = make_float2(0, 0);
		}
		for (int j = 0; j <= M; ++j) Xs[(size_t)j * F + tid] = make_float2(0, 0);
		sm[SL::POWER + tid] = 0;
		sm[SL::POWER1 + tid] = 1.0f;
		sm[SL::EH + tid] = 0;
		sm[SL::YH + tid] = 0;
		if (tid == 0) {
			sm[SL::POWER + F] = 0;
			sm[SL::POWER1 + F] = 1.0f;
			sm[SL::EH + F] = 0;
			sm[SL::YH + F] = 0;
		}
		sm[SL::LASTY + tid] = 0;
		sm[SL::LASTY + F + tid] = 0;
		reinterpret_cast<float2 *>(sm + SL::E)[tid] = make_float2(0, 0);
		sm[SL::XPREV + tid] = 0;
		if (tid < M) sm[SL::WNORM + tid] = 0;
		AecScalars z = sc;
		z.cancel_count = 0;
		z.screwed_up = 0;
		z.notch0 = z.notch1 = 0;
		z.memD = z.memE = z.memX = 0;
		z.saturated = 0;
		z.adapted = 0;
		z.sum_adapt = 0;
		z.Pey = z.Pyy = 1.0f;
		z.Davg1 = z.Davg2 = z.Dvar1 = z.Dvar2 = 0;
		if (tid == 0) a.scal[s] = z;
		a.out[(size_t)s * a.stride + tid] = out_i;
		return;
	}
	if (See < (float)(N * 100)) See = (float)(N * 100);
	Sxx += Sxx; // sic: the library accumulates the far-end energy a second time here

	// ---------------------------------------------------------------- far-end power, leak estimate
	float pw_k = sm[SL::POWER + tid];
	pw_k = a.ss_1 * pw_k + 1 + a.ss * Xf_k;
	sm[SL::POWER + tid] = pw_k;
	float pw_F = 0;
	if (tid == 0) {
		pw_F = sm[SL::POWER + F];
		pw_F = a.ss_1 * pw_F + 1 + a.ss * Xf_F;
		sm[SL::POWER + F] = pw_F;
	}
	{
		float eh = sm[SL::EH + tid], yh = sm[SL::YH + tid];
		L.e1[tid] = Rf_k - eh; // Eh differences, index k
		L.e2[tid] = Yf_k - yh;
		sm[SL::EH + tid] = (1 - a.spec_average) * eh + a.spec_average * Rf_k;
		sm[SL::YH + tid] = (1 - a.spec_average) * yh + a.spec_average * Yf_k;
		if (tid == 0) {
			eh = sm[SL::EH + F];
			yh = sm[SL::YH + F];
			L.red[8] = Rf_F - eh;
			L.red[9] = Yf_F - yh;
			sm[SL::EH + F] = (1 - a.spec_average) * eh + a.spec_average * Rf_F;
			sm[SL::YH + F] = (1 - a.spec_average) * yh + a.spec_average * Yf_F;
		}
	}
	__syncthreads();
	if (wave < 2) {
		// j = F down to 0, starting from FLOAT_ONE
		const float eF = L.red[8], yF = L.red[9];
		float acc = 1.0f;
		acc = acc + (wave == 0 ? eF * yF : yF * yF);
		acc = WaveSeq<F>::dot_desc(acc, wave == 0 ? L.e1 : L.e2, L.e2, lane);
		if (lane == 0) L.red[wave] = acc;
	}
	__syncthreads();
	float Pey = L.red[0], Pyy = L.red[1];
	Pyy = (float)sqrt((double)Pyy);
	Pey = Pey / Pyy;
	float tmp32 = a.beta0 * Syy;
	if (tmp32 > a.beta_max * See) tmp32 = a.beta_max * See;
	const float alpha = tmp32 / See;
	const float alpha_1 = 1.0f - alpha;
	sc.Pey = alpha_1 * sc.Pey + alpha * Pey;
	sc.Pyy = alpha_1 * sc.Pyy + alpha * Pyy;
	if (sc.Pyy < 1.0f) sc.Pyy = 1.0f;
	if (sc.Pey < .005f * sc.Pyy) sc.Pey = .005f * sc.Pyy;
	if (sc.Pey > sc.Pyy) sc.Pey = sc.Pyy;
	sc.leak_estimate = sc.Pey / sc.Pyy;
	float RER = (float)((.0001 * Sxx + 3. * (sc.leak_estimate * Syy)) / See);
	if (RER < Sey * Sey / (1 + See * Syy)) RER = Sey * Sey / (1 + See * Syy);
	if (RER > .5) RER = .5;
	if (!sc.adapted && sc.sum_adapt > (float)M && sc.leak_estimate * Syy > .03f * Syy) sc.adapted = 1;

	auto step = [&](float Yf, float Rf, float pw) -> float {
		float r = sc.leak_estimate * Yf;
		const float e = Rf + 1;
		if (r > .5 * e) r = (float)(.5 * e);
		r = .7f * r + .3f * (float)(RER * e);
		return r / (e * (pw + 10));
	};
	if (sc.adapted) {
		sm[SL::POWER1 + tid] = step(Yf_k, Rf_k, pw_k);
		if (tid == 0) sm[SL::POWER1 + F] = step(Yf_F, Rf_F, pw_F);
	} else {
		float adapt_rate = 0;
		if (Sxx > (float)(N * 1000)) {
			tmp32 = .25f * Sxx;
			if (tmp32 > .25 * See) tmp32 = (float)(.25 * See);
			adapt_rate = tmp32 / See;
		}
		sm[SL::POWER1 + tid] = adapt_rate / (pw_k + 10);
		if (tid == 0) sm[SL::POWER1 + F] = adapt_rate / (pw_F + 10);
		sc.sum_adapt = sc.sum_adapt + adapt_rate;
	}

	// last_y: the echo estimate used by the residual-echo spectrum
	const float ly_old = sm[SL::LASTY + F + tid];
	float ly_new = ly_old;
	if (sc.adapted) ly_new = (float)((int)mic_i - (int)out_i);
	sm[SL::LASTY + tid] = ly_old;
	sm[SL::LASTY + F + tid] = ly_new;

	// ================================================================ post-filter (speex_preprocess_run)
	if (a.flags & MI_AEC_POSTFILTER) {
		sc.nb_adapt++;
		if (sc.nb_adapt > 20000) sc.nb_adapt = 20000;
		sc.min_count++;
		float beta = 1.0f / sc.nb_adapt;
		if (beta < .03f) beta = .03f;
		const float beta_1 = 1.0f - beta;

		// residual echo spectrum (speex_echo_get_residual)
		__syncthreads();
		L.tbuf[tid] = a.t.hann[tid] * ly_old;
		L.tbuf[F + tid] = a.t.hann[F + tid] * ly_new;
		__syncthreads();
		const float2 Yr = rfft_forward<F>(L);
		float res = (tid == 0) ? Yr.x * Yr.x : Yr.x * Yr.x + Yr.y * Yr.y;
		const float leak2 = (sc.leak_estimate > .5) ? 1.f : 2 * sc.leak_estimate;
		res = (float)(int32_t)(leak2 * res);
		const int bad = __syncthreads_or(tid == 0 && !(res >= 0 && res < F * 1e9f));
		if (bad) res = 0;
		float en = sm[SL::ECHON + tid];
		{
			const float c = .6f * en;
			en = c > res ? c : res;
		}
		sm[SL::ECHON + tid] = en;
		float *vec = L.e1; // per-bin exchange vector
		float *pl = L.spec, *pr = L.spec + F; // per-bin filterbank products
		const float wl = a.t.bfl[tid], wr = a.t.bfr[tid];
		pl[tid] = wl * en;
		pr[tid] = wr * en;
		// analysis frame: [inbuf, x] * window
		const float inb = sm[SL::INBUF + tid];
		const float xcur = (float)out_i;
		sm[SL::INBUF + tid] = xcur;
		L.tbuf[tid] = inb * a.t.pwin[tid];
		L.tbuf[F + tid] = xcur * a.t.pwin[F + tid];
		__syncthreads();
		float *bandv = L.band; // [0..24) echo_noise bands, [24..48) ps bands, [48..72) noise bands, [72..96) misc
		if (tid < NB_BANDS) bandv[tid] = band_sum<F>(a.t, tid, pl, pr);
		__syncthreads();
		float2 ft = rfft_forward<F>(L);
		const float ps = (tid == 0) ? ft.x * ft.x : ft.x * ft.x + ft.y * ft.y;
		__syncthreads();
		vec[tid] = ps;
		pl[tid] = wl * ps;
		pr[tid] = wr * ps;
		__syncthreads();
		if (tid < NB_BANDS) bandv[NB_BANDS + tid] = band_sum<F>(a.t, tid, pl, pr);
		// update_noise_prob
		float S = sm[SL::S_ + tid], Smin = sm[SL::SMIN + tid], Stmp = sm[SL::STMP + tid];
		if (tid == 0 || tid == F - 1) S = .8f * S + .2f * ps;
		else S = .8f * S + .05f * vec[tid - 1] + .1f * ps + .05f * vec[tid + 1];
		if (sc.nb_adapt == 1) Smin = Stmp = 0;
		int min_range;
		if (sc.nb_adapt < 100) min_range = 15;
		else if (sc.nb_adapt < 1000) min_range = 50;
		else if (sc.nb_adapt < 10000) min_range = 150;
		else min_range = 300;
		if (sc.min_count > min_range) {
			Smin = Stmp < S ? Stmp : S;
			Stmp = S;
		} else {
			Smin = Smin < S ? Smin : S;
			Stmp = Stmp < S ? Stmp : S;
		}
		const int update_prob = (.4f * S > Smin) ? 1 : 0;
		sm[SL::S_ + tid] = S;
		sm[SL::SMIN + tid] = Smin;
		sm[SL::STMP + tid] = Stmp;
		float noise = sm[SL::NOISE + tid];
		if (!update_prob || ps < noise) {
			const float v = beta_1 * noise + beta * ps;
			noise = v > 0 ? v : 0;
		}
		sm[SL::NOISE + tid] = noise;
		__syncthreads();
		pl[tid] = wl * noise;
		pr[tid] = wr * noise;
		__syncthreads();
		if (tid < NB_BANDS) bandv[2 * NB_BANDS + tid] = band_sum<F>(a.t, tid, pl, pr);
		__syncthreads();
		if (sc.min_count > min_range) sc.min_count = 0;

		// a posteriori / a priori SNR, bins and bands
		auto snr = [&](float psv, float noisev, float echov, float oldps, float &post, float &prior) {
			const float tot_noise = 1.f + noisev + echov + 0.f;
			post = psv / tot_noise - 1.f;
			if (post > 100.f) post = 100.f;
			const float t = oldps / (oldps + tot_noise);
			const float gamma = .1f + .89f * (t * t);
			prior = gamma * (post > 0 ? post : 0) + (1.0f - gamma) * (oldps / tot_noise);
			if (prior > 100.f) prior = 100.f;
		};
		float old_ps = sm[SL::OLDPS + tid];
		if (sc.nb_adapt == 1) old_ps = ps;
		float post_k, prior_k;
		snr(ps, noise, en, old_ps, post_k, prior_k);
		float old_ps_b = 0, post_b = 0, prior_b = 0, ps_b = 0;
		if (tid < NB_BANDS) {
			ps_b = bandv[NB_BANDS + tid];
			old_ps_b = sm[SL::OLDPS + F + tid];
			if (sc.nb_adapt == 1) old_ps_b = ps_b;
			snr(ps_b, bandv[2 * NB_BANDS + tid], bandv[tid], old_ps_b, post_b, prior_b);
		}
		__syncthreads();
		vec[tid] = prior_k;
		if (tid < NB_BANDS) bandv[3 * NB_BANDS + tid] = prior_b;
		__syncthreads();
		// zeta: recursive average of the a priori SNR
		float zeta = sm[SL::ZETA + tid];
		if (tid == 0) zeta = .7f * zeta + .3f * prior_k;
		else if (tid < F - 1) zeta = .7f * zeta + .15f * prior_k + .075f * vec[tid - 1] + .075f * vec[tid + 1];
		else zeta = .7f * zeta + .3f * prior_k;
		sm[SL::ZETA + tid] = zeta;
		float zeta_b = 0;
		if (tid < NB_BANDS) {
			zeta_b = .7f * sm[SL::ZETA + F + tid] + .3f * prior_b;
			sm[SL::ZETA + F + tid] = zeta_b;
		}
		__syncthreads();
		if (tid < NB_BANDS) bandv[3 * NB_BANDS + tid] = zeta_b;
		__syncthreads();
		float Zframe = 0;
		for (int i = 0; i < NB_BANDS; ++i) Zframe = Zframe + bandv[3 * NB_BANDS + i];
		const float Pframe = .1f + .899f * qcurve(Zframe / NB_BANDS);
		const int eff_echo = (int)((1.0f - Pframe) * -40 + Pframe * -15);
		__syncthreads();
		// band gains
		if (tid < NB_BANDS) {
			const float noise_floor = (float)exp((double)(.2302585f * -15));
			const float echo_floor = (float)exp((double)(.2302585f * eff_echo));
			const float nb = bandv[2 * NB_BANDS + tid], eb = bandv[tid];
			const float gfloor = (float)(sqrt((double)(noise_floor * nb + echo_floor * eb)) / sqrt((double)(1 + nb + eb)));
			const float prior_ratio = prior_b / (prior_b + 1.f);
			const float theta = prior_ratio * (1.f + post_b);
			const float MM = hypergeom_gain(theta);
			float g = prior_ratio * MM;
			if (g > 1.f) g = 1.f;
			old_ps_b = .2f * old_ps_b + (.8f * (g * g)) * ps_b;
			sm[SL::OLDPS + F + tid] = old_ps_b;
			const float P1 = .199f + .8f * qcurve(zeta_b);
			const float q = 1.0f - Pframe * P1;
			const float g2 = (float)(1 / (1.f + (q / (1.f - q)) * (1 + prior_b) * exp((double)(-theta))));
			bandv[tid] = g2;                  // gain2 bands
			bandv[NB_BANDS + tid] = g;        // gain bands
			bandv[2 * NB_BANDS + tid] = gfloor; // gain_floor bands
		}
		__syncthreads();
		// filterbank_compute_psd16: back to linear frequency
		const int bl = a.t.bleft[tid], br = bl + 1;
		auto psd = [&](const float *mel) -> float {
			float t = mel[bl] * wl;
			t += mel[br] * wr;
			return t;
		};
		const float p = psd(bandv);
		const float gain_bark = psd(bandv + NB_BANDS);
		const float gfl = psd(bandv + 2 * NB_BANDS);
		float gain2;
		{
			const float prior_ratio = prior_k / (prior_k + 1.f);
			const float theta = prior_ratio * (1.f + post_k);
			const float MM = hypergeom_gain(theta);
			float g = prior_ratio * MM;
			if (g > 1.f) g = 1.f;
			if (.333f * g > gain_bark) g = 3 * gain_bark;
			float gain = g;
			old_ps = .2f * old_ps + (.8f * (gain * gain)) * ps;
			if (gain < gfl) gain = gfl;
			const float tmp = p * (float)sqrt((double)gain) + (1.0f - p) * (float)sqrt((double)gfl);
			gain2 = tmp * tmp;
		}
		sm[SL::OLDPS + tid] = old_ps;
		// apply: bin k scales (re,im); DC uses gain2[0]; Nyquist uses gain2[F-1]
		__syncthreads();
		vec[tid] = gain2;
		__syncthreads();
		if (tid == 0) {
			ft.x = gain2 * ft.x;
			ft.y = vec[F - 1] * ft.y;
		} else {
			ft.x = gain2 * ft.x;
			ft.y = gain2 * ft.y;
		}
		L.spec[2 * tid] = ft.x;
		L.spec[2 * tid + 1] = ft.y;
		rfft_inverse<F>(L);
		const float f_lo = L.tbuf[tid] * a.t.pwin[tid];
		const float f_hi = L.tbuf[F + tid] * a.t.pwin[F + tid];
		const float ob = sm[SL::OUTBUF + tid];
		out_i = word2int(ob + f_lo);
		sm[SL::OUTBUF + tid] = f_hi;
	}

	a.out[(size_t)s * a.stride + tid] = out_i;
	if (tid == 0) a.scal[s] = sc;
}

// ---- debug: forward/inverse transform of one 2F-point frame per block (parity of the FFT itself)
template <int F>
__global__ __launch_bounds__(F) void fft_debug_kernel(const float *in, float *out, int inverse, AecTables t) {
	__shared__ Lds<F> L;
	const int tid = threadIdx.x;
	L.tw[tid] = t.tw[tid];
	L.super[tid] = t.super[tid];
	L.perm[tid] = t.perm[tid];
	const float *src = in + (size_t)blockIdx.x * 2 * F;
	float *dst = out + (size_t)blockIdx.x * 2 * F;
	if (!inverse) {
		L.tbuf[tid] = src[tid];
		L.tbuf[F + tid] = src[F + tid];
		__syncthreads();
		const float2 r = rfft_forward<F>(L);
		dst[2 * tid] = r.x;
		dst[2 * tid + 1] = r.y;
	} else {
		L.spec[2 * tid] = src[2 * tid];
		L.spec[2 * tid + 1] = src[2 * tid + 1];
		rfft_inverse<F>(L);
		dst[tid] = L.tbuf[tid];
		dst[F + tid] = L.tbuf[F + tid];
	}
}

} // namespace

struct mi_aec {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, rate = 0, F = 0, N = 0, M = 0;
	float *d_X = nullptr, *d_W = nullptr, *d_FG = nullptr, *d_small = nullptr;
	AecScalars *d_scal = nullptr;
	void *d_tables = nullptr;
	AecTables t;
	FftPlan plan;
	int small_stride = 0;
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
	if (a->plan.nstages != 4) return MI_ENOTSUP;
	for (int s = 0; s < 4; ++s)
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
	using SL = SmallLayout<F>;
	std::vector<float> small((size_t)a->small_stride, 0.f);
	for (int i = 0; i <= F; ++i) small[(size_t)SL::POWER1 + i] = 1.0f;
	for (int i = 0; i < a->M; ++i) small[(size_t)SL::PROP + i] = a->h_prop0[(size_t)i];
	for (int i = 0; i < F + NB_BANDS; ++i) {
		small[(size_t)SL::NOISE + i] = 1.f;
		small[(size_t)SL::OLDPS + i] = 1.f;
	}
	AecScalars sc;
	memset(&sc, 0, sizeof(sc));
	sc.Pey = sc.Pyy = 1.0f;
	std::vector<float> all((size_t)count * a->small_stride);
	std::vector<AecScalars> scs((size_t)count, sc);
	for (int i = 0; i < count; ++i) memcpy(all.data() + (size_t)i * a->small_stride, small.data(), small.size() * sizeof(float));
	MI_HIP(hipStreamSynchronize(a->ctx->stream));
	MI_HIP(hipMemcpy(a->d_small + (size_t)first * a->small_stride, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(a->d_scal + first, scs.data(), scs.size() * sizeof(AecScalars), hipMemcpyHostToDevice));
	const size_t wn = (size_t)a->M * a->N, xn = (size_t)(a->M + 1) * a->N;
	MI_HIP(hipMemset(a->d_W + first * wn, 0, (size_t)count * wn * sizeof(float)));
	MI_HIP(hipMemset(a->d_FG + first * wn, 0, (size_t)count * wn * sizeof(float)));
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

int mi_aec_create(mi_ctx *ctx, int nstreams, int sample_rate, int frame_size, int filter_length, mi_aec **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && sample_rate > 0 && filter_length > 0);
	*out = nullptr;
	if (frame_size != 128 && frame_size != 256) {
		mi::set_error("frame size %d not supported: the filter's 2^k sizing (speexec.c:171-180) gives 128 at 16 kHz "
		              "and 256 at 32-48 kHz; 64 (8 kHz) is not built",
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
	a->small_stride = frame_size == 256 ? SmallLayout<256>::TOTAL : SmallLayout<128>::TOTAL;
	const size_t wn = (size_t)M * a->N, xn = (size_t)(M + 1) * a->N;
	int rc = build_tables(a);
	if (rc != MI_OK) {
		mi_aec_destroy(a);
		return rc;
	}
	if (hipMalloc((void **)&a->d_X, (size_t)nstreams * xn * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&a->d_W, (size_t)nstreams * wn * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&a->d_FG, (size_t)nstreams * wn * sizeof(float)) != hipSuccess ||
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
	*out = a;
	return MI_OK;
}

void mi_aec_destroy(mi_aec *a) {
	if (!a) return;
	(void)hipSetDevice(a->ctx->device);
	if (a->d_X) (void)hipFree(a->d_X);
	if (a->d_W) (void)hipFree(a->d_W);
	if (a->d_FG) (void)hipFree(a->d_FG);
	if (a->d_small) (void)hipFree(a->d_small);
	if (a->d_scal) (void)hipFree(a->d_scal);
	if (a->d_tables) (void)hipFree(a->d_tables);
	delete a;
}

int mi_aec_reset(mi_aec *a, int first, int count) {
	MI_CHECK_ARG(a && first >= 0 && count >= 0 && first + count <= a->nstreams);
	if (count == 0) return MI_OK;
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	return a->F == 256 ? init_state<256>(a, first, count) : init_state<128>(a, first, count);
}

size_t mi_aec_state_bytes(const mi_aec *a) {
	if (!a) return 0;
	return ((size_t)(a->M + 1) * a->N + 2 * (size_t)a->M * a->N + (size_t)a->small_stride) * sizeof(float) +
	       sizeof(AecScalars);
}

int mi_aec_process(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride,
                   const uint8_t *d_run, unsigned flags) {
	MI_CHECK_ARG(a && d_mic && d_ref && d_out && stride >= a->F);
	if (a->ctx->activate() != MI_OK) return MI_ENODEV;
	AecArgs g;
	g.mic = d_mic;
	g.ref = d_ref;
	g.out = d_out;
	g.run = d_run;
	g.stride = stride;
	g.nstreams = a->nstreams;
	g.M = a->M;
	g.flags = (int)flags;
	g.X = a->d_X;
	g.W = a->d_W;
	g.FG = a->d_FG;
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
	if (a->F == 256) hipLaunchKernelGGL(aec_kernel<256>, dim3(a->nstreams), dim3(256), 0, a->ctx->stream, g);
	else hipLaunchKernelGGL(aec_kernel<128>, dim3(a->nstreams), dim3(128), 0, a->ctx->stream, g);
	MI_LAUNCH_CHECK();
	return MI_OK;
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
	const int o_e = F, o_pw = 3 * F, o_p1 = 5 * F, o_eh = 7 * F, o_yh = 9 * F, o_ly = 11 * F, o_prop = 13 * F;
	std::vector<float> res;
	if (!strcmp(what, "W") || !strcmp(what, "foreground")) {
		std::vector<float> raw((size_t)M * N);
		const float *src = (!strcmp(what, "W") ? a->d_W : a->d_FG) + (size_t)stream * M * N;
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
	} else if (!strcmp(what, "power")) res.assign(small.begin() + o_pw, small.begin() + o_pw + F + 1);
	else if (!strcmp(what, "power_1")) res.assign(small.begin() + o_p1, small.begin() + o_p1 + F + 1);
	else if (!strcmp(what, "Eh")) res.assign(small.begin() + o_eh, small.begin() + o_eh + F + 1);
	else if (!strcmp(what, "Yh")) res.assign(small.begin() + o_yh, small.begin() + o_yh + F + 1);
	else if (!strcmp(what, "last_y")) res.assign(small.begin() + o_ly, small.begin() + o_ly + N);
	else if (!strcmp(what, "prop")) res.assign(small.begin() + o_prop, small.begin() + o_prop + M);
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

// debug entry (not in the public header): raw transform parity
int mi_debug_fft(mi_aec *a, const float *d_in, float *d_out, int nframes, int inverse) {
	MI_CHECK_ARG(a && d_in && d_out && nframes > 0);
	if (a->F == 256)
		hipLaunchKernelGGL(fft_debug_kernel<256>, dim3(nframes), dim3(256), 0, a->ctx->stream, d_in, d_out, inverse, a->t);
	else
		hipLaunchKernelGGL(fft_debug_kernel<128>, dim3(nframes), dim3(128), 0, a->ctx->stream, d_in, d_out, inverse, a->t);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

} // extern "C"
