// plc.hip -- batched MSGenericPLC for gfx950: the FFT-based packet-loss concealer the reference inserts behind decoders
// without their own PLC (G.711, L16): src/audiofilters/genericplc.c:29-241 driven by generic_plc_process,
// src/audiofilters/msgenericplc.c:59-167.
//
// Per stream (plc_context_t, genericplc.h:46-58): the last nb = rate/20 samples heard (plc_buffer), 2 nb generated samples
// (plc_out_buffer), 2 x 5 ms of continuity buffer, two 16-bit counters.  Per tick every stream has one of two events:
//   RECEIVED  a block arrived: it is remembered, delayed by 5 ms through the continuity buffer, cross-faded with the
//             concealment signal if one was playing (msgenericplc.c:63-116);
//   CONCEAL   nothing arrived: n samples are taken from the generated signal, which is (re)built on the first loss and
//             whenever it runs out by generic_plc_fftbf (genericplc.c:83-121): Hann-like window, real FFT of nb points,
//             every stored bin moved to twice its index x 0.85, inverse real FFT of 2 nb points, truncation to int16;
//             faded to silence between 100 and 150 ms (:187-203).
//
// One wavefront per stream, everything it touches in LDS.  The transforms are kiss_fft's (src/utils/kiss_fft.c) evaluated
// in ITS operation order -- digit permutation, then radix 4 / 2 / 3 / 5 stages from the innermost factor outwards, each
// butterfly the reference's expression sequence, compiled with -ffp-contract=off -- so the generated samples are the
// reference's bit for bit (sizes: nb/2 = 200 .. 1200 and nb = 400 .. 2400 complex points: factors 4, 2, 3, 5, and 11 at
// 22.05 / 44.1 kHz through the generic butterfly).
// Butterflies of a stage are independent: lanes take them round-robin.  Twiddles and window come from the host (double
// cos/sin rounded to float, kiss_fft.c:464-471, kiss_fftr.c:68-81, genericplc.c:63-65).
// Concealment is rare (a stream only computes FFTs on the tick its packet went missing and every ~nb samples
// thereafter); a RECEIVED tick costs one pass over the block.
#include "common.hpp"

#include <cmath>

namespace {

constexpr int MAXFAC = 8;
constexpr int TRANSITION_DELAY = 5, PLC_DECREASE_START = 100, MAX_PLC_LEN = 150; // ms, genericplc.h:27,:34-35
constexpr float ENERGY_ATTENUATION = 0.85f;                                       // :40

struct Factors {
	int n, count;
	int p[MAXFAC], m[MAXFAC], fs[MAXFAC]; // radix, rest, twiddle stride (= product of the radices before)
};

struct PlcArgs {
	int16_t *cont, *hist, *gen; // [streams][2T], [streams][nb], [streams][2 nb]
	uint32_t *meta;             // [streams][4]: plc_index, plc_samples_used, history ring head, unused
	const float *window;
	const float2 *tw1, *sup1, *tw2, *sup2; // forward (nb/2 complex) and inverse (nb complex) twiddles + real-FFT super twiddles
	Factors f1, f2;
	int nb, T, rate, nstreams, cap;
	int light_floats; // LDS of one wavefront of the RECEIVED kernel, in floats
	int *count, *list; // streams to conceal this tick (filled by plc_received_kernel)
	int16_t *blocks;
	size_t stride;
	const int32_t *len;
	const uint8_t *mode;
};

__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { // C_MUL, _kiss_fft_guts.h:109-113
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ int16_t to_i16(float v) { return (int16_t)(int)v; } // (int16_t)float on x86-64: cvttss2si, low half

// kf_shuffle kiss_fft.c:292-318: element with digits (d0, d1, ..) over the radices lands at sum d_s * m_s
__device__ void shuffle(float2 *dst, const float2 *src, const Factors &f, int lane) {
	for (int i = lane; i < f.n; i += 64) {
		int rem = i, o = 0;
#pragma unroll 1
		for (int s = 0; s < f.count; ++s) {
			const int d = rem % f.p[s];
			rem /= f.p[s];
			o += d * f.m[s];
		}
		dst[o] = src[i];
	}
}

// kf_work kiss_fft.c:320-408, stages from the innermost outwards, on shuffled data in LDS
__device__ void cfft(float2 *buf, const Factors &f, const float2 *tw, bool inverse, int lane) {
#pragma unroll 1
	for (int s = f.count - 1; s >= 0; --s) {
		const int p = f.p[s], m = f.m[s], fs = f.fs[s], m2 = p * m;
		const int total = fs * m; // butterflies of the stage
		for (int b = lane; b < total; b += 64) {
			const int i = b / m, j = b - i * m;
			float2 *F = buf + i * m2 + j;
			if (p == 4) { // kf_bfly4 :61-117
				const float2 s0 = cmul(F[m], tw[j * fs]), s1 = cmul(F[2 * m], tw[2 * j * fs]), s2 = cmul(F[3 * m], tw[3 * j * fs]);
				const float2 s5 = csub(F[0], s1);
				const float2 f0 = cadd(F[0], s1);
				const float2 s3 = cadd(s0, s2), s4 = csub(s0, s2);
				F[2 * m] = csub(f0, s3);
				F[0] = cadd(f0, s3);
				if (inverse) {
					F[m] = make_float2(s5.x - s4.y, s5.y + s4.x);
					F[3 * m] = make_float2(s5.x + s4.y, s5.y - s4.x);
				} else {
					F[m] = make_float2(s5.x + s4.y, s5.y - s4.x);
					F[3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
				}
			} else if (p == 2) { // kf_bfly2 :38-59
				const float2 t = cmul(F[m], tw[j * fs]);
				F[m] = csub(F[0], t);
				F[0] = cadd(F[0], t);
			} else if (p == 3) { // kf_bfly3 :150-190
				const float e = tw[fs * m].y;
				const float2 s1 = cmul(F[m], tw[j * fs]), s2 = cmul(F[2 * m], tw[2 * j * fs]);
				const float2 s3 = cadd(s1, s2);
				float2 s0 = csub(s1, s2);
				float2 fm = make_float2(F[0].x - s3.x * .5f, F[0].y - s3.y * .5f);
				s0.x *= e;
				s0.y *= e;
				F[0] = cadd(F[0], s3);
				F[2 * m] = make_float2(fm.x + s0.y, fm.y - s0.x);
				fm.x -= s0.y;
				fm.y += s0.x;
				F[m] = fm;
			} else if (p == 5) { // kf_bfly5 :192-256
				const float2 ya = tw[fs * m], yb = tw[fs * 2 * m];
				const float2 s0 = F[0];
				const float2 s1 = cmul(F[m], tw[j * fs]), s2 = cmul(F[2 * m], tw[2 * j * fs]);
				const float2 s3 = cmul(F[3 * m], tw[3 * j * fs]), s4 = cmul(F[4 * m], tw[4 * j * fs]);
				const float2 s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
				F[0] = make_float2(s0.x + (s7.x + s8.x), s0.y + (s7.y + s8.y));
				const float2 s5 = make_float2(s0.x + s7.x * ya.x + s8.x * yb.x, s0.y + s7.y * ya.x + s8.y * yb.x);
				const float2 s6 = make_float2(s10.y * ya.y + s9.y * yb.y, -(s10.x * ya.y) - s9.x * yb.y);
				F[m] = csub(s5, s6);
				F[4 * m] = cadd(s5, s6);
				const float2 s11 = make_float2(s0.x + s7.x * yb.x + s8.x * ya.x, s0.y + s7.y * yb.x + s8.y * ya.x);
				const float2 s12 = make_float2(-(s10.y * yb.y) + s9.y * ya.y, s10.x * yb.y - s9.x * ya.y);
				F[2 * m] = cadd(s11, s12);
				F[3 * m] = csub(s11, s12);
			} else { // kf_bfly_generic :259-291 (radix 7, 11, 13, 17: the 44.1 kHz family has a factor 11); C_FIXDIV is a no-op
				float2 sc[17]; // in floats; the twiddle index walks the FULL transform's table modulo its size
				const int norig = f.n;
				for (int q = 0; q < p; ++q) sc[q] = F[q * m];
				int k = j;
				for (int q1 = 0; q1 < p; ++q1) {
					int twidx = 0;
					float2 acc = sc[0];
					for (int q = 1; q < p; ++q) {
						twidx += fs * k;
						if (twidx >= norig) twidx -= norig;
						acc = cadd(acc, cmul(sc[q], tw[twidx]));
					}
					F[q1 * m] = acc;
					k += m;
				}
			}
		}
		wave_sync();
	}
}

// generic_plc_transition_mix genericplc.c:233-241
__device__ __forceinline__ int16_t mix1(int16_t inout, int16_t continuity, int i, int n) {
	const float progress = __fdiv_rn((float)i, (float)n);
	return to_i16((float)continuity * (1 - progress) + (float)inout * progress);
}

struct Lds {
	float *A, *B;  // 2 nb floats each
	int16_t *gen;  // 2 nb
	int16_t *cont; // 2 T
	int16_t *blk, *out; // cap each
};

// generic_plc_fftbf genericplc.c:83-121.  src(i): the nb input samples; result in L.gen[0 .. 2 nb).
template <typename Src>
__device__ void fftbf(const PlcArgs &a, const Lds &L, Src src, int lane) {
	const int nb = a.nb, n1 = nb / 2, n2 = nb;
	for (int i = lane; i < nb; i += 64) L.A[i] = (float)src(i) * a.window[i];
	wave_sync();
	// ---- ms_fft: kiss_fftr2 (kiss_fftr.c:175-259, float branch) then the 1/N scale (dsptools.c:358-369)
	float2 *tmp = reinterpret_cast<float2 *>(L.B);
	shuffle(tmp, reinterpret_cast<const float2 *>(L.A), a.f1, lane);
	wave_sync();
	cfft(tmp, a.f1, a.tw1, false, lane);
	const float scale = __fdiv_rn(1.f, (float)nb);
	float *f = L.A;
	for (int k = lane; k <= n1 / 2; k += 64) {
		if (k == 0) {
			f[0] = (tmp[0].x + tmp[0].y) * scale;
			f[2 * n1 - 1] = (tmp[0].x - tmp[0].y) * scale;
			continue;
		}
		const float2 x = tmp[k], y = tmp[n1 - k], w = a.sup1[k];
		const float f2r = x.x - y.x, f2i = x.y + y.y, f1r = x.x + y.x, f1i = x.y - y.y;
		const float twr = f2r * w.x - f2i * w.y, twi = f2i * w.x + f2r * w.y;
		if (k != n1 - k) { // at k == n1 - k the second pair of stores lands on the same slots and wins
			f[2 * k - 1] = (.5f * (f1r + twr)) * scale;
			f[2 * k] = (.5f * (f1i + twi)) * scale;
		}
		f[2 * (n1 - k) - 1] = (.5f * (f1r - twr)) * scale;
		f[2 * (n1 - k)] = (.5f * (twi - f1i)) * scale;
	}
	wave_sync();
	// ---- the doubled spectrum (:98-101) fed to kiss_fftri2 (kiss_fftr.c:261-296): f2[2 i] = f[i] * 0.85, f2[2 i + 1] = 0
	auto f2 = [&](int j) -> float { return (j & 1) ? 0.f : f[j >> 1] * ENERGY_ATTENUATION; };
	for (int k = lane; k <= n2 / 2; k += 64) {
		if (k == 0) {
			tmp[0] = make_float2(f2(0) + f2(2 * n2 - 1), f2(0) - f2(2 * n2 - 1));
			continue;
		}
		const float2 fk = make_float2(f2(2 * k - 1), f2(2 * k));
		const float2 fnkc = make_float2(f2(2 * (n2 - k) - 1), -f2(2 * (n2 - k)));
		const float2 fek = cadd(fk, fnkc), fok = cmul(csub(fk, fnkc), a.sup2[k]);
		if (k != n2 - k) tmp[k] = cadd(fek, fok);
		float2 r = csub(fek, fok);
		r.y *= -1;
		tmp[n2 - k] = r;
	}
	wave_sync();
	float2 *t2 = reinterpret_cast<float2 *>(L.A);
	shuffle(t2, tmp, a.f2, lane);
	wave_sync();
	cfft(t2, a.f2, a.tw2, true, lane);
	for (int i = lane; i < 2 * nb; i += 64) L.gen[i] = to_i16(L.A[i]);
	wave_sync();
}

// generic_plc_update_plc_buffer genericplc.c:199-210 on a ring: logical sample i lives at (head + i) % nb
__device__ void update_history(int16_t *hist, uint32_t &head, const int16_t *data, int n, int nb, int lane) {
	if (n < nb) {
		for (int i = lane; i < n; i += 64) hist[(head + (uint32_t)i) % (uint32_t)nb] = data[i];
		head = (head + (uint32_t)n) % (uint32_t)nb;
	} else {
		for (int i = lane; i < nb; i += 64) hist[i] = data[n - nb + i];
		head = 0;
	}
}

// A tick is three launches.  plc_list_kernel compacts the ids of the streams to conceal (one atomic per 64 streams);
// plc_received_kernel serves the RECEIVED ones (a pass over the block: tiny LDS, four streams per workgroup);
// plc_conceal_kernel walks the list with a bounded grid, one stream per wavefront with the transform buffers in LDS.  (One kernel for both, sized for the transforms, kept
// 18 workgroups per CU for work that needs 1 KB: 48 us for 65 536 clean legs.)
constexpr int PLC_LIGHT_WAVES = 4;

template <bool CONCEAL>
__device__ __forceinline__ void plc_stream(const PlcArgs &a, float *lds_f, int s, int mode, int len, uint4 meta, int lane) {
	const int nb = a.nb, T = a.T;
	Lds L;
	if (CONCEAL) {
		L.A = lds_f;
		L.B = L.A + 2 * nb;
		L.gen = reinterpret_cast<int16_t *>(L.B + 2 * nb);
		L.cont = L.gen + 2 * nb;
	} else { // a received block needs no transform buffers
		L.A = L.B = nullptr;
		L.gen = nullptr;
		L.cont = reinterpret_cast<int16_t *>(lds_f);
	}
	L.blk = L.cont + 2 * T;
	L.out = L.blk + a.cap;
	int16_t *g_cont = a.cont + (size_t)s * 2 * T, *g_hist = a.hist + (size_t)s * nb, *g_gen = a.gen + (size_t)s * 2 * nb;
	int16_t *row = a.blocks + (size_t)s * a.stride;
	const int n = min(max(len, 0), a.cap);
	uint32_t index = meta.x & 0xffffu, used = meta.y & 0xffffu, head = meta.z;
	if (n == 0) return;
	for (int i = lane; i < 2 * T; i += 64) L.cont[i] = g_cont[i];

	if constexpr (!CONCEAL) { // ---- a block arrived: msgenericplc.c:63-116
		for (int i = lane; i < n; i += 64) L.blk[i] = row[i];
		wave_sync();
		update_history(g_hist, head, L.blk, n, nb, lane);
		// generic_plc_update_continuity_buffer :212-231: the block leaves 5 ms late
		const int Tc = min(T, n);
		for (int i = lane; i < n; i += 64) L.out[i] = i < Tc ? L.cont[i] : L.blk[i - Tc];
		wave_sync();
		for (int i = lane; i < Tc; i += 64) L.cont[i] = L.blk[n - Tc + i];
		wave_sync();
		if ((mode & 4) && n >= 2 * T) { // resuming after comfort noise (silence without bcg729): :76-89
			for (int i = lane; i < T; i += 64) {
				L.out[i] = 0;
				L.out[T + i] = mix1(L.out[T + i], 0, i, T);
			}
			wave_sync();
		}
		if (used != 0) { // resuming after concealment: cross-fade with what the concealer would have played :91-112
			if (n >= 2 * T) {
				for (int i = lane; i < T; i += 64) L.out[T + i] = mix1(L.out[T + i], L.cont[T + i], i, T);
			} else {
				for (int i = lane; i < T; i += 64) L.cont[i] = mix1(L.cont[i], L.cont[T + i], i, T); // the block is inside the buffer
			}
			wave_sync();
		}
		index = 0;
		used = 0;
	} else { // ---- nothing arrived: generic_plc_generate_samples genericplc.c:123-197, then :155 remembers it
		const uint32_t max_len = (uint32_t)(MAX_PLC_LEN * a.rate / 1000), fade_from = (uint32_t)(PLC_DECREASE_START * a.rate / 1000);
		if (used >= max_len) { // :127-133
			used = (used + (uint32_t)n) & 0xffffu;
			for (int i = lane; i < n; i += 64) L.out[i] = 0;
			for (int i = lane; i < 2 * T; i += 64) L.cont[i] = 0;
			wave_sync();
		} else {
			bool gen_dirty = false;
			if (used == 0) { // first missing packet :136-144
				const uint32_t h = head;
				fftbf(a, L, [&](int i) { return g_hist[(h + (uint32_t)i) % (uint32_t)nb]; }, lane);
				for (int i = lane; i < T; i += 64) L.gen[i] = mix1(L.gen[i], L.cont[i], i, T);
				wave_sync();
				gen_dirty = true;
			} else {
				for (int i = lane; i < 2 * nb; i += 64) L.gen[i] = g_gen[i];
				wave_sync();
			}
			if ((int)index + n + 2 * T > 2 * nb) { // the generated signal runs out: extend it from itself :148-175
				int ready = (int)((uint32_t)(2 * nb - (int)index - T) & 0xffffu);
				if (ready > n) ready = n;
				for (int i = lane; i < ready; i += 64) L.out[i] = L.gen[index + i];
				for (int i = lane; i < T; i += 64) L.cont[i] = L.gen[index + ready + i];
				wave_sync();
				const int16_t *g = L.gen;
				fftbf(a, L, [&](int i) { return g[i]; }, lane);
				for (int i = lane; i < T; i += 64) L.gen[i] = mix1(L.gen[i], L.cont[i], i, T);
				wave_sync();
				for (int i = lane; i < n - ready; i += 64) L.out[ready + i] = L.gen[i];
				index = (uint32_t)(n - ready);
				for (int i = lane; i < 2 * T; i += 64) L.cont[i] = L.gen[index + i];
				gen_dirty = true;
			} else { // :176-184
				for (int i = lane; i < n; i += 64) L.out[i] = L.gen[index + i];
				index = (index + (uint32_t)n) & 0xffffu;
				for (int i = lane; i < 2 * T; i += 64) L.cont[i] = L.gen[index + i];
			}
			wave_sync();
			if (used + (uint32_t)n > fade_from) { // :187-203 (a double expression: the literal 1.0)
				const int from = max((int)fade_from - (int)used, 0);
				for (int i = from + lane; i < n; i += 64) {
					if (used + (uint32_t)i >= max_len) L.out[i] = 0;
					else {
						const float q = __fdiv_rn((float)((int)fade_from - (int)(used + (uint32_t)i)),
						                          (float)((MAX_PLC_LEN - PLC_DECREASE_START) * a.rate / 1000));
						L.out[i] = (int16_t)(int)((1.0 + (double)q) * (double)(float)L.out[i]);
					}
				}
				wave_sync();
			}
			used = (used + (uint32_t)n) & 0xffffu;
			if (gen_dirty)
				for (int i = lane; i < 2 * nb; i += 64) g_gen[i] = L.gen[i];
		}
		update_history(g_hist, head, L.out, n, nb, lane);
	}
	for (int i = lane; i < n; i += 64) row[i] = L.out[i];
	for (int i = lane; i < 2 * T; i += 64) g_cont[i] = L.cont[i];
	if (lane == 0) {
		a.meta[4 * s] = index;
		a.meta[4 * s + 1] = used;
		a.meta[4 * s + 2] = head;
	}
}

// the streams to conceal this tick, compacted: one atomic per wavefront of 64 streams
__global__ __launch_bounds__(256) void plc_list_kernel(PlcArgs a) {
	const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
	const bool c = s < a.nstreams && (a.mode[s] & 3) == 2 && a.len[s] > 0;
	const unsigned long long mask = __ballot(c);
	if (mask == 0) return;
	int base = 0;
	if (lane == 0) base = atomicAdd(a.count, __popcll(mask));
	base = __shfl(base, 0);
	if (c) a.list[base + __popcll(mask & ((1ull << lane) - 1ull))] = s;
}

__global__ __launch_bounds__(64 * PLC_LIGHT_WAVES) void plc_received_kernel(PlcArgs a) {
	extern __shared__ float lds_all[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int s = blockIdx.x * PLC_LIGHT_WAVES + wave;
	if (s >= a.nstreams) return;
	// one round trip for everything the decision needs
	const int mode = a.mode[s], len = a.len[s];
	const uint4 meta = reinterpret_cast<const uint4 *>(a.meta)[s];
	if ((mode & 3) != 1) return;
	plc_stream<false>(a, lds_all + (size_t)wave * a.light_floats, s, mode, len, meta, lane);
}

__global__ __launch_bounds__(64) void plc_conceal_kernel(PlcArgs a) {
	extern __shared__ float lds_all[];
	const int lane = threadIdx.x;
	const int count = *a.count;
	for (int k = blockIdx.x; k < count; k += gridDim.x) {
		const int s = a.list[k];
		plc_stream<true>(a, lds_all, s, MI_PLC_CONCEAL, a.len[s], reinterpret_cast<const uint4 *>(a.meta)[s], lane);
		wave_sync(); // the next stream reuses the LDS
	}
}

// kf_factor kiss_fft.c:412-435: 4s first, then 2, 3, 5, ...
bool factorize(int n, Factors *f) {
	f->n = n;
	f->count = 0;
	int p = 4, left = n, stride = 1;
	do {
		while (left % p) {
			switch (p) {
			case 4: p = 2; break;
			case 2: p = 3; break;
			default: p += 2; break;
			}
			if (p > 32000 || p * p > left) p = left;
		}
		left /= p;
		if (p > 17 || f->count >= MAXFAC) return false; // kiss_fft.c:266: the generic butterfly takes a radix up to 17
		f->p[f->count] = p;
		f->m[f->count] = left;
		f->fs[f->count] = stride;
		stride *= p;
		f->count++;
	} while (left > 1);
	return true;
}

void twiddles(int n, bool inverse, std::vector<float2> &tw, std::vector<float2> &sup) {
	const double pi = 3.14159265358979323846264338327;
	tw.resize((size_t)n);
	sup.resize((size_t)n);
	for (int i = 0; i < n; ++i) {
		double phase = (-2 * pi / n) * i; // kiss_fft.c:464-471
		if (inverse) phase *= -1;
		tw[(size_t)i] = make_float2((float)std::cos(phase), (float)std::sin(phase));
		double sp = pi * (((double)i) / n + .5); // kiss_fftr.c:68-81
		if (!inverse) sp = -sp;
		sup[(size_t)i] = make_float2((float)std::cos(sp), (float)std::sin(sp));
	}
}

} // namespace

struct mi_plc {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, rate = 0, nb = 0, T = 0, cap = 0;
	Factors f1, f2;
	int16_t *d_cont = nullptr, *d_hist = nullptr, *d_gen = nullptr;
	uint32_t *d_meta = nullptr;
	float *d_window = nullptr;
	float2 *d_tw1 = nullptr, *d_sup1 = nullptr, *d_tw2 = nullptr, *d_sup2 = nullptr;
	size_t lds = 0, lds_light = 0;
	int *d_count = nullptr, *d_list = nullptr;
};

extern "C" {

void mi_plc_destroy(mi_plc *p) {
	if (!p) return;
	if (p->ctx->activate() == MI_OK) {
		(void)hipStreamSynchronize(p->ctx->stream);
		void *dv[] = {p->d_cont, p->d_hist, p->d_gen, p->d_meta, p->d_window, p->d_tw1, p->d_sup1, p->d_tw2, p->d_sup2, p->d_count, p->d_list};
		for (void *v : dv)
			if (v) (void)hipFree(v);
	}
	delete p;
}

int mi_plc_create(mi_ctx *c, int nstreams, int rate, int max_block, mi_plc **out) {
	MI_CHECK_ARG(c && out && nstreams > 0 && rate >= 8000 && rate <= 48000 && max_block > 0 && max_block <= 4096);
	*out = nullptr;
	int rc;
	if ((rc = c->activate()) != MI_OK) return rc;
	mi_plc *p = new mi_plc();
	p->ctx = c, p->nstreams = nstreams, p->rate = rate, p->cap = max_block;
	p->nb = ((rate * 2 / 40) / 100) * 100; // generic_plc_create_context genericplc.c:46-48 (PLC_BUFFER_LEN is the tokens 2 / 40)
	p->T = rate * TRANSITION_DELAY / 1000;
	if (p->nb < 4 || (p->nb & 1) || p->T > 256 || !factorize(p->nb / 2, &p->f1) || !factorize(p->nb, &p->f2)) {
		mi::set_error("mi_plc_create: %d Hz gives transforms of %d / %d points with a prime factor above 17 (kiss_fft itself refuses those)",
		              rate, p->nb / 2, p->nb);
		delete p;
		return MI_ENOTSUP;
	}
	const size_t n = (size_t)nstreams;
	p->lds = sizeof(float) * 4 * (size_t)p->nb + sizeof(int16_t) * (2 * (size_t)p->nb + 2 * (size_t)p->T + 2 * (size_t)p->cap);
	p->lds = (p->lds + 15) & ~(size_t)15;
	p->lds_light = (sizeof(int16_t) * (2 * (size_t)p->T + 2 * (size_t)p->cap) + 15) & ~(size_t)15;
	if (p->lds > 64 * 1024) {
		mi::set_error("mi_plc_create: %zu bytes of LDS per stream (rate %d, blocks of %d) exceed 64 KB", p->lds, rate, max_block);
		delete p;
		return MI_ENOTSUP;
	}
	auto fail = [&](int code) {
		mi_plc_destroy(p);
		return code;
	};
	if (hipMalloc(&p->d_cont, n * 2 * p->T * 2) != hipSuccess || hipMalloc(&p->d_hist, n * p->nb * 2) != hipSuccess ||
	    hipMalloc(&p->d_gen, n * 2 * p->nb * 2) != hipSuccess || hipMalloc(&p->d_meta, n * 16) != hipSuccess ||
	    hipMalloc(&p->d_window, sizeof(float) * p->nb) != hipSuccess || hipMalloc(&p->d_tw1, sizeof(float2) * (p->nb / 2)) != hipSuccess ||
	    hipMalloc(&p->d_sup1, sizeof(float2) * (p->nb / 2)) != hipSuccess || hipMalloc(&p->d_tw2, sizeof(float2) * p->nb) != hipSuccess ||
	    hipMalloc(&p->d_sup2, sizeof(float2) * p->nb) != hipSuccess || hipMalloc(&p->d_count, sizeof(int)) != hipSuccess ||
	    hipMalloc(&p->d_list, sizeof(int) * n) != hipSuccess) {
		mi::set_error("mi_plc_create: out of device memory");
		return fail(MI_ENOMEM);
	}
	std::vector<float> win((size_t)p->nb);
	for (int i = 0; i < p->nb; ++i) win[(size_t)i] = (float)(0.75 - 0.25 * std::cos(2 * 3.14159265 * i / (p->nb))); // genericplc.c:63-65
	std::vector<float2> tw1, sup1, tw2, sup2;
	twiddles(p->nb / 2, false, tw1, sup1);
	twiddles(p->nb, true, tw2, sup2);
	if (hipMemcpy(p->d_window, win.data(), sizeof(float) * win.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(p->d_tw1, tw1.data(), sizeof(float2) * tw1.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(p->d_sup1, sup1.data(), sizeof(float2) * sup1.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(p->d_tw2, tw2.data(), sizeof(float2) * tw2.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(p->d_sup2, sup2.data(), sizeof(float2) * sup2.size(), hipMemcpyHostToDevice) != hipSuccess) {
		mi::set_error("mi_plc_create: table upload failed");
		return fail(MI_ENODEV);
	}
	if ((rc = mi_plc_reset(p, 0, nstreams)) != MI_OK) return fail(rc);
	*out = p;
	return MI_OK;
}

int mi_plc_reset(mi_plc *p, int first, int count) { // a fresh plc_context_t: everything zero (ms_malloc0, genericplc.c:42-57)
	MI_CHECK_ARG(p && first >= 0 && count >= 0 && first + count <= p->nstreams);
	if (count == 0) return MI_OK;
	int rc;
	if ((rc = p->ctx->activate()) != MI_OK) return rc;
	hipStream_t st = p->ctx->stream;
	MI_HIP(hipMemsetAsync(p->d_cont + (size_t)first * 2 * p->T, 0, (size_t)count * 2 * p->T * 2, st));
	MI_HIP(hipMemsetAsync(p->d_hist + (size_t)first * p->nb, 0, (size_t)count * p->nb * 2, st));
	MI_HIP(hipMemsetAsync(p->d_gen + (size_t)first * 2 * p->nb, 0, (size_t)count * 2 * p->nb * 2, st));
	MI_HIP(hipMemsetAsync(p->d_meta + (size_t)first * 4, 0, (size_t)count * 16, st));
	return MI_OK;
}

int mi_plc_process(mi_plc *p, int16_t *d_blocks, size_t stride, const int32_t *d_len, const uint8_t *d_mode) {
	MI_CHECK_ARG(p && d_blocks && d_len && d_mode && stride > 0);
	int rc;
	if ((rc = p->ctx->activate()) != MI_OK) return rc;
	PlcArgs a{};
	a.cont = p->d_cont, a.hist = p->d_hist, a.gen = p->d_gen, a.meta = p->d_meta, a.window = p->d_window;
	a.tw1 = p->d_tw1, a.sup1 = p->d_sup1, a.tw2 = p->d_tw2, a.sup2 = p->d_sup2;
	a.f1 = p->f1, a.f2 = p->f2;
	a.nb = p->nb, a.T = p->T, a.rate = p->rate, a.nstreams = p->nstreams, a.cap = (int)std::min<size_t>((size_t)p->cap, stride);
	a.blocks = d_blocks, a.stride = stride, a.len = d_len, a.mode = d_mode;
	a.light_floats = (int)(p->lds_light / sizeof(float));
	a.count = p->d_count, a.list = p->d_list;
	MI_HIP(hipMemsetAsync(p->d_count, 0, sizeof(int), p->ctx->stream));
	hipLaunchKernelGGL(plc_list_kernel, dim3((p->nstreams + 255) / 256), dim3(256), 0, p->ctx->stream, a);
	MI_LAUNCH_CHECK();
	hipLaunchKernelGGL(plc_received_kernel, dim3((p->nstreams + PLC_LIGHT_WAVES - 1) / PLC_LIGHT_WAVES), dim3(64 * PLC_LIGHT_WAVES),
	                   p->lds_light * PLC_LIGHT_WAVES, p->ctx->stream, a);
	MI_LAUNCH_CHECK();
	const int grid = std::min(p->nstreams, (p->ctx->cu_count > 0 ? p->ctx->cu_count : 256) * 16);
	hipLaunchKernelGGL(plc_conceal_kernel, dim3(grid), dim3(64), p->lds, p->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_plc_info(mi_plc *p, int stream, int32_t out3[3]) {
	MI_CHECK_ARG(p && out3 && stream >= 0 && stream < p->nstreams);
	int rc;
	if ((rc = p->ctx->activate()) != MI_OK) return rc;
	uint32_t m[4];
	MI_HIP(hipMemcpyAsync(m, p->d_meta + (size_t)stream * 4, 16, hipMemcpyDeviceToHost, p->ctx->stream));
	MI_HIP(hipStreamSynchronize(p->ctx->stream));
	out3[0] = p->nb, out3[1] = (int32_t)m[0], out3[2] = (int32_t)m[1];
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_plc(reinterpret_cast<const void *>(&plc_list_kernel));
