// resample.hip -- batched polyphase resampler for gfx950 (MI355X).
//
// Replaces, for a whole batch of streams per launch, what
// src/audiofilters/msresample.c:150-177 does per stream through
// speex_resampler_process_int (libspeexdsp, un-vendored): Kaiser-windowed-sinc
// polyphase FIR, quality 3 ("VOIP": 48 taps/phase, oversample 8, cut-offs
// 0.917 up / 0.895 down).  Filter design runs once on the host at create time;
// per tick the kernels stream packed int16 frames HBM -> LDS -> HBM.
//
// Kernels
//   resample_up_kernel<DEN,FILT,R>  integer up-sampling (num_rate == 1, e.g.
//       16k->48k, 8k->48k): persistent wavefronts, one stream at a time per
//       wave, next stream's row prefetched while the current one is computed.
//       Each lane owns one polyphase row (FILT taps in VGPRs) and R consecutive
//       input positions; the FILT-1+R sample window comes out of LDS with
//       16-byte reads; outputs are staged in LDS and leave as 16-byte stores.
//   resample_down_kernel<NUM,48,8>  integer down-sampling (den_rate == 1, e.g.
//       48k->16k, 48k->8k, 16k->8k): the input is split by phase while it is
//       staged and each lane (tile, phase) runs the same 48-tap tile FIR on its
//       phase; the NUM shares of an output meet in LDS.
//   resample_ratio_kernel<L,M,24,8> L/M = 3/2 or 2/3 (32k<->48k, 16k<->24k, ...):
//       M input phases x L output residues, the same tile FIR with 24 taps.
//   resample_generic_kernel         any other ratio (direct table or the
//       oversampled table + 4-point cubic interpolation), one block per stream.
//
// HBM traffic per stream-tick (16k->48k): 320 B in + 960 B out (+ 96 B history
// read + 96 B written).
#include "common.hpp"

#include <cmath>

namespace {

// ---------------------------------------------------------------- host design
// Kaiser(beta=8) window sampled at (i-1)/32, as tabulated by the library for
// its quality 3/4 rows.
const double kKaiser8[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751, 0.87580811,
    0.83439927, 0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482, 0.52566725, 0.47185369,
    0.41941150, 0.36897272, 0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758,
    0.10562887, 0.08273982, 0.06335451, 0.04724088, 0.03412321, 0.02369490, 0.01563093, 0.00959968,
    0.00527363, 0.00233883, 0.00050000, 0.00000000};

double kaiser8_at(float x) {
	const float y = x * 32;
	const int ind = (int)floor(y);
	const float frac = y - ind;
	double c[4];
	c[3] = -0.1666666667 * frac + 0.1666666667 * (frac * frac * frac);
	c[2] = frac + 0.5 * (frac * frac) - 0.5 * (frac * frac * frac);
	c[0] = -0.3333333333 * frac + 0.5 * (frac * frac) - 0.1666666667 * (frac * frac * frac);
	c[1] = 1.f - c[3] - c[2] - c[0];
	return c[0] * kKaiser8[ind] + c[1] * kKaiser8[ind + 1] + c[2] * kKaiser8[ind + 2] + c[3] * kKaiser8[ind + 3];
}

float windowed_sinc(float cutoff, float x, int N) {
	const float xx = x * cutoff;
	if (fabs(x) < 1e-6) return cutoff;
	if (fabs(x) > .5 * N) return 0;
	return (float)(cutoff * sin(M_PI * xx) / (M_PI * xx) * kaiser8_at((float)fabs(2. * x / N)));
}

struct Design {
	uint32_t num = 0, den = 0, filt_len = 0, oversample = 0;
	int int_advance = 0, frac_advance = 0, direct = 0;
	float cutoff = 0;
	std::vector<float> table;
	// For the interpolated mode (filt_len*den > filt_len*oversample + 8, e.g. 44.1k <-> 48k): the library evaluates four
	// partial sums against the oversampled table and blends them with cubic weights that depend only on the output's
	// phase.  The same filter, blended ONCE per phase on the host: row ph = sum_k interp_k(ph) * table[.. + k].  The
	// kernels then run it like a direct table (one FMA per tap instead of four); the result differs from the library's
	// evaluation order by float rounding only (tolerance of the float path, tests hold it to 1 LSB / 1e-4 RMS).
	std::vector<float> phase_table; // [den][filt_len], empty when the table is direct or too large to be worth it
};

uint32_t gcd_u32(uint32_t a, uint32_t b) {
	while (b) {
		uint32_t t = a % b;
		a = b;
		b = t;
	}
	return a;
}

bool design_filter(uint32_t in_rate, uint32_t out_rate, int quality, Design &d) {
	int base;
	float down_bw, up_bw;
	if (quality == 3) {
		base = 48, down_bw = 0.895f, up_bw = 0.917f;
	} else if (quality == 4) {
		base = 64, down_bw = 0.921f, up_bw = 0.940f;
	} else {
		return false;
	}
	const uint32_t g = gcd_u32(in_rate, out_rate);
	d.num = in_rate / g;
	d.den = out_rate / g;
	d.int_advance = (int)(d.num / d.den);
	d.frac_advance = (int)(d.num % d.den);
	d.oversample = 8;
	d.filt_len = (uint32_t)base;
	if (d.num > d.den) {
		d.cutoff = down_bw * d.den / d.num;
		d.filt_len = d.filt_len * d.num / d.den;
		d.filt_len = ((d.filt_len - 1) & (~0x7u)) + 8;
		if (2 * d.den < d.num) d.oversample >>= 1;
		if (4 * d.den < d.num) d.oversample >>= 1;
		if (8 * d.den < d.num) d.oversample >>= 1;
		if (16 * d.den < d.num) d.oversample >>= 1;
		if (d.oversample < 1) d.oversample = 1;
	} else {
		d.cutoff = up_bw;
	}
	d.direct = d.filt_len * d.den <= d.filt_len * d.oversample + 8;
	if (d.direct) {
		d.table.resize((size_t)d.filt_len * d.den);
		for (uint32_t i = 0; i < d.den; i++)
			for (int32_t j = 0; j < (int32_t)d.filt_len; j++)
				d.table[i * d.filt_len + j] = windowed_sinc(
				    d.cutoff, ((j - (int32_t)d.filt_len / 2 + 1) - ((float)i) / d.den), (int)d.filt_len);
	} else {
		d.table.resize((size_t)d.filt_len * d.oversample + 8);
		for (int32_t k = -4; k < (int32_t)(d.oversample * d.filt_len + 4); k++)
			d.table[k + 4] = windowed_sinc(d.cutoff, (k / (float)d.oversample - d.filt_len / 2), (int)d.filt_len);
		if ((size_t)d.den * d.filt_len * sizeof(float) <= 256 * 1024) {
			d.phase_table.resize((size_t)d.den * d.filt_len);
			for (uint32_t ph = 0; ph < d.den; ++ph) {
				const int offset = (int)(ph * d.oversample / d.den);
				const float fr = ((float)((ph * d.oversample) % d.den)) / d.den;
				// cubic_coef of the library, in its float arithmetic
				const float i0 = -0.16667f * fr + 0.16667f * fr * fr * fr;
				const float i1 = fr + 0.5f * fr * fr - 0.5f * fr * fr * fr;
				const float i3 = -0.33333f * fr + 0.5f * fr * fr - 0.16667f * fr * fr * fr;
				const float i2 = (float)(1. - i0 - i1 - i3);
				for (uint32_t j = 0; j < d.filt_len; ++j) {
					const float *tt = d.table.data() + 4 + (j + 1) * d.oversample - offset;
					d.phase_table[(size_t)ph * d.filt_len + j] =
					    (float)((double)i0 * tt[-2] + (double)i1 * tt[-1] + (double)i2 * tt[0] + (double)i3 * tt[1]);
				}
			}
		}
	}
	return true;
}

// ------------------------------------------------------------------- kernels
#include "resample_tile.hpp"


struct UpArgs {
	const int16_t *in;
	int16_t *out;
	int32_t *out_len;
	int16_t *hist;
	const float *table;
	const uint8_t *run;
	int in_len, in_stride, out_stride, hist_stride, nstreams;
	int tiles; // ceil(in_len / R)
	int lds_per_wave;
};

// Integer up-sampling (num == 1): out[m*DEN + p] = sum_j table[p][j] * x[m + j].
// Persistent wavefronts, one stream at a time per wave, no workgroup barriers.  A wave walks streams
// blockIdx.x, +gridDim.x, ...; the next stream's history and input (8-byte loads, at most two per lane)
// are already in flight while the current one is computed, so a wave always has a row of HBM reads
// outstanding.  Lane (tile, phase) owns one polyphase row (FILT taps in VGPRs, loaded once per wave)
// and R consecutive input positions; its FILT-1+R sample window comes out of LDS with 16-byte reads
// (window start = 32*tile bytes, DEN lanes per address -> broadcast), outputs are staged in LDS and
// leave as 16-byte stores.
constexpr int UP_WAVES = 4; // most wavefronts per workgroup of the one-wave-per-stream kernels (the launchers pick 2 or 4)
inline int waves_per_workgroup(int nstreams) { return nstreams <= 16384 ? UP_WAVES : 2; }

template <int DEN, int FILT, int R, bool MULTI, bool TWO>
__global__ __launch_bounds__(64 * UP_WAVES, 2) void resample_up_kernel(UpArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_all[];
	// Two or four independent wavefronts share a workgroup only to cut the number of workgroups the dispatcher has to
	// place (0.4 us, then another 0.25 us, of a 6.6 us launch at 4096 streams); each has its own LDS slice and never
	// waits for the others.  Four per workgroup lose 10 % on big batches (62 vs 56 us at 65 536 streams), so the
	// launchers take four up to 16 384 streams and two above.
	char *smem = smem_all + (size_t)(threadIdx.x >> 6) * a.lds_per_wave;
	constexpr int HIST = FILT - 1;
	const int lane = threadIdx.x & 63;
	const int out_per_stream = a.in_len * DEN;
	const int xn = ((HIST + a.in_len + R + 1) + 3) & ~3;
	float *x = reinterpret_cast<float *>(smem);                         // [xn] history ++ input ++ zero slack
	int16_t *obuf = reinterpret_cast<int16_t *>(smem + (size_t)xn * 4); // [out_per_stream]

	// ---- staging plan: quads [0, hq) are history, [hq, hq+iq) input; lane takes quads lane and lane+64
	const int hq = a.hist_stride >> 2, nq = hq + (a.in_len >> 2);
	constexpr bool two = TWO; // nq > 64: a second staging quad per lane
	auto quad_ptr = [&](int s, int i) -> const short4 * {
		const int16_t *p = i < hq ? a.hist + (size_t)s * a.hist_stride + 4 * i
		                          : a.in + (size_t)s * a.in_stride + 4 * (i - hq);
		return reinterpret_cast<const short4 *>(p);
	};
	// ---- the first stream's row goes out FIRST (it is the HBM miss on the critical path of a one-stream wave), the
	// run flag next, the L2-resident tap rows last; nothing waits before the row is needed for staging
	short4 v0 = make_short4(0, 0, 0, 0), v1 = v0;
	const int nwaves = gridDim.x * (blockDim.x >> 6);
	int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	int runv = 1;
	if (s < a.nstreams) {
		if (lane < nq) v0 = *quad_ptr(s, lane);
		if (two && lane + 64 < nq) v1 = *quad_ptr(s, lane + 64);
		if (a.run) runv = a.run[s];
	}
	// ---- the polyphase table (DEN x FILT floats, L2-resident) goes through LDS: one 16-byte load per lane instead of
	// FILT/4 per lane (every lane of a phase would fetch the same row: 12 KB per wave through the vector L1)
	const int nlanes = DEN * a.tiles;
	float4 *tab4 = reinterpret_cast<float4 *>(smem + (size_t)xn * 4 + (size_t)((out_per_stream + 7) & ~7) * 2);
	for (int i = lane; i < DEN * FILT / 4; i += 64) tab4[i] = reinterpret_cast<const float4 *>(a.table)[i];
	f2 t2[FILT / 2]; // taps as register pairs: v_pk_fma_f32 broadcasts either half through op_sel
	auto load_row = [&](int p) {
		const float4 *tp = tab4 + p * (FILT / 4);
#pragma unroll
		for (int j = 0; j < FILT / 4; ++j) {
			const float4 v = tp[j];
			t2[2 * j] = (f2){v.x, v.y}, t2[2 * j + 1] = (f2){v.z, v.w};
		}
	};
	for (int i = HIST + a.in_len + lane; i < xn; i += 64) x[i] = 0.f; // slack stays zero for every stream
	wave_sync(); // single wave: LDS fence, the table is in place
	if (!MULTI) {
		const int l0 = lane < nlanes ? lane : 0;
		load_row(l0 - (l0 / DEN) * DEN);
	}

	for (; s < a.nstreams; s += nwaves) {
		// ---- the prefetched row goes to LDS as float
		if (lane < nq) {
			const int b = 4 * lane - (lane < hq ? 0 : 4 * hq - HIST);
			if (lane < hq) {
				if (b + 0 < HIST) x[b + 0] = (float)v0.x;
				if (b + 1 < HIST) x[b + 1] = (float)v0.y;
				if (b + 2 < HIST) x[b + 2] = (float)v0.z;
				if (b + 3 < HIST) x[b + 3] = (float)v0.w;
			} else {
				x[b + 0] = (float)v0.x, x[b + 1] = (float)v0.y, x[b + 2] = (float)v0.z, x[b + 3] = (float)v0.w;
			}
		}
		if (two && lane + 64 < nq) {
			const int b = HIST + 4 * (lane + 64 - hq);
			x[b + 0] = (float)v1.x, x[b + 1] = (float)v1.y, x[b + 2] = (float)v1.z, x[b + 3] = (float)v1.w;
		}
		const int cur_run = runv;
		// ---- next stream's row: in flight during this stream's arithmetic
		const int sn = s + nwaves;
		if (sn < a.nstreams) {
			if (lane < nq) v0 = *quad_ptr(sn, lane);
			if (two && lane + 64 < nq) v1 = *quad_ptr(sn, lane + 64);
			if (a.run) runv = a.run[sn];
		}
		if (!cur_run) { // masked out: no output, state untouched
			if (lane == 0 && a.out_len) a.out_len[s] = 0;
			wave_sync();
			continue;
		}
		for (int base = 0; base < (MULTI ? nlanes : 1); base += 64) { // !MULTI: nlanes <= 64, one trip
			const int l = base + lane;
			const bool on = l < nlanes;
			const int tile = on ? l / DEN : 0, p = on ? l - tile * DEN : 0;
			const int m0 = tile * R;
			if (MULTI) load_row(p); // several trips: the phase pattern shifts by 64 % DEN
			wave_sync(); // single wave: LDS fence between staging and the window reads
			f2 acc2[R / 2];
#pragma unroll
			for (int q = 0; q < R / 2; ++q) acc2[q] = (f2){0.f, 0.f};
			fir_tile<FILT, R>(x + m0, t2, acc2);
			float acc[R];
#pragma unroll
			for (int q = 0; q < R / 2; ++q) acc[2 * q] = acc2[q].x, acc[2 * q + 1] = acc2[q].y;
			if (on) {
#pragma unroll
				for (int r = 0; r < R; ++r)
					if (m0 + r < a.in_len) obuf[(m0 + r) * DEN + p] = rs_word2int(acc[r]);
			}
		}
		wave_sync();
		// ---- outputs: 16-byte coalesced stores (launch_up checked the layout)
		int16_t *o = a.out + (size_t)s * a.out_stride;
		for (int q = lane; q < (out_per_stream >> 3); q += 64)
			*reinterpret_cast<uint4 *>(o + 8 * q) = *reinterpret_cast<const uint4 *>(obuf + 8 * q);
		// ---- new history = last FILT-1 samples of (history ++ input); the pad slot takes the zero slack
		if (lane < hq) {
			const float *hx = x + a.in_len + 4 * lane;
			short4 h;
			h.x = (int16_t)hx[0], h.y = (int16_t)hx[1], h.z = (int16_t)hx[2], h.w = (int16_t)hx[3];
			*reinterpret_cast<short4 *>(a.hist + (size_t)s * a.hist_stride + 4 * lane) = h;
		}
		if (lane == 0 && a.out_len) a.out_len[s] = out_per_stream;
		wave_sync(); // LDS reads above complete before the next row overwrites x / obuf
	}
}

// Integer down-sampling (den == 1, e.g. 48k->16k, 48k->8k, 16k->8k): out[k] = sum_j T[j] * X[NUM*k + j] with
// FILT*NUM taps, X = history ++ input.  Split by input phase, X_p[m] = X[NUM*m + p], it is NUM stride-1 FIRs of FILT
// taps each, out[k] = sum_p sum_i T[NUM*i + p] * X_p[k + i] -- the up-sampler's tile FIR with the roles of the phases
// turned around: lane (tile, phase) computes phase p's share of R = 8 consecutive outputs, the NUM shares meet in LDS.
// One wavefront per stream (persistent, UP_WAVES per workgroup), input de-interleaved by phase while it is staged.
// Accumulation is phase-major (the library runs j upward): within 1 LSB of it, like the FMA contraction already is.
struct DownArgs {
	const int16_t *in;
	int16_t *out;
	int32_t *out_len;
	int16_t *hist;
	const float *table; // natural order, FILT*NUM taps
	const uint8_t *run;
	int in_len, in_stride, out_stride, hist_stride, nstreams;
	int tiles;  // ceil(out_len / R)
	int plen;   // floats per phase array (multiple of 4)
	int lds_per_wave;
};

template <int NUM, int FILT, int R>
__global__ __launch_bounds__(64 * UP_WAVES, 2) void resample_down_kernel(DownArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_all[];
	char *smem = smem_all + (size_t)(threadIdx.x >> 6) * a.lds_per_wave;
	constexpr int NT = NUM * FILT, HIST = NT - 1;
	const int lane = threadIdx.x & 63;
	const int out_len = a.in_len / NUM;
	float *xp = reinterpret_cast<float *>(smem);                       // [NUM][plen] phase arrays of history ++ input
	float *part = xp + (size_t)NUM * a.plen;                           // [tiles*R][NUM] partial sums
	float4 *tab4 = reinterpret_cast<float4 *>(part + (size_t)a.tiles * R * NUM); // [NUM][FILT] taps, phase-major
	{
		float *tab = reinterpret_cast<float *>(tab4);
		for (int i = lane; i < NT; i += 64) tab[(i % NUM) * FILT + i / NUM] = a.table[i];
	}
	// zero the tails of the phase arrays once (slack the window reads may touch)
	for (int i = lane; i < NUM * a.plen; i += 64) xp[i] = 0.f;
	wave_sync();
	const int hq = a.hist_stride >> 2, nq = hq + (a.in_len >> 2);
	const int nlanes = NUM * a.tiles;
	const int nwaves = gridDim.x * (blockDim.x >> 6);
	f2 t2[FILT / 2];
	auto load_row = [&](int p) {
		const float4 *tp = tab4 + p * (FILT / 4);
#pragma unroll
		for (int j = 0; j < FILT / 4; ++j) {
			const float4 v = tp[j];
			t2[2 * j] = (f2){v.x, v.y}, t2[2 * j + 1] = (f2){v.z, v.w};
		}
	};
	if (nlanes <= 64) {
		const int l0 = lane < nlanes ? lane : 0;
		load_row(l0 - (l0 / NUM) * NUM);
	}
	for (int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < a.nstreams; s += nwaves) {
		if (a.run && !a.run[s]) {
			if (lane == 0 && a.out_len) a.out_len[s] = 0;
			continue;
		}
		// ---- stage history ++ input, de-interleaved by phase; up to three 8-byte loads per lane, all issued first
		const int16_t *hs = a.hist + (size_t)s * a.hist_stride, *xin = a.in + (size_t)s * a.in_stride;
		short4 v[3];
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int q = lane + 64 * u;
			v[u] = make_short4(0, 0, 0, 0);
			if (q < nq) v[u] = *reinterpret_cast<const short4 *>(q < hq ? hs + 4 * q : xin + 4 * (q - hq));
		}
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int q = lane + 64 * u;
			if (q < nq) {
				const int b = q < hq ? 4 * q : HIST + 4 * (q - hq); // index in history ++ input
				const short e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int i = b + k;
					if (q >= hq || i < HIST) xp[(i % NUM) * a.plen + i / NUM] = (float)e[k];
				}
			}
		}
		for (int base = 0; base < nlanes; base += 64) {
			const int l = base + lane;
			const bool on = l < nlanes;
			const int tile = on ? l / NUM : 0, p = on ? l - tile * NUM : 0;
			if (nlanes > 64) load_row(p);
			wave_sync();
			f2 acc2[R / 2];
#pragma unroll
			for (int q = 0; q < R / 2; ++q) acc2[q] = (f2){0.f, 0.f};
			fir_tile<FILT, R>(xp + (size_t)p * a.plen + tile * R, t2, acc2);
			if (on) {
				float *d = part + ((size_t)tile * R) * NUM + p;
#pragma unroll
				for (int q = 0; q < R / 2; ++q) d[(2 * q) * NUM] = acc2[q].x, d[(2 * q + 1) * NUM] = acc2[q].y;
			}
		}
		wave_sync();
		// ---- the NUM shares of each output meet, 8 outputs = one 16-byte store per lane
		int16_t *o = a.out + (size_t)s * a.out_stride;
		for (int g = lane; g < (out_len >> 3); g += 64) {
			const float *ps = part + (size_t)g * 8 * NUM;
			short r16[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				float sum = 0.f;
#pragma unroll
				for (int p = 0; p < NUM; ++p) sum += ps[k * NUM + p];
				r16[k] = rs_word2int(sum);
			}
			uint4 pk;
			pk.x = (unsigned)(r16[0] & 0xffff) | ((unsigned)r16[1] << 16);
			pk.y = (unsigned)(r16[2] & 0xffff) | ((unsigned)r16[3] << 16);
			pk.z = (unsigned)(r16[4] & 0xffff) | ((unsigned)r16[5] << 16);
			pk.w = (unsigned)(r16[6] & 0xffff) | ((unsigned)r16[7] << 16);
			*reinterpret_cast<uint4 *>(o + 8 * g) = pk;
		}
		// ---- new history = last HIST samples of (history ++ input); the pad slot of the row takes a zero
		for (int q = lane; q < hq; q += 64) {
			short4 h;
			short *hp = &h.x;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int i = a.in_len + 4 * q + k;
				hp[k] = (4 * q + k < HIST) ? (short)xp[(i % NUM) * a.plen + i / NUM] : (short)0;
			}
			*reinterpret_cast<short4 *>(a.hist + (size_t)s * a.hist_stride + 4 * q) = h;
		}
		if (lane == 0 && a.out_len) a.out_len[s] = out_len;
		wave_sync(); // LDS reads above complete before the next stream is staged
	}
}

// Rational ratios L/M with a direct table (32k<->48k, 16k<->24k, 8k<->12k: L/M = 3/2 or 2/3).  Output n = q*L + r reads
// table row ph_r = (r*M) % L at input position q*M + off_r, off_r = (r*M) / L: for a fixed residue r consecutive q shift
// the window by M samples, so -- as in the down-sampler -- the input is split into M phases X_p[m] = X[m*M + p] and tap
// j = i*M + p' of row ph_r meets X_{u % M}[q + i + u / M], u = off_r + p'.  Lane (tile, r, p') runs the shared tile FIR
// with FT = filt_len / M taps over 8 consecutive q; u / M is 0 or 1, and a window must start on a 16-byte boundary, so
// every phase array is staged twice, the second copy one float early.  The M shares of an output meet in LDS.
struct RatioArgs {
	const int16_t *in;
	int16_t *out;
	int32_t *out_len;
	int16_t *hist;
	const float *table; // [L][filt_len]
	const uint8_t *run;
	int in_len, in_stride, out_stride, hist_stride, nstreams;
	int tiles; // ceil(periods / R), periods = in_len / M
	int plen;  // floats per phase array copy (multiple of 4)
	int lds_per_wave;
};

template <int L, int M, int FT, int R>
__global__ __launch_bounds__(64 * UP_WAVES, 2) void resample_ratio_kernel(RatioArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_all[];
	char *smem = smem_all + (size_t)(threadIdx.x >> 6) * a.lds_per_wave;
	constexpr int NT = FT * M, HIST = NT - 1, ROWS = L * M;
	const int lane = threadIdx.x & 63;
	const int periods = a.in_len / M, out_len = periods * L;
	float *xp = reinterpret_cast<float *>(smem);                // [2][M][plen]: copy c holds X_p[m] at index m + c
	float *part = xp + (size_t)2 * M * a.plen;                  // [tiles*R][L][M] partial sums
	float4 *tab4 = reinterpret_cast<float4 *>(part + (size_t)a.tiles * R * ROWS); // [L*M][FT] taps per (r, p')
	{
		float *tab = reinterpret_cast<float *>(tab4);
		for (int i = lane; i < ROWS * FT; i += 64) {
			const int row = i / FT, k = i - row * FT, r = row / M, pp = row - r * M;
			tab[i] = a.table[((r * M) % L) * NT + k * M + pp];
		}
	}
	for (int i = lane; i < 2 * M * a.plen; i += 64) xp[i] = 0.f;
	wave_sync();
	const int hq = a.hist_stride >> 2, nq = hq + (a.in_len >> 2);
	const int nlanes = ROWS * a.tiles;
	const int nwaves = gridDim.x * (blockDim.x >> 6);
	f2 t2[FT / 2];
	for (int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < a.nstreams; s += nwaves) {
		if (a.run && !a.run[s]) {
			if (lane == 0 && a.out_len) a.out_len[s] = 0;
			continue;
		}
		const int16_t *hs = a.hist + (size_t)s * a.hist_stride, *xin = a.in + (size_t)s * a.in_stride;
		short4 v[3];
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int q = lane + 64 * u;
			v[u] = make_short4(0, 0, 0, 0);
			if (q < nq) v[u] = *reinterpret_cast<const short4 *>(q < hq ? hs + 4 * q : xin + 4 * (q - hq));
		}
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int q = lane + 64 * u;
			if (q < nq) {
				const int b = q < hq ? 4 * q : HIST + 4 * (q - hq);
				const short e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int i = b + k;
					if (q >= hq || i < HIST) {
						const int ph = i % M, m = i / M;
						const float f = (float)e[k];
						xp[ph * a.plen + m] = f;                       // copy 0
						if (m >= 1) xp[(M + ph) * a.plen + m - 1] = f; // copy 1: X_p[m] at m - 1, i.e. window start + 1
					}
				}
			}
		}
		for (int base = 0; base < nlanes; base += 64) {
			const int l = base + lane;
			const bool on = l < nlanes;
			const int tile = on ? l / ROWS : 0, row = on ? l - tile * ROWS : 0;
			const int r = row / M, pp = row - r * M;
			const int u = (r * M) / L + pp; // off_r + p'
			{
				const float4 *tp = tab4 + row * (FT / 4);
#pragma unroll
				for (int j = 0; j < FT / 4; ++j) {
					const float4 w = tp[j];
					t2[2 * j] = (f2){w.x, w.y}, t2[2 * j + 1] = (f2){w.z, w.w};
				}
			}
			wave_sync();
			f2 acc2[R / 2];
#pragma unroll
			for (int q = 0; q < R / 2; ++q) acc2[q] = (f2){0.f, 0.f};
			fir_tile<FT, R>(xp + (size_t)((u / M) * M + (u % M)) * a.plen + tile * R, t2, acc2);
			if (on) {
				float *d = part + ((size_t)tile * R) * ROWS + row;
#pragma unroll
				for (int q = 0; q < R / 2; ++q) d[(2 * q) * ROWS] = acc2[q].x, d[(2 * q + 1) * ROWS] = acc2[q].y;
			}
		}
		wave_sync();
		// ---- output n = q*L + r = the sum over p' of part[q][r][p']: consecutive n are consecutive (q, r) pairs
		int16_t *o = a.out + (size_t)s * a.out_stride;
		for (int g = lane; g < (out_len >> 3); g += 64) {
			const float *ps = part + (size_t)g * 8 * M;
			short r16[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				float sum = 0.f;
#pragma unroll
				for (int pq = 0; pq < M; ++pq) sum += ps[k * M + pq];
				r16[k] = rs_word2int(sum);
			}
			uint4 pk;
			pk.x = (unsigned)(r16[0] & 0xffff) | ((unsigned)r16[1] << 16);
			pk.y = (unsigned)(r16[2] & 0xffff) | ((unsigned)r16[3] << 16);
			pk.z = (unsigned)(r16[4] & 0xffff) | ((unsigned)r16[5] << 16);
			pk.w = (unsigned)(r16[6] & 0xffff) | ((unsigned)r16[7] << 16);
			*reinterpret_cast<uint4 *>(o + 8 * g) = pk;
		}
		for (int q = lane; q < hq; q += 64) {
			short4 h;
			short *hp = &h.x;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int i = a.in_len + 4 * q + k;
				hp[k] = (4 * q + k < HIST) ? (short)xp[(i % M) * a.plen + i / M] : (short)0;
			}
			*reinterpret_cast<short4 *>(a.hist + (size_t)s * a.hist_stride + 4 * q) = h;
		}
		if (lane == 0 && a.out_len) a.out_len[s] = out_len;
		wave_sync();
	}
}

struct GenArgs {
	const int16_t *in;
	int16_t *out;
	int32_t *out_len;
	int16_t *hist;
	int2 *pos;
	const float *table;
	const uint8_t *run;
	int table_len, table_in_lds;
	int in_len, in_stride, out_stride, out_cap, hist_stride, nstreams;
	int filt_len, num, den, oversample, direct;
};

__global__ __launch_bounds__(256) void resample_generic_kernel(GenArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *x = reinterpret_cast<float *>(smem);
	const int N = a.filt_len;
	const int xlen = N - 1 + a.in_len;
	float *tl = x + ((xlen + 3) & ~3);
	const int s = blockIdx.x;
	const int tid = threadIdx.x;
	if (a.run && !a.run[s]) {
		if (tid == 0 && a.out_len) a.out_len[s] = 0;
		return;
	}
	const int16_t *hin = a.hist + (size_t)s * a.hist_stride;
	for (int i = tid; i < N - 1; i += 256) x[i] = (float)hin[i];
	const int16_t *sin_ = a.in + (size_t)s * a.in_stride;
	for (int i = tid; i < a.in_len; i += 256) x[N - 1 + i] = (float)sin_[i];
	const float *tab = a.table;
	if (a.table_in_lds) {
		for (int i = tid; i < a.table_len; i += 256) tl[i] = a.table[i];
		tab = tl;
	}
	const int2 p0 = a.pos[s];
	__syncthreads();

	const long long last0 = p0.x, f0 = p0.y;
	const long long avail = (long long)a.in_len - last0;
	long long n_out = 0;
	if (avail > 0) n_out = (avail * a.den - f0 + a.num - 1) / a.num;
	if (n_out > a.out_cap) n_out = a.out_cap;
	if (n_out < 0) n_out = 0;

	int16_t *o = a.out + (size_t)s * a.out_stride;
	for (int k = tid; k < (int)n_out; k += 256) {
		const long long t = f0 + (long long)k * a.num;
		const int last = (int)(last0 + t / a.den);
		const unsigned frac = (unsigned)(t % a.den);
		const float *ip = x + last;
		float sum;
		if (a.direct) {
			const float *sc = tab + (size_t)frac * N;
			sum = 0.f;
			for (int j = 0; j < N; ++j) sum = __builtin_fmaf(sc[j], ip[j], sum);
		} else {
			const int offset = (int)(frac * (unsigned)a.oversample / (unsigned)a.den);
			const float fr = ((float)((frac * (unsigned)a.oversample) % (unsigned)a.den)) / a.den;
			float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
			for (int j = 0; j < N; ++j) {
				const float c = ip[j];
				const float *tt = tab + 4 + (j + 1) * a.oversample - offset;
				a0 = __builtin_fmaf(c, tt[-2], a0);
				a1 = __builtin_fmaf(c, tt[-1], a1);
				a2 = __builtin_fmaf(c, tt[0], a2);
				a3 = __builtin_fmaf(c, tt[1], a3);
			}
			const float i0 = -0.16667f * fr + 0.16667f * fr * fr * fr;
			const float i1 = fr + 0.5f * fr * fr - 0.5f * fr * fr * fr;
			const float i3 = -0.33333f * fr + 0.5f * fr * fr - 0.16667f * fr * fr * fr;
			const float i2 = (float)(1. - i0 - i1 - i3);
			sum = i0 * a0 + i1 * a1 + i2 * a2 + i3 * a3;
		}
		o[k] = rs_word2int(sum);
	}
	__syncthreads();
	// carry state exactly like speex_resampler_process_native
	const long long tend = f0 + n_out * a.num;
	const long long last_end = last0 + tend / a.den;
	const int consumed = (int)(last_end < a.in_len ? last_end : a.in_len);
	int16_t *hout = a.hist + (size_t)s * a.hist_stride;
	for (int i = tid; i < N - 1; i += 256) hout[i] = (int16_t)x[consumed + i];
	if (tid == 0) {
		a.pos[s] = make_int2((int)(last_end - consumed), (int)(tend % a.den));
		if (a.out_len) a.out_len[s] = (int)n_out;
	}
}

__global__ void fill_pos_kernel(int2 *pos, int first, int count) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) pos[first + i] = make_int2(0, 0);
}

} // namespace

struct mi_resampler {
	mi_ctx *ctx = nullptr;
	int nstreams = 0;
	uint32_t in_rate = 0, out_rate = 0;
	int quality = 3;
	Design d;
	int hist_stride = 0;
	int16_t *d_hist = nullptr;
	int2 *d_pos = nullptr;
	float *d_table = nullptr;
	bool phase_zero = true; // every stream's (last_sample, frac) is (0, 0): all blocks so far were whole output periods
	// what the kernels read: the direct table, or the per-phase blend of the interpolated one
	const std::vector<float> &dev_table() const { return d.phase_table.empty() ? d.table : d.phase_table; }
	bool dev_direct() const { return d.direct || !d.phase_table.empty(); }
};

void mi_resampler_view(mi_resampler *r, ResamplerView *v) {
	*v = ResamplerView();
	if (!r) return;
	v->hist = r->d_hist;
	v->table = r->d_table;
	v->hist_stride = r->hist_stride;
	v->den = (int)r->d.den;
	v->filt = (int)r->d.filt_len;
	v->nstreams = r->nstreams;
	v->device = r->ctx->device;
	v->ok = r->d.num == 1 && r->d.den > 1 && r->dev_direct() && r->d.phase_table.empty() && r->d.filt_len == 48 && r->phase_zero;
}

template <int DEN, int FILT, int R>
static int launch_up(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                     int out_stride, int32_t *d_out_len, const uint8_t *d_run, bool *done) {
	*done = false;
	const int tiles = mi::ceil_div(in_len, R);
	const int xn = ((FILT - 1 + in_len + R + 1) + 3) & ~3;
	const size_t lds = (size_t)xn * sizeof(float) + (size_t)((in_len * DEN + 7) & ~7) * sizeof(int16_t) +
	                   (size_t)DEN * FILT * sizeof(float); // window, output staging, polyphase table
	const int nq = (r->hist_stride >> 2) + (in_len >> 2);
	// 8-byte row loads (at most two per lane) and 16-byte output stores; any other layout takes the generic kernel
	if (lds > 48 * 1024 || ((in_len | in_stride) & 3) != 0 || nq > 128 || (reinterpret_cast<uintptr_t>(d_in) & 7) != 0 ||
	    (((in_len * DEN) | out_stride) & 7) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15) != 0)
		return MI_OK;
	UpArgs a;
	a.in = d_in;
	a.out = d_out;
	a.out_len = d_out_len;
	a.hist = r->d_hist;
	a.table = r->d_table;
	a.run = d_run;
	a.in_len = in_len;
	a.in_stride = in_stride;
	a.out_stride = out_stride;
	a.hist_stride = r->hist_stride;
	a.nstreams = r->nstreams;
	a.tiles = tiles;
	// persistent waves: enough to fill every SIMD, each walking the same number of streams
	static const int waves_per_cu = [] {
		const char *e = getenv("MSMI355X_RESAMPLE_WAVES_PER_CU");
		const int v = e ? atoi(e) : 0;
		return v > 0 ? v : 16;
	}();
	const int max_waves = (r->ctx->cu_count > 0 ? r->ctx->cu_count : 256) * waves_per_cu;
	const int per_wave = mi::ceil_div(r->nstreams, max_waves);
	const int nwaves = mi::ceil_div(r->nstreams, per_wave);
	const int wpw = waves_per_workgroup(r->nstreams);
	const int grid = mi::ceil_div(nwaves, wpw);
	a.lds_per_wave = (int)((lds + 15) & ~(size_t)15);
	if (DEN * tiles <= 64 && nq <= 64)
		hipLaunchKernelGGL((resample_up_kernel<DEN, FILT, R, false, false>), dim3(grid), dim3(64 * wpw),
		                   (size_t)a.lds_per_wave * wpw, r->ctx->stream, a);
	else
		hipLaunchKernelGGL((resample_up_kernel<DEN, FILT, R, true, true>), dim3(grid), dim3(64 * wpw),
		                   (size_t)a.lds_per_wave * wpw, r->ctx->stream, a);
	MI_LAUNCH_CHECK();
	*done = true;
	return MI_OK;
}

template <int NUM, int FILT, int R>
static int launch_down(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out, int out_stride,
                       int32_t *d_out_len, const uint8_t *d_run, bool *done) {
	*done = false;
	const int out_len = in_len / NUM;
	const int nq = (r->hist_stride >> 2) + (in_len >> 2);
	if ((in_len % NUM) != 0 || ((in_len | in_stride) & 3) != 0 || nq > 192 || ((out_len | out_stride) & 7) != 0 ||
	    (reinterpret_cast<uintptr_t>(d_in) & 7) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15) != 0 || !r->phase_zero)
		return MI_OK; // the generic kernel takes every other layout (and any stream state off the phase grid)
	DownArgs a;
	a.in = d_in;
	a.out = d_out;
	a.out_len = d_out_len;
	a.hist = r->d_hist;
	a.table = r->d_table;
	a.run = d_run;
	a.in_len = in_len;
	a.in_stride = in_stride;
	a.out_stride = out_stride;
	a.hist_stride = r->hist_stride;
	a.nstreams = r->nstreams;
	a.tiles = mi::ceil_div(out_len, R);
	const int xlen = NUM * FILT - 1 + in_len;
	a.plen = (mi::ceil_div(xlen, NUM) + R + 8 + 3) & ~3;
	const size_t lds = ((size_t)NUM * a.plen + (size_t)a.tiles * R * NUM + (size_t)NUM * FILT) * sizeof(float);
	if (lds > 30 * 1024) return MI_OK;
	a.lds_per_wave = (int)((lds + 15) & ~(size_t)15);
	const int max_waves = (r->ctx->cu_count > 0 ? r->ctx->cu_count : 256) * 16;
	const int per_wave = mi::ceil_div(r->nstreams, max_waves);
	const int nwaves = mi::ceil_div(r->nstreams, per_wave);
	const int wpw = waves_per_workgroup(r->nstreams);
	hipLaunchKernelGGL((resample_down_kernel<NUM, FILT, R>), dim3(mi::ceil_div(nwaves, wpw)), dim3(64 * wpw),
	                   (size_t)a.lds_per_wave * wpw, r->ctx->stream, a);
	MI_LAUNCH_CHECK();
	*done = true;
	return MI_OK;
}

template <int L, int M, int FT, int R>
static int launch_ratio(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out, int out_stride,
                        int32_t *d_out_len, const uint8_t *d_run, bool *done) {
	*done = false;
	const int periods = in_len / M, out_len = periods * L;
	const int nq = (r->hist_stride >> 2) + (in_len >> 2);
	if ((in_len % M) != 0 || ((in_len | in_stride) & 3) != 0 || nq > 192 || ((out_len | out_stride) & 7) != 0 ||
	    (reinterpret_cast<uintptr_t>(d_in) & 7) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15) != 0 || !r->phase_zero)
		return MI_OK;
	RatioArgs a;
	a.in = d_in;
	a.out = d_out;
	a.out_len = d_out_len;
	a.hist = r->d_hist;
	a.table = r->d_table;
	a.run = d_run;
	a.in_len = in_len;
	a.in_stride = in_stride;
	a.out_stride = out_stride;
	a.hist_stride = r->hist_stride;
	a.nstreams = r->nstreams;
	a.tiles = mi::ceil_div(periods, R);
	const int xlen = FT * M - 1 + in_len;
	a.plen = (mi::ceil_div(xlen, M) + R + 8 + 3) & ~3;
	const size_t lds = ((size_t)2 * M * a.plen + (size_t)a.tiles * R * L * M + (size_t)L * M * FT) * sizeof(float);
	if (lds > 30 * 1024) return MI_OK;
	a.lds_per_wave = (int)((lds + 15) & ~(size_t)15);
	const int max_waves = (r->ctx->cu_count > 0 ? r->ctx->cu_count : 256) * 16;
	const int per_wave = mi::ceil_div(r->nstreams, max_waves);
	const int nwaves = mi::ceil_div(r->nstreams, per_wave);
	const int wpw = waves_per_workgroup(r->nstreams);
	hipLaunchKernelGGL((resample_ratio_kernel<L, M, FT, R>), dim3(mi::ceil_div(nwaves, wpw)), dim3(64 * wpw),
	                   (size_t)a.lds_per_wave * wpw, r->ctx->stream, a);
	MI_LAUNCH_CHECK();
	*done = true;
	return MI_OK;
}

extern "C" {

int mi_resampler_create(mi_ctx *ctx, int nstreams, uint32_t in_rate, uint32_t out_rate, int quality,
                        mi_resampler **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && in_rate > 0 && out_rate > 0);
	*out = nullptr;
	if (in_rate == out_rate) {
		mi::set_error("equal rates are a pass-through in the filter (msresample.c:126-135), no batch needed");
		return MI_ENOTSUP;
	}
	mi_resampler *r = new mi_resampler();
	r->ctx = ctx;
	r->nstreams = nstreams;
	r->in_rate = in_rate;
	r->out_rate = out_rate;
	r->quality = quality;
	if (!design_filter(in_rate, out_rate, quality, r->d)) {
		mi::set_error("resampler quality %d not supported (msresample.c uses 3)", quality);
		delete r;
		return MI_ENOTSUP;
	}
	if (ctx->activate() != MI_OK) {
		delete r;
		return MI_ENODEV;
	}
	r->hist_stride = (int)mi::round_up(r->d.filt_len - 1, 8);
	const size_t hb = (size_t)nstreams * r->hist_stride * sizeof(int16_t);
	if (hipMalloc((void **)&r->d_hist, hb) != hipSuccess ||
	    hipMalloc((void **)&r->d_pos, (size_t)nstreams * sizeof(int2)) != hipSuccess ||
	    hipMalloc((void **)&r->d_table, r->dev_table().size() * sizeof(float)) != hipSuccess) {
		mi::set_error("hipMalloc failed for resampler state");
		mi_resampler_destroy(r);
		return MI_ENOMEM;
	}
	if (hipMemsetAsync(r->d_hist, 0, hb, ctx->stream) != hipSuccess ||
	    hipMemsetAsync(r->d_pos, 0, (size_t)nstreams * sizeof(int2), ctx->stream) != hipSuccess ||
	    hipMemcpyAsync(r->d_table, r->dev_table().data(), r->dev_table().size() * sizeof(float), hipMemcpyHostToDevice,
	                   ctx->stream) != hipSuccess ||
	    hipStreamSynchronize(ctx->stream) != hipSuccess) {
		mi::set_error("resampler state upload failed");
		mi_resampler_destroy(r);
		return MI_ENODEV;
	}
	*out = r;
	return MI_OK;
}

void mi_resampler_destroy(mi_resampler *r) {
	if (!r) return;
	(void)hipSetDevice(r->ctx->device);
	if (r->d_hist) (void)hipFree(r->d_hist);
	if (r->d_pos) (void)hipFree(r->d_pos);
	if (r->d_table) (void)hipFree(r->d_table);
	delete r;
}

int mi_resampler_reset(mi_resampler *r, int first, int count) {
	MI_CHECK_ARG(r && first >= 0 && count >= 0 && first + count <= r->nstreams);
	if (count == 0) return MI_OK;
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipMemsetAsync(r->d_hist + (size_t)first * r->hist_stride, 0,
	                      (size_t)count * r->hist_stride * sizeof(int16_t), r->ctx->stream));
	hipLaunchKernelGGL(fill_pos_kernel, dim3(mi::ceil_div(count, 256)), dim3(256), 0, r->ctx->stream, r->d_pos,
	                   first, count);
	MI_LAUNCH_CHECK();
	if (first == 0 && count == r->nstreams) r->phase_zero = true;
	return MI_OK;
}

// One stream's running state -- (last_sample, frac) and the history row -- to and from host memory: what a speex resampler handle
// carries from call to call (msresample.c keeps its handle across a detach of the graph; the plugin's fused legs move the state from
// bank slot to bank slot with it).  Both wait for the stream.
int mi_resampler_state_bytes(const mi_resampler *r) { return r ? (int)(sizeof(int2) + (size_t)r->hist_stride * sizeof(int16_t)) : MI_EINVAL; }
int mi_resampler_get_state(mi_resampler *r, int stream, void *h_state, size_t cap) {
	MI_CHECK_ARG(r && h_state && stream >= 0 && stream < r->nstreams && cap >= (size_t)mi_resampler_state_bytes(r));
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;
	uint8_t *dst = static_cast<uint8_t *>(h_state);
	MI_HIP(hipStreamSynchronize(r->ctx->stream));
	MI_HIP(hipMemcpy(dst, r->d_pos + stream, sizeof(int2), hipMemcpyDeviceToHost));
	MI_HIP(hipMemcpy(dst + sizeof(int2), r->d_hist + (size_t)stream * r->hist_stride, (size_t)r->hist_stride * sizeof(int16_t), hipMemcpyDeviceToHost));
	return MI_OK;
}
int mi_resampler_set_state(mi_resampler *r, int stream, const void *h_state, size_t bytes) {
	MI_CHECK_ARG(r && h_state && stream >= 0 && stream < r->nstreams && bytes == (size_t)mi_resampler_state_bytes(r));
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;
	const uint8_t *src = static_cast<const uint8_t *>(h_state);
	int2 pos;
	memcpy(&pos, src, sizeof(pos));
	MI_HIP(hipStreamSynchronize(r->ctx->stream));
	MI_HIP(hipMemcpy(r->d_pos + stream, src, sizeof(int2), hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(r->d_hist + (size_t)stream * r->hist_stride, src + sizeof(int2), (size_t)r->hist_stride * sizeof(int16_t), hipMemcpyHostToDevice));
	if (pos.x != 0 || pos.y != 0) r->phase_zero = false;
	return MI_OK;
}

// the same for streams [first, first + count), mi_resampler_state_bytes() each, back to back: one round trip for a conference's members
int mi_resampler_get_states(mi_resampler *r, int first, int count, void *h_states, size_t cap) {
	const size_t each = r ? (size_t)mi_resampler_state_bytes(r) : 0;
	MI_CHECK_ARG(r && h_states && first >= 0 && count >= 0 && first + count <= r->nstreams && cap >= each * (size_t)count);
	if (count == 0) return MI_OK;
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;
	std::vector<int2> pos((size_t)count);
	std::vector<int16_t> hist((size_t)count * r->hist_stride);
	MI_HIP(hipStreamSynchronize(r->ctx->stream));
	MI_HIP(hipMemcpy(pos.data(), r->d_pos + first, pos.size() * sizeof(int2), hipMemcpyDeviceToHost));
	MI_HIP(hipMemcpy(hist.data(), r->d_hist + (size_t)first * r->hist_stride, hist.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
	uint8_t *dst = static_cast<uint8_t *>(h_states);
	for (int k = 0; k < count; ++k) {
		memcpy(dst + (size_t)k * each, &pos[(size_t)k], sizeof(int2));
		memcpy(dst + (size_t)k * each + sizeof(int2), hist.data() + (size_t)k * r->hist_stride, (size_t)r->hist_stride * sizeof(int16_t));
	}
	return MI_OK;
}
int mi_resampler_set_states(mi_resampler *r, int first, int count, const void *h_states, size_t bytes) {
	const size_t each = r ? (size_t)mi_resampler_state_bytes(r) : 0;
	MI_CHECK_ARG(r && h_states && first >= 0 && count >= 0 && first + count <= r->nstreams && bytes == each * (size_t)count);
	if (count == 0) return MI_OK;
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;
	std::vector<int2> pos((size_t)count);
	std::vector<int16_t> hist((size_t)count * r->hist_stride);
	const uint8_t *src = static_cast<const uint8_t *>(h_states);
	for (int k = 0; k < count; ++k) {
		memcpy(&pos[(size_t)k], src + (size_t)k * each, sizeof(int2));
		memcpy(hist.data() + (size_t)k * r->hist_stride, src + (size_t)k * each + sizeof(int2), (size_t)r->hist_stride * sizeof(int16_t));
		if (pos[(size_t)k].x != 0 || pos[(size_t)k].y != 0) r->phase_zero = false;
	}
	MI_HIP(hipStreamSynchronize(r->ctx->stream));
	MI_HIP(hipMemcpy(r->d_pos + first, pos.data(), pos.size() * sizeof(int2), hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(r->d_hist + (size_t)first * r->hist_stride, hist.data(), hist.size() * sizeof(int16_t), hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_resampler_out_capacity(const mi_resampler *r, int in_len) {
	if (!r || in_len < 0) return MI_EINVAL;
	return (int)((((uint32_t)in_len * r->out_rate) / r->in_rate) + 1);
}

int mi_resampler_info(const mi_resampler *r, int *filt_len, int *den_rate, int *num_rate, int *direct) {
	MI_CHECK_ARG(r != nullptr);
	if (filt_len) *filt_len = (int)r->d.filt_len;
	if (den_rate) *den_rate = (int)r->d.den;
	if (num_rate) *num_rate = (int)r->d.num;
	if (direct) *direct = r->d.direct;
	return MI_OK;
}

int mi_resampler_get_table(const mi_resampler *r, float *h_dst, int cap) {
	if (!r) return MI_EINVAL;
	const int n = (int)r->d.table.size();
	if (h_dst && cap >= n) memcpy(h_dst, r->d.table.data(), sizeof(float) * (size_t)n);
	return n;
}

int mi_resampler_process(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                         int out_stride, int32_t *d_out_len) {
	return mi_resampler_process_masked(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, nullptr);
}

int mi_resampler_process_masked(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                                int out_stride, int32_t *d_out_len, const uint8_t *d_run) {
	MI_CHECK_ARG(r && d_in && d_out && in_len > 0 && in_stride >= in_len);
	const int cap = mi_resampler_out_capacity(r, in_len);
	// the reference allocates cap samples (msresample.c:154); integer up-sampling
	// produces exactly in_len*den, so cap-1 is enough there.
	const int need = (r->d.num == 1) ? in_len * (int)r->d.den : cap;
	MI_CHECK_ARG(out_stride >= need);
	if (r->ctx->activate() != MI_OK) return MI_ENODEV;

	if (r->d.num == 1 && r->d.direct && r->d.filt_len == 48) {
		bool done = false;
		int rc = MI_OK;
		switch (r->d.den) {
			case 2: rc = launch_up<2, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 3: rc = launch_up<3, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 4: rc = launch_up<4, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 6: rc = launch_up<6, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			default: break;
		}
		if (rc != MI_OK) return rc;
		if (done) return MI_OK;
	}

	if (r->d.den == 1 && r->d.direct && r->d.filt_len == 48 * r->d.num) {
		bool done = false;
		int rc = MI_OK;
		switch (r->d.num) {
			case 2: rc = launch_down<2, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 3: rc = launch_down<3, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 4: rc = launch_down<4, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			case 6: rc = launch_down<6, 48, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done); break;
			default: break;
		}
		if (rc != MI_OK) return rc;
		if (done) return MI_OK;
		// a block that is not a whole number of output periods leaves (last_sample, frac) off zero for good
		if (in_len % (int)r->d.num) r->phase_zero = false;
	}

	if (r->d.direct && ((r->d.den == 3 && r->d.num == 2 && r->d.filt_len == 48) || (r->d.den == 2 && r->d.num == 3 && r->d.filt_len == 72))) {
		bool done = false;
		const int rc = (r->d.den == 3) ? launch_ratio<3, 2, 24, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done)
		                               : launch_ratio<2, 3, 24, 8>(r, d_in, in_len, in_stride, d_out, out_stride, d_out_len, d_run, &done);
		if (rc != MI_OK) return rc;
		if (done) return MI_OK;
		if (in_len % (int)r->d.num) r->phase_zero = false;
	}

	GenArgs a;
	a.in = d_in;
	a.out = d_out;
	a.out_len = d_out_len;
	a.hist = r->d_hist;
	a.pos = r->d_pos;
	a.table = r->d_table;
	a.run = d_run;
	a.table_len = (int)r->dev_table().size();
	a.in_len = in_len;
	a.in_stride = in_stride;
	a.out_stride = out_stride;
	a.out_cap = out_stride < cap ? out_stride : cap;
	a.hist_stride = r->hist_stride;
	a.nstreams = r->nstreams;
	a.filt_len = (int)r->d.filt_len;
	a.num = (int)r->d.num;
	a.den = (int)r->d.den;
	a.oversample = (int)r->d.oversample;
	a.direct = r->dev_direct() ? 1 : 0;
	const size_t xbytes = (size_t)((a.filt_len - 1 + in_len + 3) & ~3) * sizeof(float);
	const size_t tbytes = (size_t)a.table_len * sizeof(float);
	a.table_in_lds = (xbytes + tbytes <= 60 * 1024) ? 1 : 0;
	const size_t lds = xbytes + (a.table_in_lds ? tbytes : 0);
	if (lds > 64 * 1024) {
		mi::set_error("input block of %d samples too large for the resampler kernel", in_len);
		return MI_ENOTSUP;
	}
	hipLaunchKernelGGL(resample_generic_kernel, dim3(r->nstreams), dim3(256), lds, r->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_resampler_process_host(mi_resampler *r, const int16_t *h_in, int in_len, int in_stride, int16_t *h_out,
                              int out_stride, int32_t *h_out_len) {
	MI_CHECK_ARG(r && h_in && h_out);
	mi_ctx *c = r->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t ib = (size_t)r->nstreams * in_stride * sizeof(int16_t);
	const size_t ob = (size_t)r->nstreams * out_stride * sizeof(int16_t);
	void *din, *dout, *dlen;
	int rc;
	if ((rc = c->ensure_scratch(0, ib, &din)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, ob, &dout)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(2, (size_t)r->nstreams * sizeof(int32_t), &dlen)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(din, h_in, ib, hipMemcpyHostToDevice, c->stream));
	MI_HIP(hipMemsetAsync(dout, 0, ob, c->stream));
	rc = mi_resampler_process(r, (const int16_t *)din, in_len, in_stride, (int16_t *)dout, out_stride,
	                          (int32_t *)dlen);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_out, dout, ob, hipMemcpyDeviceToHost, c->stream));
	if (h_out_len)
		MI_HIP(hipMemcpyAsync(h_out_len, dlen, (size_t)r->nstreams * sizeof(int32_t), hipMemcpyDeviceToHost,
		                      c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_resample(reinterpret_cast<const void *>(&fill_pos_kernel));
