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
//       16k->48k, 8k->48k): each lane owns one polyphase row (FILT taps in
//       VGPRs) and R consecutive input positions; the input window slides
//       through registers, so one LDS read feeds R FMAs.  Input/history are
//       staged in LDS as float with an (i + i/8) skew that makes the stride-R
//       window reads bank-conflict free; outputs are staged in LDS and leave
//       as 16-byte coalesced stores.
//   resample_generic_kernel         any other ratio (direct table or the
//       oversampled table + 4-point cubic interpolation), one block per stream.
//
// HBM traffic per stream-tick (16k->48k): 320 B in + 960 B out (+ 96 B history
// read + 96 B written) -- the kernel is HBM/launch bound, not VALU bound.
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
	}
	return true;
}

// ------------------------------------------------------------------- kernels
// WORD2INT of the library: round half up in double, clamp to int16.
__device__ __forceinline__ int16_t word2int(float x) {
	if (x < -32767.5f) return (int16_t)-32768;
	if (x > 32766.5f) return (int16_t)32767;
	return (int16_t)(int)floor(0.5 + (double)x);
}

__device__ __forceinline__ int skew(int i) { return i + (i >> 3); }

struct UpArgs {
	const int16_t *in;
	int16_t *out;
	int32_t *out_len;
	int16_t *hist;
	const float *table;
	const uint8_t *run;
	int in_len, in_stride, out_stride, hist_stride, nstreams;
	int tiles; // ceil(in_len / R)
	int spb;   // streams per block
	int xs;    // floats of LDS per stream (skewed)
};

template <int DEN, int FILT, int R>
__global__ __launch_bounds__(256) void resample_up_kernel(UpArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *xbuf = reinterpret_cast<float *>(smem);
	const int out_per_stream = a.in_len * DEN;
	const int ostage_stride = (out_per_stream + 7) & ~7;
	int16_t *obuf = reinterpret_cast<int16_t *>(smem + (size_t)a.spb * a.xs * sizeof(float));

	const int tid = threadIdx.x;
	const int s0 = blockIdx.x * a.spb;
	const int nloc = min(a.spb, a.nstreams - s0);
	constexpr int HIST = FILT - 1;
	auto live = [&](int sl) -> bool { return a.run == nullptr || a.run[s0 + sl] != 0; };

	// ---- stage history + input as float (coalesced 8-byte loads)
	{
		const int hq = a.hist_stride >> 2;
		for (int i = tid; i < nloc * hq; i += 256) {
			const int sl = i / hq, q = i - sl * hq;
			const short4 v = *reinterpret_cast<const short4 *>(a.hist + (size_t)(s0 + sl) * a.hist_stride + 4 * q);
			float *x = xbuf + sl * a.xs;
			const int b = 4 * q;
			if (b + 0 < HIST) x[skew(b + 0)] = (float)v.x;
			if (b + 1 < HIST) x[skew(b + 1)] = (float)v.y;
			if (b + 2 < HIST) x[skew(b + 2)] = (float)v.z;
			if (b + 3 < HIST) x[skew(b + 3)] = (float)v.w;
		}
		if (((a.in_len | a.in_stride) & 3) == 0) {
			const int iq = a.in_len >> 2;
			for (int i = tid; i < nloc * iq; i += 256) {
				const int sl = i / iq, q = i - sl * iq;
				const short4 v = *reinterpret_cast<const short4 *>(a.in + (size_t)(s0 + sl) * a.in_stride + 4 * q);
				float *x = xbuf + sl * a.xs;
				const int b = HIST + 4 * q;
				x[skew(b + 0)] = (float)v.x;
				x[skew(b + 1)] = (float)v.y;
				x[skew(b + 2)] = (float)v.z;
				x[skew(b + 3)] = (float)v.w;
			}
		} else {
			for (int i = tid; i < nloc * a.in_len; i += 256) {
				const int sl = i / a.in_len, q = i - sl * a.in_len;
				xbuf[sl * a.xs + skew(HIST + q)] = (float)a.in[(size_t)(s0 + sl) * a.in_stride + q];
			}
		}
		// zero the slack the last (partial) tile reads
		const int used = HIST + a.in_len;
		for (int i = tid; i < nloc * R; i += 256) {
			const int sl = i / R, q = i - sl * R;
			xbuf[sl * a.xs + skew(used + q)] = 0.f;
		}
	}
	__syncthreads();

	// ---- compute: lane = (stream, tile, phase)
	const int lps = DEN * a.tiles;
	const int sl = tid / lps;
	if (sl < nloc && live(sl)) {
		const int rem = tid - sl * lps;
		const int tile = rem / DEN, p = rem - tile * DEN;
		const int m0 = tile * R;
		float t[FILT];
		const float4 *tp = reinterpret_cast<const float4 *>(a.table + p * FILT);
#pragma unroll
		for (int j = 0; j < FILT / 4; ++j) {
			const float4 v = tp[j];
			t[4 * j + 0] = v.x;
			t[4 * j + 1] = v.y;
			t[4 * j + 2] = v.z;
			t[4 * j + 3] = v.w;
		}
		const float *x = xbuf + sl * a.xs;
		float w[R], acc[R];
#pragma unroll
		for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
		for (int r = 0; r < R - 1; ++r) w[r] = x[skew(m0 + r)];
#pragma unroll
		for (int j = 0; j < FILT; ++j) {
			w[(j + R - 1) % R] = x[skew(m0 + R - 1 + j)];
#pragma unroll
			for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(t[j], w[(j + r) % R], acc[r]);
		}
		int16_t *o = obuf + sl * ostage_stride;
#pragma unroll
		for (int r = 0; r < R; ++r)
			if (m0 + r < a.in_len) o[(m0 + r) * DEN + p] = word2int(acc[r]);
	}
	__syncthreads();

	// ---- outputs: 16-byte coalesced stores when the layout allows
	if (((out_per_stream | a.out_stride) & 7) == 0) {
		const int oq = out_per_stream >> 3;
		for (int i = tid; i < nloc * oq; i += 256) {
			const int s = i / oq, q = i - s * oq;
			if (!live(s)) continue;
			const uint4 v = *reinterpret_cast<const uint4 *>(obuf + s * ostage_stride + 8 * q);
			*reinterpret_cast<uint4 *>(a.out + (size_t)(s0 + s) * a.out_stride + 8 * q) = v;
		}
	} else {
		for (int i = tid; i < nloc * out_per_stream; i += 256) {
			const int s = i / out_per_stream, q = i - s * out_per_stream;
			if (!live(s)) continue;
			a.out[(size_t)(s0 + s) * a.out_stride + q] = obuf[s * ostage_stride + q];
		}
	}
	// ---- new history = last FILT-1 samples of (history ++ input)
	for (int i = tid; i < nloc * HIST; i += 256) {
		const int s = i / HIST, h = i - s * HIST;
		if (!live(s)) continue;
		a.hist[(size_t)(s0 + s) * a.hist_stride + h] = (int16_t)xbuf[s * a.xs + skew(a.in_len + h)];
	}
	if (a.out_len)
		for (int i = tid; i < nloc; i += 256) a.out_len[s0 + i] = live(i) ? out_per_stream : 0;
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
		o[k] = word2int(sum);
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
};

template <int DEN, int FILT, int R>
static int launch_up(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                     int out_stride, int32_t *d_out_len, const uint8_t *d_run, bool *done) {
	*done = false;
	const int tiles = mi::ceil_div(in_len, R);
	const int lps = DEN * tiles;
	if (lps > 256) return MI_OK;
	const int spb = 256 / lps;
	const int xn = FILT - 1 + in_len + R;
	const int xs = xn + (xn >> 3) + 1;
	const int ostage = (in_len * DEN + 7) & ~7;
	const size_t lds = (size_t)spb * xs * sizeof(float) + (size_t)spb * ostage * sizeof(int16_t);
	if (lds > 64 * 1024) return MI_OK;
	// keep the int16 staging area 16-byte aligned
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
	a.spb = spb;
	a.xs = (xs + 3) & ~3;
	const size_t lds2 = (size_t)spb * a.xs * sizeof(float) + (size_t)spb * ostage * sizeof(int16_t);
	const int grid = mi::ceil_div(r->nstreams, spb);
	hipLaunchKernelGGL((resample_up_kernel<DEN, FILT, R>), dim3(grid), dim3(256), lds2, r->ctx->stream, a);
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
	    hipMalloc((void **)&r->d_table, r->d.table.size() * sizeof(float)) != hipSuccess) {
		mi::set_error("hipMalloc failed for resampler state");
		mi_resampler_destroy(r);
		return MI_ENOMEM;
	}
	if (hipMemsetAsync(r->d_hist, 0, hb, ctx->stream) != hipSuccess ||
	    hipMemsetAsync(r->d_pos, 0, (size_t)nstreams * sizeof(int2), ctx->stream) != hipSuccess ||
	    hipMemcpyAsync(r->d_table, r->d.table.data(), r->d.table.size() * sizeof(float), hipMemcpyHostToDevice,
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
	MI_HIP(hipMemsetAsync(r->d_hist + (size_t)first * r->hist_stride, 0,
	                      (size_t)count * r->hist_stride * sizeof(int16_t), r->ctx->stream));
	hipLaunchKernelGGL(fill_pos_kernel, dim3(mi::ceil_div(count, 256)), dim3(256), 0, r->ctx->stream, r->d_pos,
	                   first, count);
	MI_LAUNCH_CHECK();
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

	GenArgs a;
	a.in = d_in;
	a.out = d_out;
	a.out_len = d_out_len;
	a.hist = r->d_hist;
	a.pos = r->d_pos;
	a.table = r->d_table;
	a.run = d_run;
	a.table_len = (int)r->d.table.size();
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
	a.direct = r->d.direct;
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
