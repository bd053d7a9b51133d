/* oracle/resample.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 *
 * CPU restatement of the libspeexdsp resampler, float build, as driven by
 * /root/reference/src/audiofilters/msresample.c:102-115 (init, quality =
 * SPEEX_RESAMPLER_QUALITY_VOIP = 3) and :150-177 (speex_resampler_process_int
 * per input block).  libspeexdsp is a third-party dependency that is NOT in
 * /root/reference (CMakeLists.txt:207-209 find_package(SpeexDSP), no version
 * pin; configure.ac:567 speexdsp >= 1.2beta3); this follows the published
 * algorithm of speexdsp 1.2.x resample.c: Kaiser-windowed sinc, polyphase
 * "direct" table when filt_len*den_rate <= filt_len*oversample+8, otherwise an
 * oversampled table with 4-point cubic interpolation.  The generic-C inner
 * product order (j ascending, one accumulator) is used; SSE/NEON builds of
 * the library sum in a different order, hence the 1e-4 RMS tolerance.
 *
 * Only the KAISER8 qualities (3 = VOIP, 4) are restated; msresample.c never
 * asks for another one on x86.
 */
#include "ms2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Kaiser window, beta = 8, sampled at x = (i-1)/32; the last two entries are
 * the library's guard values.  (Entries 0..33 equal I0(8*sqrt(1-x^2))/I0(8)
 * to 8 decimals -- checked in tests/test_oracle_cpu.py.) */
static const double kaiser8_table[36] = {
    0.99635258, 1.00000000, 0.99635258, 0.98548012, 0.96759014, 0.94302200, 0.91223751, 0.87580811,
    0.83439927, 0.78875245, 0.73966538, 0.68797126, 0.63451750, 0.58014482, 0.52566725, 0.47185369,
    0.41941150, 0.36897272, 0.32108304, 0.27619388, 0.23465776, 0.19672670, 0.16255380, 0.13219758,
    0.10562887, 0.08273982, 0.06335451, 0.04724088, 0.03412321, 0.02369490, 0.01563093, 0.00959968,
    0.00527363, 0.00233883, 0.00050000, 0.00000000};
#define KAISER8_OVERSAMPLE 32

typedef struct {
	int base_length, oversample;
	float down_bw, up_bw;
} QualityRow;
/* rows 3 and 4 of the library's quality_map */
static const QualityRow kq3 = {48, 8, 0.895f, 0.917f};
static const QualityRow kq4 = {64, 8, 0.921f, 0.940f};

struct OrcResampler {
	uint32_t in_rate, out_rate, num_rate, den_rate;
	int quality;
	uint32_t filt_len, mem_alloc_size, buffer_size;
	int int_advance, frac_advance;
	float cutoff;
	uint32_t oversample;
	int direct;
	int32_t last_sample;
	uint32_t samp_frac_num;
	float *mem;
	float *sinc_table;
	uint32_t sinc_table_length;
};

static double window_value(float x) {
	float y, frac;
	double interp[4];
	int ind;
	y = x * KAISER8_OVERSAMPLE;
	ind = (int)floor(y);
	frac = (y - ind);
	interp[3] = -0.1666666667 * frac + 0.1666666667 * (frac * frac * frac);
	interp[2] = frac + 0.5 * (frac * frac) - 0.5 * (frac * frac * frac);
	interp[0] = -0.3333333333 * frac + 0.5 * (frac * frac) - 0.1666666667 * (frac * frac * frac);
	interp[1] = 1.f - interp[3] - interp[2] - interp[0];
	return interp[0] * kaiser8_table[ind] + interp[1] * kaiser8_table[ind + 1] +
	       interp[2] * kaiser8_table[ind + 2] + interp[3] * kaiser8_table[ind + 3];
}

static float sinc_tap(float cutoff, float x, int N) {
	float xx = x * cutoff;
	if (fabs(x) < 1e-6) return cutoff;
	else if (fabs(x) > .5 * N) return 0;
	return (float)(cutoff * sin(M_PI * xx) / (M_PI * xx) * window_value((float)fabs(2. * x / N)));
}

static uint32_t gcd_u32(uint32_t a, uint32_t b) {
	while (b != 0) {
		uint32_t t = a;
		a = b;
		b = t % b;
	}
	return a;
}

static void build_filter(OrcResampler *st) {
	const QualityRow *q = (st->quality == 4) ? &kq4 : &kq3;
	uint32_t i;
	st->int_advance = (int)(st->num_rate / st->den_rate);
	st->frac_advance = (int)(st->num_rate % st->den_rate);
	st->oversample = (uint32_t)q->oversample;
	st->filt_len = (uint32_t)q->base_length;
	if (st->num_rate > st->den_rate) {
		/* down-sampling: stretch the filter, narrow the cutoff */
		st->cutoff = q->down_bw * st->den_rate / st->num_rate;
		st->filt_len = st->filt_len * st->num_rate / st->den_rate;
		st->filt_len = ((st->filt_len - 1) & (~0x7u)) + 8;
		if (2 * st->den_rate < st->num_rate) st->oversample >>= 1;
		if (4 * st->den_rate < st->num_rate) st->oversample >>= 1;
		if (8 * st->den_rate < st->num_rate) st->oversample >>= 1;
		if (16 * st->den_rate < st->num_rate) st->oversample >>= 1;
		if (st->oversample < 1) st->oversample = 1;
	} else {
		st->cutoff = q->up_bw;
	}
	st->direct = st->filt_len * st->den_rate <= st->filt_len * st->oversample + 8;
	if (st->direct) {
		st->sinc_table_length = st->filt_len * st->den_rate;
		st->sinc_table = (float *)malloc(sizeof(float) * st->sinc_table_length);
		for (i = 0; i < st->den_rate; i++) {
			int32_t j;
			for (j = 0; j < (int32_t)st->filt_len; j++) {
				st->sinc_table[i * st->filt_len + j] =
				    sinc_tap(st->cutoff, ((j - (int32_t)st->filt_len / 2 + 1) - ((float)i) / st->den_rate),
				             (int)st->filt_len);
			}
		}
	} else {
		int32_t k;
		st->sinc_table_length = st->filt_len * st->oversample + 8;
		st->sinc_table = (float *)malloc(sizeof(float) * st->sinc_table_length);
		for (k = -4; k < (int32_t)(st->oversample * st->filt_len + 4); k++)
			st->sinc_table[k + 4] =
			    sinc_tap(st->cutoff, (k / (float)st->oversample - st->filt_len / 2), (int)st->filt_len);
	}
	st->buffer_size = 160;
	st->mem_alloc_size = st->filt_len - 1 + st->buffer_size;
	st->mem = (float *)calloc(st->mem_alloc_size, sizeof(float));
}

OrcResampler *orc_resampler_new(uint32_t in_rate, uint32_t out_rate, int quality) {
	OrcResampler *st;
	uint32_t g;
	if (quality != 3 && quality != 4) return NULL;
	if (in_rate == 0 || out_rate == 0) return NULL;
	st = (OrcResampler *)calloc(1, sizeof(*st));
	st->in_rate = in_rate;
	st->out_rate = out_rate;
	st->quality = quality;
	g = gcd_u32(in_rate, out_rate);
	st->num_rate = in_rate / g;
	st->den_rate = out_rate / g;
	build_filter(st);
	return st;
}

void orc_resampler_free(OrcResampler *r) {
	if (!r) return;
	free(r->mem);
	free(r->sinc_table);
	free(r);
}

static void cubic_coef(float frac, float interp[4]) {
	interp[0] = -0.16667f * frac + 0.16667f * frac * frac * frac;
	interp[1] = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
	interp[3] = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
	interp[2] = 1. - interp[0] - interp[1] - interp[3];
}

/* one pass over `mem` (history + *in_len fresh samples): returns outputs made */
static int run_native(OrcResampler *st, uint32_t *in_len, float *out, uint32_t *out_len) {
	const int N = (int)st->filt_len;
	int out_sample = 0;
	int32_t last_sample = st->last_sample;
	uint32_t frac_num = st->samp_frac_num;
	const float *in = st->mem;
	uint32_t ilen;
	int j;

	while (!(last_sample >= (int32_t)*in_len || out_sample >= (int32_t)*out_len)) {
		const float *iptr = &in[last_sample];
		float sum;
		if (st->direct) {
			const float *sinct = &st->sinc_table[frac_num * N];
			sum = 0;
			for (j = 0; j < N; j++) sum += sinct[j] * iptr[j];
		} else {
			const int offset = (int)(frac_num * st->oversample / st->den_rate);
			const float frac = ((float)((frac_num * st->oversample) % st->den_rate)) / st->den_rate;
			float interp[4];
			float accum[4] = {0, 0, 0, 0};
			for (j = 0; j < N; j++) {
				const float cur = iptr[j];
				accum[0] += cur * st->sinc_table[4 + (j + 1) * st->oversample - offset - 2];
				accum[1] += cur * st->sinc_table[4 + (j + 1) * st->oversample - offset - 1];
				accum[2] += cur * st->sinc_table[4 + (j + 1) * st->oversample - offset];
				accum[3] += cur * st->sinc_table[4 + (j + 1) * st->oversample - offset + 1];
			}
			cubic_coef(frac, interp);
			sum = interp[0] * accum[0] + interp[1] * accum[1] + interp[2] * accum[2] + interp[3] * accum[3];
		}
		out[out_sample++] = sum;
		last_sample += st->int_advance;
		frac_num += (uint32_t)st->frac_advance;
		if (frac_num >= st->den_rate) {
			frac_num -= st->den_rate;
			last_sample++;
		}
	}
	st->last_sample = last_sample;
	st->samp_frac_num = frac_num;

	if (st->last_sample < (int32_t)*in_len) *in_len = (uint32_t)st->last_sample;
	*out_len = (uint32_t)out_sample;
	st->last_sample -= (int32_t)*in_len;
	ilen = *in_len;
	for (j = 0; j < N - 1; ++j) st->mem[j] = st->mem[j + ilen];
	return out_sample;
}

static int16_t word2int(float x) {
	return (int16_t)(x < -32767.5f ? -32768 : (x > 32766.5f ? 32767 : floor(.5 + x)));
}

void orc_resampler_process(OrcResampler *st, const int16_t *in, uint32_t *in_len, int16_t *out,
                           uint32_t *out_len) {
	uint32_t ilen = *in_len, olen = *out_len, j;
	const uint32_t xlen = st->mem_alloc_size - (st->filt_len - 1);
	float ystack[1024];
	while (ilen && olen) {
		uint32_t ichunk = (ilen > xlen) ? xlen : ilen;
		uint32_t ochunk = (olen > 1024) ? 1024 : olen;
		for (j = 0; j < ichunk; ++j) st->mem[j + st->filt_len - 1] = in[j];
		run_native(st, &ichunk, ystack, &ochunk);
		for (j = 0; j < ochunk; ++j) out[j] = word2int(ystack[j]);
		ilen -= ichunk;
		olen -= ochunk;
		out += ochunk;
		in += ichunk;
	}
	*in_len -= ilen;
	*out_len -= olen;
}

int orc_resampler_filt_len(const OrcResampler *r) { return (int)r->filt_len; }
int orc_resampler_den_rate(const OrcResampler *r) { return (int)r->den_rate; }
int orc_resampler_num_rate(const OrcResampler *r) { return (int)r->num_rate; }
int orc_resampler_is_direct(const OrcResampler *r) { return r->direct; }
int orc_resampler_table(const OrcResampler *r, float *dst, int cap) {
	int n = (int)r->sinc_table_length;
	if (dst && cap >= n) memcpy(dst, r->sinc_table, sizeof(float) * (size_t)n);
	return n;
}

/* msresample.c:151-152 */
uint32_t orc_msresample_outcap(uint32_t inlen, uint32_t in_rate, uint32_t out_rate) {
	return (uint32_t)(((inlen * out_rate) / in_rate) + 1);
}
