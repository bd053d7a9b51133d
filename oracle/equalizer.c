/* oracle/equalizer.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 *
 * Float-build restatement of
 *   /root/reference/src/utils/kiss_fft.c   (butterflies :38-149, work :320-408,
 *                                           factor/alloc :412-475)
 *   /root/reference/src/utils/kiss_fftr.c  (alloc :40-81, fftr2 :175-259, fftri2 :261-296)
 *   /root/reference/src/utils/dsptools.c   (ms_fir_mem16 :253-268, ms_fft/ms_ifft :333-376)
 *   /root/reference/src/audiofilters/equalizer.c (state :45-86, gains :88-172,
 *                                           impulse response :184-237, run :263-269)
 * Compile with -ffp-contract=off (x86-64 reference build is unfused).
 */
#include "ms2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
	float r, i;
} cpx;

#define MAXFAC 32
typedef struct {
	int n, inverse;
	int fac[2 * MAXFAC]; /* p0,m0,p1,m1,... (kiss_fft.c:412-435) */
	cpx *tw;
} CFft;

static void cmul(cpx *m, const cpx *a, const cpx *b) { /* _kiss_fft_guts.h:109-113 */
	m->r = a->r * b->r - a->i * b->i;
	m->i = a->r * b->i + a->i * b->r;
}

static void cfft_init(CFft *st, int n, int inverse) {
	int i, p = 4, *f = st->fac, left = n;
	st->n = n;
	st->inverse = inverse;
	st->tw = (cpx *)malloc(sizeof(cpx) * (size_t)n);
	for (i = 0; i < n; ++i) { /* kiss_fft.c:464-471 */
		const double pi = 3.14159265358979323846264338327;
		double phase = (-2 * pi / n) * i;
		if (inverse) phase *= -1;
		st->tw[i].r = (float)cos(phase);
		st->tw[i].i = (float)sin(phase);
	}
	do { /* kf_factor: 4s first, then 2s, then odd primes */
		while (left % p) {
			switch (p) {
				case 4: p = 2; break;
				case 2: p = 3; break;
				default: p += 2; break;
			}
			if (p > 32000 || (int32_t)p * (int32_t)p > left) p = left;
		}
		left /= p;
		*f++ = p;
		*f++ = left;
	} while (left > 1);
}

/* kf_shuffle kiss_fft.c:292-318 */
static void shuffle(cpx *out, const cpx *f, size_t fstride, const int *fac) {
	const int p = fac[0], m = fac[1];
	int j;
	if (m == 1) {
		for (j = 0; j < p; j++) {
			out[j] = *f;
			f += fstride;
		}
	} else {
		for (j = 0; j < p; j++) {
			shuffle(out, f, fstride * p, fac + 2);
			f += fstride;
			out += m;
		}
	}
}

static void bfly2(cpx *base, size_t fstride, const CFft *st, int m, int N, int mm) {
	int i, j;
	for (i = 0; i < N; i++) {
		cpx *a = base + i * mm, *b = a + m;
		const cpx *tw = st->tw;
		for (j = 0; j < m; j++) {
			cpx t;
			cmul(&t, b, tw);
			tw += fstride;
			b->r = a->r - t.r;
			b->i = a->i - t.i;
			a->r += t.r;
			a->i += t.i;
			++a;
			++b;
		}
	}
}

static void bfly4(cpx *base, size_t fstride, const CFft *st, int m, int N, int mm) {
	const size_t m2 = 2 * (size_t)m, m3 = 3 * (size_t)m;
	int i, j;
	for (i = 0; i < N; i++) {
		cpx *F = base + i * mm;
		const cpx *t1 = st->tw, *t2 = st->tw, *t3 = st->tw;
		for (j = 0; j < m; j++) {
			cpx s0, s1, s2, s3, s4, s5;
			cmul(&s0, &F[m], t1);
			cmul(&s1, &F[m2], t2);
			cmul(&s2, &F[m3], t3);
			s5.r = F->r - s1.r;
			s5.i = F->i - s1.i;
			F->r += s1.r;
			F->i += s1.i;
			s3.r = s0.r + s2.r;
			s3.i = s0.i + s2.i;
			s4.r = s0.r - s2.r;
			s4.i = s0.i - s2.i;
			F[m2].r = F->r - s3.r;
			F[m2].i = F->i - s3.i;
			t1 += fstride;
			t2 += fstride * 2;
			t3 += fstride * 3;
			F->r += s3.r;
			F->i += s3.i;
			if (st->inverse) {
				F[m].r = s5.r - s4.i;
				F[m].i = s5.i + s4.r;
				F[m3].r = s5.r + s4.i;
				F[m3].i = s5.i - s4.r;
			} else {
				F[m].r = s5.r + s4.i;
				F[m].i = s5.i - s4.r;
				F[m3].r = s5.r - s4.i;
				F[m3].i = s5.i + s4.r;
			}
			++F;
		}
	}
}

/* kf_bfly3 kiss_fft.c:150-190 (float build: HALF_OF(x) = x * .5f, no C_FIXDIV) */
static void bfly3(cpx *F, size_t fstride, const CFft *st, int m) {
	const size_t m2 = 2 * (size_t)m;
	const cpx *t1 = st->tw, *t2 = st->tw;
	const float e = st->tw[fstride * (size_t)m].i;
	int j;
	for (j = 0; j < m; j++) {
		cpx s0, s1, s2, s3;
		cmul(&s1, &F[m], t1);
		cmul(&s2, &F[m2], t2);
		s3.r = s1.r + s2.r;
		s3.i = s1.i + s2.i;
		s0.r = s1.r - s2.r;
		s0.i = s1.i - s2.i;
		t1 += fstride;
		t2 += fstride * 2;
		F[m].r = F->r - s3.r * .5f;
		F[m].i = F->i - s3.i * .5f;
		s0.r *= e;
		s0.i *= e;
		F->r += s3.r;
		F->i += s3.i;
		F[m2].r = F[m].r + s0.i;
		F[m2].i = F[m].i - s0.r;
		F[m].r -= s0.i;
		F[m].i += s0.r;
		++F;
	}
}

/* kf_bfly5 kiss_fft.c:192-256 */
static void bfly5(cpx *F, size_t fstride, const CFft *st, int m) {
	cpx *F0 = F, *F1 = F + m, *F2 = F + 2 * m, *F3 = F + 3 * m, *F4 = F + 4 * m;
	const cpx ya = st->tw[fstride * (size_t)m], yb = st->tw[fstride * 2 * (size_t)m];
	int u;
	for (u = 0; u < m; ++u) {
		cpx s0 = *F0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;
		cmul(&s1, F1, &st->tw[(size_t)u * fstride]);
		cmul(&s2, F2, &st->tw[2 * (size_t)u * fstride]);
		cmul(&s3, F3, &st->tw[3 * (size_t)u * fstride]);
		cmul(&s4, F4, &st->tw[4 * (size_t)u * fstride]);
		s7.r = s1.r + s4.r;
		s7.i = s1.i + s4.i;
		s10.r = s1.r - s4.r;
		s10.i = s1.i - s4.i;
		s8.r = s2.r + s3.r;
		s8.i = s2.i + s3.i;
		s9.r = s2.r - s3.r;
		s9.i = s2.i - s3.i;
		F0->r += s7.r + s8.r;
		F0->i += s7.i + s8.i;
		s5.r = s0.r + s7.r * ya.r + s8.r * yb.r;
		s5.i = s0.i + s7.i * ya.r + s8.i * yb.r;
		s6.r = s10.i * ya.i + s9.i * yb.i;
		s6.i = -(s10.r * ya.i) - s9.r * yb.i;
		F1->r = s5.r - s6.r;
		F1->i = s5.i - s6.i;
		F4->r = s5.r + s6.r;
		F4->i = s5.i + s6.i;
		s11.r = s0.r + s7.r * yb.r + s8.r * ya.r;
		s11.i = s0.i + s7.i * yb.r + s8.i * ya.r;
		s12.r = -(s10.i * yb.i) + s9.i * ya.i;
		s12.i = s10.r * yb.i - s9.r * ya.i;
		F2->r = s11.r + s12.r;
		F2->i = s11.i + s12.i;
		F3->r = s11.r - s12.r;
		F3->i = s11.i - s12.i;
		++F0, ++F1, ++F2, ++F3, ++F4;
	}
}

/* kf_bfly_generic kiss_fft.c:258-290 (radix up to 17) */
static void bfly_generic(cpx *F, size_t fstride, const CFft *st, int m, int p) {
	cpx buf[17];
	int u, k, q1, q;
	if (p > 17) abort();
	for (u = 0; u < m; ++u) {
		k = u;
		for (q1 = 0; q1 < p; ++q1) {
			buf[q1] = F[k];
			k += m;
		}
		k = u;
		for (q1 = 0; q1 < p; ++q1) {
			int twidx = 0;
			F[k] = buf[0];
			for (q = 1; q < p; ++q) {
				cpx t;
				twidx += (int)(fstride * (size_t)k);
				if (twidx >= st->n) twidx -= st->n;
				cmul(&t, &buf[q], &st->tw[twidx]);
				F[k].r += t.r;
				F[k].i += t.i;
			}
			k += m;
		}
	}
}

/* kf_work kiss_fft.c:320-408: deepest stage first */
static void work(cpx *out, size_t fstride, const int *fac, const CFft *st, int N, int m2) {
	const int p = fac[0], m = fac[1];
	if (m != 1) work(out, fstride * p, fac + 2, st, N * p, m);
	if (p == 2) bfly2(out, fstride, st, m, N, m2);
	else if (p == 4) bfly4(out, fstride, st, m, N, m2);
	else { /* :383-405: the odd radices take one sub-transform at a time */
		int i;
		for (i = 0; i < N; i++) {
			if (p == 3) bfly3(out + (size_t)i * m2, fstride, st, m);
			else if (p == 5) bfly5(out + (size_t)i * m2, fstride, st, m);
			else bfly_generic(out + (size_t)i * m2, fstride, st, m, p);
		}
	}
}

static void cfft(const CFft *st, const cpx *in, cpx *out) {
	shuffle(out, in, 1, st->fac);
	work(out, 1, st->fac, st, 1, 1);
}

typedef struct {
	CFft sub;
	cpx *tmp, *super;
} RFft;

static void rfft_init(RFft *st, int nfft, int inverse) { /* kiss_fftr.c:40-81 */
	int i, n = nfft >> 1;
	cfft_init(&st->sub, n, inverse);
	st->tmp = (cpx *)malloc(sizeof(cpx) * (size_t)n);
	st->super = (cpx *)malloc(sizeof(cpx) * (size_t)n);
	for (i = 0; i < n; ++i) {
		const double pi = 3.14159265358979323846264338327;
		double phase = pi * (((double)i) / n + .5);
		if (!inverse) phase = -phase;
		st->super[i].r = (float)cos(phase);
		st->super[i].i = (float)sin(phase);
	}
}

static void rfft_free(RFft *st) {
	free(st->sub.tw);
	free(st->tmp);
	free(st->super);
}

/* kiss_fftr2 kiss_fftr.c:175-259, float branch: packed [DC,r1,i1,...,Nyq] */
static void rfft_forward(RFft *st, const float *t, float *f) {
	int k, n = st->sub.n;
	cfft(&st->sub, (const cpx *)t, st->tmp);
	f[0] = st->tmp[0].r + st->tmp[0].i;
	f[2 * n - 1] = st->tmp[0].r - st->tmp[0].i;
	for (k = 1; k <= n / 2; ++k) {
		float f2r = st->tmp[k].r - st->tmp[n - k].r;
		float f2i = st->tmp[k].i + st->tmp[n - k].i;
		float f1r = st->tmp[k].r + st->tmp[n - k].r;
		float f1i = st->tmp[k].i - st->tmp[n - k].i;
		float twr = f2r * st->super[k].r - f2i * st->super[k].i;
		float twi = f2i * st->super[k].r + f2r * st->super[k].i;
		f[2 * k - 1] = .5f * (f1r + twr);
		f[2 * k] = .5f * (f1i + twi);
		f[2 * (n - k) - 1] = .5f * (f1r - twr);
		f[2 * (n - k)] = .5f * (twi - f1i);
	}
}

/* kiss_fftri2 kiss_fftr.c:261-296 */
static void rfft_inverse(RFft *st, const float *f, float *t) {
	int k, n = st->sub.n;
	st->tmp[0].r = f[0] + f[2 * n - 1];
	st->tmp[0].i = f[0] - f[2 * n - 1];
	for (k = 1; k <= n / 2; ++k) {
		cpx fk, fnkc, fek, fok, d;
		fk.r = f[2 * k - 1];
		fk.i = f[2 * k];
		fnkc.r = f[2 * (n - k) - 1];
		fnkc.i = -f[2 * (n - k)];
		fek.r = fk.r + fnkc.r;
		fek.i = fk.i + fnkc.i;
		d.r = fk.r - fnkc.r;
		d.i = fk.i - fnkc.i;
		cmul(&fok, &d, &st->super[k]);
		st->tmp[k].r = fek.r + fok.r;
		st->tmp[k].i = fek.i + fok.i;
		st->tmp[n - k].r = fek.r - fok.r;
		st->tmp[n - k].i = fek.i - fok.i;
		st->tmp[n - k].i *= -1;
	}
	cfft(&st->sub, st->tmp, (cpx *)t);
}

/* persistent handle = ms_fft_init (dsptools.c:333-341): forward + backward tables */
struct OrcFft {
	RFft fwd, bwd;
	int n;
};

OrcFft *orc_fft_new(int nfft) {
	OrcFft *t = (OrcFft *)malloc(sizeof(*t));
	rfft_init(&t->fwd, nfft, 0);
	rfft_init(&t->bwd, nfft, 1);
	t->n = nfft;
	return t;
}

void orc_fft_free(OrcFft *t) {
	if (!t) return;
	rfft_free(&t->fwd);
	rfft_free(&t->bwd);
	free(t);
}

void orc_fft_forward(OrcFft *t, const float *in, float *out) { /* == ms_fft == speex spx_fft (kiss backend) */
	int i;
	float scale = 1.f / t->n;
	rfft_forward(&t->fwd, in, out);
	for (i = 0; i < t->n; i++) out[i] *= scale;
}

void orc_fft_inverse(OrcFft *t, const float *in, float *out) { rfft_inverse(&t->bwd, in, out); }

void orc_ms_fft(int nfft, const float *in, float *out) { /* dsptools.c:358-369 */
	RFft st;
	int i;
	float scale = 1.f / nfft;
	rfft_init(&st, nfft, 0);
	rfft_forward(&st, in, out);
	for (i = 0; i < nfft; i++) out[i] *= scale;
	rfft_free(&st);
}

void orc_ms_ifft(int nfft, const float *in, float *out) { /* dsptools.c:373-376 */
	RFft st;
	rfft_init(&st, nfft, 1);
	rfft_inverse(&st, in, out);
	rfft_free(&st);
}

/* dsptools.c:253-268 */
void orc_fir_mem16(const float *x, const float *num, float *y, int N, int ord, float *mem) {
	int i, j;
	for (i = 0; i < N; ++i) {
		float acc;
		mem[0] = x[i];
		acc = mem[ord - 1] * num[ord - 1];
		for (j = ord - 2; j >= 0; --j) {
			acc += num[j] * mem[j];
			mem[j + 1] = mem[j];
		}
		y[i] = acc;
	}
}

/* ---- equalizer.c ---- */
#define GAIN_ZERODB 1.0f

static void flatten(OrcEqualizer *s) { /* :49-55 */
	int i;
	float val = (float)(GAIN_ZERODB / s->nfft);
	s->fft_cpx[0] = val;
	for (i = 1; i < s->nfft; i += 2) s->fft_cpx[i] = val;
}

void orc_equalizer_set_rate(OrcEqualizer *s, int rate) { /* :57-79 */
	int n = rate < 16000 ? 128 : (rate < 32000 ? 256 : 512);
	s->rate = rate;
	s->nfft = n;
	free(s->fft_cpx);
	free(s->fir);
	free(s->mem);
	s->fft_cpx = (float *)calloc((size_t)n, sizeof(float));
	flatten(s);
	s->fir_len = n;
	s->fir = (float *)calloc((size_t)n, sizeof(float));
	s->mem = (float *)calloc((size_t)n, sizeof(float));
	s->needs_update = 1;
}

OrcEqualizer *orc_equalizer_new(int rate) {
	OrcEqualizer *s = (OrcEqualizer *)calloc(1, sizeof(*s));
	orc_equalizer_set_rate(s, rate);
	s->active = 1;
	return s;
}

void orc_equalizer_free(OrcEqualizer *s) {
	if (!s) return;
	free(s->fft_cpx);
	free(s->fir);
	free(s->mem);
	free(s);
}

static int hz_to_index(const OrcEqualizer *s, int hz) { /* :95-108 */
	int ret;
	if (hz < 0) return -1;
	if (hz > (s->rate / 2)) hz = (s->rate / 2);
	ret = ((hz * s->nfft) + (s->rate / 2)) / s->rate;
	if (ret == s->nfft / 2) ret = (s->nfft / 2) - 1;
	return ret;
}

static int index2hz(const OrcEqualizer *s, int index) { return (index * s->rate + s->nfft / 2) / s->nfft; }

static float gainpoint(int f, int freq_0, float sqrt_gain, int freq_bw) { /* :128-135 */
	float k1, k2;
	k1 = ((float)(f * f) - (float)(freq_0 * freq_0));
	k1 *= k1;
	k2 = (float)(f * freq_bw);
	k2 *= k2;
	return (k1 + k2 * sqrt_gain) / (k1 + k2 / sqrt_gain);
}

static void point_set(OrcEqualizer *s, int i, float gain) { /* :137-145 */
	int index = 1 + ((i - 1) * 2);
	if (index >= 0 && index < s->nfft) s->fft_cpx[index] = (s->fft_cpx[index] * (int)(gain * 32768)) / 32768;
}

void orc_equalizer_set_gain(OrcEqualizer *s, int freq_0, float gain, int freq_bw) { /* :147-172 */
	int i, f;
	int delta_f = index2hz(s, 1);
	float sqrt_gain = (float)sqrt(gain);
	int mid = hz_to_index(s, freq_0);
	freq_bw -= delta_f / 2;
	if (freq_bw < delta_f / 2) freq_bw = delta_f / 2;
	i = mid;
	point_set(s, i, gain);
	do {
		i++;
		f = index2hz(s, i);
		gain = gainpoint(f - delta_f, freq_0, sqrt_gain, freq_bw);
		point_set(s, i, gain);
	} while (i < s->nfft / 2 && (gain > 1.1 || gain < 0.9));
	i = mid;
	do {
		i--;
		f = index2hz(s, i);
		gain = gainpoint(f + delta_f, freq_0, sqrt_gain, freq_bw);
		point_set(s, i, gain);
	} while (i >= 0 && (gain > 1.1 || gain < 0.9));
	s->needs_update = 1;
}

void orc_equalizer_design(OrcEqualizer *s) { /* :215-237 with :184-213 */
	int i, half = s->fir_len / 2;
	orc_ms_ifft(s->nfft, s->fft_cpx, s->fir);
	for (i = 0; i < half; ++i) { /* time_shift */
		float tmp = s->fir[i];
		s->fir[i] = s->fir[i + half];
		s->fir[i + half] = tmp;
	}
	for (i = 0; i < s->fir_len; ++i) { /* norm_and_apodize: Hamming */
		float x = (float)((float)i * 2 * M_PI / (float)s->fir_len);
		float w = (float)(0.54 - (0.46 * cos(x)));
		s->fir[i] = w * (float)s->fir[i];
	}
	s->needs_update = 0;
}

void orc_equalizer_run(OrcEqualizer *s, int16_t *samples, int nsamples) { /* :263-269 */
	float *w = (float *)malloc(sizeof(float) * (size_t)nsamples);
	int i;
	if (s->needs_update) orc_equalizer_design(s);
	for (i = 0; i < nsamples; ++i) w[i] = (float)samples[i];
	orc_fir_mem16(w, s->fir, w, nsamples, s->fir_len, s->mem);
	for (i = 0; i < nsamples; ++i) {
		float v = w[i];
		/* reference: (int16_t)v, UB out of range (equalizer.c:251-255); we saturate */
		samples[i] = (int16_t)(v >= 32767.f ? 32767 : (v <= -32768.f ? -32768 : (int)v));
	}
	free(w);
}
