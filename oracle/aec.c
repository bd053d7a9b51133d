/* oracle/aec.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 *
 * CPU restatement of what /root/reference/src/audiofilters/speexec.c:297-298
 * calls per frame: speex_echo_cancellation (libspeexdsp mdf.c, the MDF /
 * AUMDF two-path echo canceller of J.-M. Valin, float build, mono) followed by
 * speex_preprocess_run (libspeexdsp preprocess.c with an echo state attached,
 * denoise on, AGC/VAD/dereverb off -- the only configuration speexec.c:200-203
 * sets up).  libspeexdsp is third-party, un-vendored and unpinned
 * (CMakeLists.txt:207-209); this follows the published algorithm of speexdsp
 * 1.2.x with the library's float-build macro semantics (shifts are no-ops,
 * Q15 multiplies are plain products), including two quirks of that build:
 * the far-end energy Sxx is accumulated twice per frame, and the "N*10000>>6"
 * style thresholds are not shifted.  FFT backend: the kiss_fft real transform
 * (the library's USE_KISS_FFT backend == src/utils/kiss_fftr.c here); the
 * library's default float backend (smallft) differs by rounding only.
 *
 * Compile with -ffp-contract=off.  Expressions are written with explicit
 * float/double types so the HIP kernel can mirror them operation by operation.
 */
#include "ms2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int adjust_framesize_8000(int framesize, int samplerate) { /* speexec.c:171-180 */
	int newsize = (framesize * samplerate) / 8000;
	int n = 1, next;
	while ((next = n << 1) <= newsize) n = next;
	return n;
}

struct OrcEcho {
	int frame_size, window_size, M;
	int cancel_count, adapted, saturated, screwed_up;
	int sampling_rate;
	float spec_average, beta0, beta_max, sum_adapt, leak_estimate;
	float *e, *x, *X, *input, *y, *last_y, *Y, *E, *PHI, *W, *foreground;
	float Davg1, Davg2, Dvar1, Dvar2;
	float *power, *power_1, *wtmp, *Rf, *Yf, *Xf, *Eh, *Yh;
	float Pey, Pyy;
	float *window, *prop;
	OrcFft *fft;
	float memX, memD, memE, preemph, notch_radius;
	float notch_mem[2];
};

static float *zalloc(int n) { return (float *)calloc((size_t)n, sizeof(float)); }

static void set_rate(OrcEcho *st, int rate) { /* SPEEX_ECHO_SET_SAMPLING_RATE */
	st->sampling_rate = rate;
	st->spec_average = (float)st->frame_size / (float)rate;
	st->beta0 = (2.0f * st->frame_size) / rate;
	st->beta_max = (.5f * st->frame_size) / rate;
	if (rate < 12000) st->notch_radius = .9f;
	else if (rate < 24000) st->notch_radius = .982f;
	else st->notch_radius = .992f;
}

OrcEcho *orc_echo_new(int frame_size, int filter_length, int sample_rate) {
	OrcEcho *st = (OrcEcho *)calloc(1, sizeof(*st));
	int i, N, M;
	st->frame_size = frame_size;
	st->window_size = N = 2 * frame_size;
	st->M = M = (filter_length + frame_size - 1) / frame_size;
	st->fft = orc_fft_new(N);
	st->e = zalloc(N);
	st->x = zalloc(N);
	st->input = zalloc(frame_size);
	st->y = zalloc(N);
	st->last_y = zalloc(N);
	st->Yf = zalloc(frame_size + 1);
	st->Rf = zalloc(frame_size + 1);
	st->Xf = zalloc(frame_size + 1);
	st->Yh = zalloc(frame_size + 1);
	st->Eh = zalloc(frame_size + 1);
	st->X = zalloc((M + 1) * N);
	st->Y = zalloc(N);
	st->E = zalloc(N);
	st->W = zalloc(M * N);
	st->foreground = zalloc(M * N);
	st->PHI = zalloc(N);
	st->power = zalloc(frame_size + 1);
	st->power_1 = zalloc(frame_size + 1);
	st->window = zalloc(N);
	st->prop = zalloc(M);
	st->wtmp = zalloc(N);
	for (i = 0; i < N; i++) st->window[i] = (float)(.5 - .5 * cos(2 * M_PI * i / N));
	for (i = 0; i <= frame_size; i++) st->power_1[i] = 1.0f;
	{
		float sum, decay = (float)exp(-(2.4f / M));
		st->prop[0] = .7f;
		sum = st->prop[0];
		for (i = 1; i < M; i++) {
			st->prop[i] = st->prop[i - 1] * decay;
			sum = sum + st->prop[i];
		}
		for (i = M - 1; i >= 0; i--) st->prop[i] = (.8f * st->prop[i]) / sum;
	}
	st->preemph = .9f;
	st->Pey = st->Pyy = 1.0f;
	set_rate(st, 8000); /* init default, then speexec.c:202 sets the real rate */
	set_rate(st, sample_rate);
	return st;
}

void orc_echo_free(OrcEcho *st) {
	if (!st) return;
	free(st->e); free(st->x); free(st->input); free(st->y); free(st->last_y);
	free(st->Yf); free(st->Rf); free(st->Xf); free(st->Yh); free(st->Eh);
	free(st->X); free(st->Y); free(st->E); free(st->W); free(st->foreground);
	free(st->PHI); free(st->power); free(st->power_1); free(st->window);
	free(st->prop); free(st->wtmp);
	orc_fft_free(st->fft);
	free(st);
}

static void echo_reset(OrcEcho *st) { /* speex_echo_state_reset */
	int i, N = st->window_size, M = st->M;
	st->cancel_count = 0;
	st->screwed_up = 0;
	for (i = 0; i < N * M; i++) st->W[i] = 0;
	for (i = 0; i < N * M; i++) st->foreground[i] = 0;
	for (i = 0; i < N * (M + 1); i++) st->X[i] = 0;
	for (i = 0; i <= st->frame_size; i++) {
		st->power[i] = 0;
		st->power_1[i] = 1.0f;
		st->Eh[i] = 0;
		st->Yh[i] = 0;
	}
	for (i = 0; i < N; i++) st->last_y[i] = 0;
	for (i = 0; i < N; i++) st->E[i] = 0;
	for (i = 0; i < N; i++) st->x[i] = 0;
	st->notch_mem[0] = st->notch_mem[1] = 0;
	st->memD = st->memE = st->memX = 0;
	st->saturated = 0;
	st->adapted = 0;
	st->sum_adapt = 0;
	st->Pey = st->Pyy = 1.0f;
	st->Davg1 = st->Davg2 = 0;
	st->Dvar1 = st->Dvar2 = 0;
}

/* mdf_inner_prod, float build: pairs, then a running sum */
static float inner_prod(const float *x, const float *y, int len) {
	float sum = 0;
	len >>= 1;
	while (len--) {
		float part = 0;
		part = part + (*x) * (*y);
		x++, y++;
		part = part + (*x) * (*y);
		x++, y++;
		sum = sum + part;
	}
	return sum;
}

static void power_spectrum_accum(const float *X, float *ps, int N) {
	int i, j;
	ps[0] += X[0] * X[0];
	for (i = 1, j = 1; i < N - 1; i += 2, j++) ps[j] += X[i] * X[i] + X[i + 1] * X[i + 1];
	ps[j] += X[i] * X[i];
}

static void power_spectrum(const float *X, float *ps, int N) {
	int i, j;
	ps[0] = X[0] * X[0];
	for (i = 1, j = 1; i < N - 1; i += 2, j++) ps[j] = X[i] * X[i] + X[i + 1] * X[i + 1];
	ps[j] = X[i] * X[i];
}

static void spectral_mul_accum(const float *X, const float *Y, float *acc, int N, int M) {
	int i, j;
	for (i = 0; i < N; i++) acc[i] = 0;
	for (j = 0; j < M; j++) {
		acc[0] += X[0] * Y[0];
		for (i = 1; i < N - 1; i += 2) {
			acc[i] += (X[i] * Y[i] - X[i + 1] * Y[i + 1]);
			acc[i + 1] += (X[i + 1] * Y[i] + X[i] * Y[i + 1]);
		}
		acc[i] += X[i] * Y[i];
		X += N;
		Y += N;
	}
}

static void weighted_spectral_mul_conj(const float *w, const float p, const float *X, const float *Y, float *prod,
                                       int N) {
	int i, j;
	float W;
	W = p * w[0];
	prod[0] = W * (X[0] * Y[0]);
	for (i = 1, j = 1; i < N - 1; i += 2, j++) {
		W = p * w[j];
		prod[i] = W * ((X[i] * Y[i]) + X[i + 1] * Y[i + 1]);
		prod[i + 1] = W * (((-X[i + 1]) * Y[i]) + X[i] * Y[i + 1]);
	}
	W = p * w[j];
	prod[i] = W * (X[i] * Y[i]);
}

static void adjust_prop(const float *W, int N, int M, float *prop) {
	int i, j;
	float max_sum = 1, prop_sum = 1;
	for (i = 0; i < M; i++) {
		float tmp = 1;
		for (j = 0; j < N; j++) tmp += W[i * N + j] * W[i * N + j];
		prop[i] = (float)sqrt(tmp);
		if (prop[i] > max_sum) max_sum = prop[i];
	}
	for (i = 0; i < M; i++) {
		prop[i] += .1f * max_sum;
		prop_sum += prop[i];
	}
	for (i = 0; i < M; i++) prop[i] = (.99f * prop[i]) / prop_sum;
}

static int16_t word2int(float x) {
	return (int16_t)(x < -32767.5f ? -32768 : (x > 32766.5f ? 32767 : floor(.5 + x)));
}

void orc_echo_cancel(OrcEcho *st, const int16_t *in, const int16_t *far_end, int16_t *out) {
	int i, j;
	const int N = st->window_size, M = st->M, F = st->frame_size;
	float Syy, See, Sxx, Sdd, Sff, Dbf, Sey;
	int update_foreground;
	float ss, ss_1;
	float Pey = 1.0f, Pyy = 1.0f;
	float alpha, alpha_1, RER, tmp32;

	st->cancel_count++;
	ss = (float)(.35 / M);
	ss_1 = 1 - ss;

	/* DC notch (filter_dc_notch16) then pre-emphasis on the microphone signal */
	{
		const float radius = st->notch_radius;
		const float den2 = (float)(radius * radius + .7 * (1 - radius) * (1 - radius));
		for (i = 0; i < F; i++) {
			float vin = in[i];
			float vout = st->notch_mem[0] + vin;
			st->notch_mem[0] = st->notch_mem[1] + 2 * (-vin + radius * vout);
			st->notch_mem[1] = vin - den2 * vout;
			st->input[i] = radius * vout;
		}
		for (i = 0; i < F; i++) {
			float t = st->input[i] - st->preemph * st->memD;
			st->memD = st->input[i];
			st->input[i] = t;
		}
	}
	/* far end: shift, pre-emphasis */
	for (i = 0; i < F; i++) {
		float t;
		st->x[i] = st->x[i + F];
		t = (float)far_end[i] - st->preemph * st->memX;
		st->x[i + F] = t;
		st->memX = far_end[i];
	}
	/* shift the spectral history, newest block first */
	for (j = M - 1; j >= 0; j--)
		for (i = 0; i < N; i++) st->X[(j + 1) * N + i] = st->X[j * N + i];
	orc_fft_forward(st->fft, st->x, st->X);

	Sxx = 0;
	Sxx += inner_prod(st->x + F, st->x + F, F);
	power_spectrum_accum(st->X, st->Xf, N);

	/* foreground filter output and its error */
	Sff = 0;
	spectral_mul_accum(st->X, st->foreground, st->Y, N, M);
	orc_fft_inverse(st->fft, st->Y, st->e);
	for (i = 0; i < F; i++) st->e[i] = st->input[i] - st->e[i + F];
	Sff += inner_prod(st->e, st->e, F);

	if (st->adapted) adjust_prop(st->W, N, M, st->prop);
	/* background weight gradient from the PREVIOUS frame's error spectrum */
	if (st->saturated == 0) {
		for (j = M - 1; j >= 0; j--) {
			weighted_spectral_mul_conj(st->power_1, st->prop[j], &st->X[(j + 1) * N], st->E, st->PHI, N);
			for (i = 0; i < N; i++) st->W[j * N + i] += st->PHI[i];
		}
	} else {
		st->saturated--;
	}
	/* AUMDF constraint: block 0 every frame, one other block round-robin */
	for (j = 0; j < M; j++) {
		if (j == 0 || st->cancel_count % (M - 1) == j - 1) {
			orc_fft_inverse(st->fft, &st->W[j * N], st->wtmp);
			for (i = F; i < N; i++) st->wtmp[i] = 0;
			orc_fft_forward(st->fft, st->wtmp, &st->W[j * N]);
		}
	}

	for (i = 0; i <= F; i++) st->Rf[i] = st->Yf[i] = st->Xf[i] = 0;

	Dbf = 0;
	See = 0;
	/* background filter output; difference to the foreground response */
	spectral_mul_accum(st->X, st->W, st->Y, N, M);
	orc_fft_inverse(st->fft, st->Y, st->y);
	for (i = 0; i < F; i++) st->e[i] = st->e[i + F] - st->y[i + F];
	Dbf += 10 + inner_prod(st->e, st->e, F);
	for (i = 0; i < F; i++) st->e[i] = st->input[i] - st->y[i + F];
	See += inner_prod(st->e, st->e, F);

	/* two-path control */
	st->Davg1 = .6f * st->Davg1 + .4f * (Sff - See);
	st->Davg2 = .85f * st->Davg2 + .15f * (Sff - See);
	st->Dvar1 = .36f * st->Dvar1 + (.4f * Sff) * (.4f * Dbf);
	st->Dvar2 = .7225f * st->Dvar2 + (.15f * Sff) * (.15f * Dbf);

	update_foreground = 0;
	if ((Sff - See) * fabsf(Sff - See) > Sff * Dbf) update_foreground = 1;
	else if (st->Davg1 * fabsf(st->Davg1) > .5f * st->Dvar1) update_foreground = 1;
	else if (st->Davg2 * fabsf(st->Davg2) > .25f * st->Dvar2) update_foreground = 1;

	if (update_foreground) {
		st->Davg1 = st->Davg2 = 0;
		st->Dvar1 = st->Dvar2 = 0;
		for (i = 0; i < N * M; i++) st->foreground[i] = st->W[i];
		for (i = 0; i < F; i++)
			st->e[i + F] = st->window[i + F] * st->e[i + F] + st->window[i] * st->y[i + F];
	} else {
		int reset_background = 0;
		if ((-(Sff - See)) * fabsf(Sff - See) > 4.f * (Sff * Dbf)) reset_background = 1;
		if ((-st->Davg1) * fabsf(st->Davg1) > 4.f * st->Dvar1) reset_background = 1;
		if ((-st->Davg2) * fabsf(st->Davg2) > 4.f * st->Dvar2) reset_background = 1;
		if (reset_background) {
			for (i = 0; i < N * M; i++) st->W[i] = st->foreground[i];
			for (i = 0; i < F; i++) st->y[i + F] = st->e[i + F];
			for (i = 0; i < F; i++) st->e[i] = st->input[i] - st->y[i + F];
			See = Sff;
			st->Davg1 = st->Davg2 = 0;
			st->Dvar1 = st->Dvar2 = 0;
		}
	}

	Sey = Syy = Sdd = 0;
	/* output: error of the foreground path, de-emphasised */
	for (i = 0; i < F; i++) {
		float tmp_out = st->input[i] - st->e[i + F];
		tmp_out = tmp_out + st->preemph * st->memE;
		if (in[i] <= -32000 || in[i] >= 32000) {
			if (st->saturated == 0) st->saturated = 1;
		}
		out[i] = word2int(tmp_out);
		st->memE = tmp_out;
	}
	/* error signal for the filter update (background path), zero-padded in front */
	for (i = 0; i < F; i++) {
		st->e[i + F] = st->e[i];
		st->e[i] = 0;
	}
	Sey += inner_prod(st->e + F, st->y + F, F);
	Syy += inner_prod(st->y + F, st->y + F, F);
	Sdd += inner_prod(st->input, st->input, F);

	orc_fft_forward(st->fft, st->e, st->E);
	for (i = 0; i < F; i++) st->y[i] = 0;
	orc_fft_forward(st->fft, st->y, st->Y);
	power_spectrum_accum(st->E, st->Rf, N);
	power_spectrum_accum(st->Y, st->Yf, N);

	/* sanity checks */
	if (!(Syy >= 0 && Sxx >= 0 && See >= 0) || !(Sff < N * 1e9 && Syy < N * 1e9 && Sxx < N * 1e9)) {
		st->screwed_up += 50;
		for (i = 0; i < F; i++) out[i] = 0;
	} else if (Sff > Sdd + (float)(N * 10000)) {
		st->screwed_up++;
	} else {
		st->screwed_up = 0;
	}
	if (st->screwed_up >= 50) {
		echo_reset(st);
		return;
	}

	if (See < (float)(N * 100)) See = (float)(N * 100);

	Sxx += inner_prod(st->x + F, st->x + F, F); /* sic: accumulated a second time */
	power_spectrum_accum(st->X, st->Xf, N);

	for (j = 0; j <= F; j++) st->power[j] = ss_1 * st->power[j] + 1 + ss * st->Xf[j];

	for (j = F; j >= 0; j--) {
		float Eh, Yh;
		Eh = st->Rf[j] - st->Eh[j];
		Yh = st->Yf[j] - st->Yh[j];
		Pey = Pey + Eh * Yh;
		Pyy = Pyy + Yh * Yh;
		st->Eh[j] = (1 - st->spec_average) * st->Eh[j] + st->spec_average * st->Rf[j];
		st->Yh[j] = (1 - st->spec_average) * st->Yh[j] + st->spec_average * st->Yf[j];
	}
	Pyy = (float)sqrt(Pyy);
	Pey = Pey / Pyy;

	tmp32 = st->beta0 * Syy;
	if (tmp32 > st->beta_max * See) tmp32 = st->beta_max * See;
	alpha = tmp32 / See;
	alpha_1 = 1.0f - alpha;
	st->Pey = alpha_1 * st->Pey + alpha * Pey;
	st->Pyy = alpha_1 * st->Pyy + alpha * Pyy;
	if (st->Pyy < 1.0f) st->Pyy = 1.0f;
	if (st->Pey < .005f * st->Pyy) st->Pey = .005f * st->Pyy;
	if (st->Pey > st->Pyy) st->Pey = st->Pyy;
	st->leak_estimate = st->Pey / st->Pyy;

	RER = (float)((.0001 * Sxx + 3. * (st->leak_estimate * Syy)) / See);
	if (RER < Sey * Sey / (1 + See * Syy)) RER = Sey * Sey / (1 + See * Syy);
	if (RER > .5) RER = .5;

	if (!st->adapted && st->sum_adapt > (float)M && st->leak_estimate * Syy > .03f * Syy) st->adapted = 1;

	if (st->adapted) {
		for (i = 0; i <= F; i++) {
			float r, e;
			r = st->leak_estimate * st->Yf[i];
			e = st->Rf[i] + 1;
			if (r > .5 * e) r = (float)(.5 * e);
			r = .7f * r + .3f * (float)(RER * e);
			st->power_1[i] = r / (e * (st->power[i] + 10));
		}
	} else {
		float adapt_rate = 0;
		if (Sxx > (float)(N * 1000)) {
			tmp32 = .25f * Sxx;
			if (tmp32 > .25 * See) tmp32 = (float)(.25 * See);
			adapt_rate = tmp32 / See;
		}
		for (i = 0; i <= F; i++) st->power_1[i] = adapt_rate / (st->power[i] + 10);
		st->sum_adapt = st->sum_adapt + adapt_rate;
	}

	for (i = 0; i < F; i++) st->last_y[i] = st->last_y[F + i];
	if (st->adapted) {
		for (i = 0; i < F; i++) st->last_y[F + i] = (float)(in[i] - out[i]);
	}
}

/* speex_echo_get_residual */
static void echo_get_residual(OrcEcho *st, float *residual_echo) {
	int i, N = st->window_size;
	float leak2;
	for (i = 0; i < N; i++) st->y[i] = st->window[i] * st->last_y[i];
	orc_fft_forward(st->fft, st->y, st->Y);
	power_spectrum(st->Y, residual_echo, N);
	if (st->leak_estimate > .5) leak2 = 1;
	else leak2 = 2 * st->leak_estimate;
	for (i = 0; i <= st->frame_size; i++) residual_echo[i] = (float)(int32_t)(leak2 * residual_echo[i]);
}

int orc_echo_get(const OrcEcho *st, const char *what, float *dst, int cap) {
	const float *src = NULL;
	float scal[16];
	int n = 0, N = st->window_size, M = st->M, F = st->frame_size;
	if (!strcmp(what, "W")) src = st->W, n = M * N;
	else if (!strcmp(what, "foreground")) src = st->foreground, n = M * N;
	else if (!strcmp(what, "X")) src = st->X, n = (M + 1) * N;
	else if (!strcmp(what, "E")) src = st->E, n = N;
	else if (!strcmp(what, "power")) src = st->power, n = F + 1;
	else if (!strcmp(what, "power_1")) src = st->power_1, n = F + 1;
	else if (!strcmp(what, "Eh")) src = st->Eh, n = F + 1;
	else if (!strcmp(what, "Yh")) src = st->Yh, n = F + 1;
	else if (!strcmp(what, "prop")) src = st->prop, n = M;
	else if (!strcmp(what, "last_y")) src = st->last_y, n = N;
	else if (!strcmp(what, "window")) src = st->window, n = N;
	else if (!strcmp(what, "scalars")) {
		scal[0] = st->Davg1, scal[1] = st->Davg2, scal[2] = st->Dvar1, scal[3] = st->Dvar2;
		scal[4] = st->Pey, scal[5] = st->Pyy, scal[6] = st->sum_adapt, scal[7] = st->leak_estimate;
		scal[8] = (float)st->adapted, scal[9] = (float)st->saturated, scal[10] = (float)st->screwed_up;
		scal[11] = (float)st->cancel_count, scal[12] = st->memX, scal[13] = st->memD, scal[14] = st->memE;
		scal[15] = st->notch_mem[0];
		src = scal, n = 16;
	} else return -1;
	if (n > cap) n = cap;
	memcpy(dst, src, sizeof(float) * (size_t)n);
	return n;
}

/* ===================================================================== */
/* preprocessor (preprocess.c + filterbank.c), float build               */
#define NB_BANDS 24

typedef struct {
	int *bank_left, *bank_right;
	float *filter_left, *filter_right;
	int nb_banks, len;
} Bank;

static float to_bark(float n) {
	return (float)(13.1f * atan(.00074f * n) + 2.24f * atan(n * n * 1.85e-8f) + 1e-4f * n);
}

static Bank *bank_new(int banks, float sampling, int len) {
	Bank *b = (Bank *)calloc(1, sizeof(*b));
	float df, max_mel, mel_interval;
	int i, id1, id2;
	df = sampling / (float)(2 * len);
	max_mel = to_bark(sampling / 2);
	mel_interval = max_mel / (float)(banks - 1);
	b->nb_banks = banks;
	b->len = len;
	b->bank_left = (int *)calloc((size_t)len, sizeof(int));
	b->bank_right = (int *)calloc((size_t)len, sizeof(int));
	b->filter_left = zalloc(len);
	b->filter_right = zalloc(len);
	for (i = 0; i < len; i++) {
		float curr_freq, mel, val;
		curr_freq = (float)i * df;
		mel = to_bark(curr_freq);
		if (mel > max_mel) break;
		id1 = (int)(floor(mel / mel_interval));
		if (id1 > banks - 2) {
			id1 = banks - 2;
			val = 1.0f;
		} else {
			val = (mel - id1 * mel_interval) / mel_interval;
		}
		id2 = id1 + 1;
		b->bank_left[i] = id1;
		b->filter_left[i] = 1.0f - val;
		b->bank_right[i] = id2;
		b->filter_right[i] = val;
	}
	return b;
}

static void bank_free(Bank *b) {
	free(b->bank_left); free(b->bank_right); free(b->filter_left); free(b->filter_right);
	free(b);
}

static void bank_compute_bank32(const Bank *b, const float *ps, float *mel) {
	int i;
	for (i = 0; i < b->nb_banks; i++) mel[i] = 0;
	for (i = 0; i < b->len; i++) {
		mel[b->bank_left[i]] += b->filter_left[i] * ps[i];
		mel[b->bank_right[i]] += b->filter_right[i] * ps[i];
	}
}

static void bank_compute_psd16(const Bank *b, const float *mel, float *ps) {
	int i;
	for (i = 0; i < b->len; i++) {
		float tmp = mel[b->bank_left[i]] * b->filter_left[i];
		tmp += mel[b->bank_right[i]] * b->filter_right[i];
		ps[i] = tmp;
	}
}

struct OrcPreproc {
	int frame_size, ps_size, sampling_rate, nbands;
	Bank *bank;
	int noise_suppress, echo_suppress, echo_suppress_active;
	OrcEcho *echo_state;
	float *frame, *ft, *ps, *gain2, *gain_floor, *window, *noise, *reverb_estimate, *old_ps, *gain, *prior, *post;
	float *S, *Smin, *Stmp;
	int *update_prob;
	float *zeta, *echo_noise, *residual_echo;
	float *inbuf, *outbuf;
	int nb_adapt, min_count;
	OrcFft *fft;
};

static void conj_window(float *w, int len) {
	int i;
	for (i = 0; i < len; i++) {
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

OrcPreproc *orc_preproc_new(int frame_size, int sample_rate, OrcEcho *echo) {
	OrcPreproc *st = (OrcPreproc *)calloc(1, sizeof(*st));
	int i, N, M;
	st->frame_size = frame_size;
	st->ps_size = N = frame_size;
	st->sampling_rate = sample_rate;
	st->noise_suppress = -15;
	st->echo_suppress = -40;
	st->echo_suppress_active = -15;
	st->echo_state = echo;
	st->nbands = M = NB_BANDS;
	st->bank = bank_new(M, (float)sample_rate, N);
	st->frame = zalloc(2 * N);
	st->window = zalloc(2 * N);
	st->ft = zalloc(2 * N);
	st->ps = zalloc(N + M);
	st->noise = zalloc(N + M);
	st->echo_noise = zalloc(N + M);
	st->residual_echo = zalloc(N + M);
	st->reverb_estimate = zalloc(N + M);
	st->old_ps = zalloc(N + M);
	st->prior = zalloc(N + M);
	st->post = zalloc(N + M);
	st->gain = zalloc(N + M);
	st->gain2 = zalloc(N + M);
	st->gain_floor = zalloc(N + M);
	st->zeta = zalloc(N + M);
	st->S = zalloc(N);
	st->Smin = zalloc(N);
	st->Stmp = zalloc(N);
	st->update_prob = (int *)calloc((size_t)N, sizeof(int));
	st->inbuf = zalloc(N);
	st->outbuf = zalloc(N);
	conj_window(st->window, 2 * N);
	for (i = 0; i < N + M; i++) {
		st->noise[i] = 1.f;
		st->reverb_estimate[i] = 0;
		st->old_ps[i] = 1;
		st->gain[i] = 1.0f;
		st->post[i] = 1;
		st->prior[i] = 1;
	}
	for (i = 0; i < N; i++) st->update_prob[i] = 1;
	st->fft = orc_fft_new(2 * N);
	return st;
}

void orc_preproc_free(OrcPreproc *st) {
	if (!st) return;
	bank_free(st->bank);
	free(st->frame); free(st->window); free(st->ft); free(st->ps); free(st->noise); free(st->echo_noise);
	free(st->residual_echo); free(st->reverb_estimate); free(st->old_ps); free(st->prior); free(st->post);
	free(st->gain); free(st->gain2); free(st->gain_floor); free(st->zeta); free(st->S); free(st->Smin);
	free(st->Stmp); free(st->update_prob); free(st->inbuf); free(st->outbuf);
	orc_fft_free(st->fft);
	free(st);
}

static float qcurve(float x) { return 1.f / (1.f + .15f / x); }

static float hypergeom_gain(float xx) {
	int ind;
	float integer, frac, x;
	static const float table[21] = {0.82157f, 1.02017f, 1.20461f, 1.37534f, 1.53363f, 1.68092f, 1.81865f,
	                                1.94811f, 2.07038f, 2.18638f, 2.29688f, 2.40255f, 2.50391f, 2.60144f,
	                                2.69551f, 2.78647f, 2.87458f, 2.96015f, 3.04333f, 3.12431f, 3.20326f};
	x = xx;
	integer = (float)floor(2 * x);
	ind = (int)integer;
	if (ind < 0) return 1.f;
	if (ind > 19) return (float)(1.f * (1 + .1296 / x));
	frac = 2 * x - integer;
	return (float)(((1 - frac) * table[ind] + frac * table[ind + 1]) / sqrt(x + .0001f));
}

static void preprocess_analysis(OrcPreproc *st, const int16_t *x) {
	int i, N = st->ps_size;
	float *ps = st->ps;
	for (i = 0; i < N; i++) st->frame[i] = st->inbuf[i];
	for (i = 0; i < N; i++) st->frame[N + i] = x[i];
	for (i = 0; i < N; i++) st->inbuf[i] = x[i];
	for (i = 0; i < 2 * N; i++) st->frame[i] = st->frame[i] * st->window[i];
	orc_fft_forward(st->fft, st->frame, st->ft);
	ps[0] = st->ft[0] * st->ft[0];
	for (i = 1; i < N; i++) ps[i] = st->ft[2 * i - 1] * st->ft[2 * i - 1] + st->ft[2 * i] * st->ft[2 * i];
	bank_compute_bank32(st->bank, ps, ps + N);
}

static void update_noise_prob(OrcPreproc *st) {
	int i, min_range, N = st->ps_size;
	for (i = 1; i < N - 1; i++)
		st->S[i] = .8f * st->S[i] + .05f * st->ps[i - 1] + .1f * st->ps[i] + .05f * st->ps[i + 1];
	st->S[0] = .8f * st->S[0] + .2f * st->ps[0];
	st->S[N - 1] = .8f * st->S[N - 1] + .2f * st->ps[N - 1];
	if (st->nb_adapt == 1)
		for (i = 0; i < N; i++) st->Smin[i] = st->Stmp[i] = 0;
	if (st->nb_adapt < 100) min_range = 15;
	else if (st->nb_adapt < 1000) min_range = 50;
	else if (st->nb_adapt < 10000) min_range = 150;
	else min_range = 300;
	if (st->min_count > min_range) {
		st->min_count = 0;
		for (i = 0; i < N; i++) {
			st->Smin[i] = st->Stmp[i] < st->S[i] ? st->Stmp[i] : st->S[i];
			st->Stmp[i] = st->S[i];
		}
	} else {
		for (i = 0; i < N; i++) {
			st->Smin[i] = st->Smin[i] < st->S[i] ? st->Smin[i] : st->S[i];
			st->Stmp[i] = st->Stmp[i] < st->S[i] ? st->Stmp[i] : st->S[i];
		}
	}
	for (i = 0; i < N; i++) st->update_prob[i] = (.4f * st->S[i] > st->Smin[i]) ? 1 : 0;
}

void orc_preproc_run(OrcPreproc *st, int16_t *x) {
	int i, N = st->ps_size, M = st->nbands;
	float *ps = st->ps;
	float Zframe, Pframe, beta, beta_1;
	int effective_echo_suppress;

	st->nb_adapt++;
	if (st->nb_adapt > 20000) st->nb_adapt = 20000;
	st->min_count++;
	beta = 1.0f / st->nb_adapt;
	if (beta < .03f) beta = .03f;
	beta_1 = 1.0f - beta;

	if (st->echo_state) {
		echo_get_residual(st->echo_state, st->residual_echo);
		if (!(st->residual_echo[0] >= 0 && st->residual_echo[0] < N * 1e9f))
			for (i = 0; i < N; i++) st->residual_echo[i] = 0;
		for (i = 0; i < N; i++) {
			float a = .6f * st->echo_noise[i];
			st->echo_noise[i] = a > st->residual_echo[i] ? a : st->residual_echo[i];
		}
		bank_compute_bank32(st->bank, st->echo_noise, st->echo_noise + N);
	} else {
		for (i = 0; i < N + M; i++) st->echo_noise[i] = 0;
	}
	preprocess_analysis(st, x);
	update_noise_prob(st);

	for (i = 0; i < N; i++) {
		if (!st->update_prob[i] || st->ps[i] < st->noise[i]) {
			float v = beta_1 * st->noise[i] + beta * st->ps[i];
			st->noise[i] = v > 0 ? v : 0;
		}
	}
	bank_compute_bank32(st->bank, st->noise, st->noise + N);

	if (st->nb_adapt == 1)
		for (i = 0; i < N + M; i++) st->old_ps[i] = ps[i];

	for (i = 0; i < N + M; i++) {
		float gamma, t;
		float tot_noise = 1.f + st->noise[i] + st->echo_noise[i] + st->reverb_estimate[i];
		st->post[i] = ps[i] / tot_noise - 1.f;
		if (st->post[i] > 100.f) st->post[i] = 100.f;
		t = st->old_ps[i] / (st->old_ps[i] + tot_noise);
		gamma = .1f + .89f * (t * t);
		st->prior[i] = gamma * (st->post[i] > 0 ? st->post[i] : 0) + (1.0f - gamma) * (st->old_ps[i] / tot_noise);
		if (st->prior[i] > 100.f) st->prior[i] = 100.f;
	}

	st->zeta[0] = .7f * st->zeta[0] + .3f * st->prior[0];
	for (i = 1; i < N - 1; i++)
		st->zeta[i] = .7f * st->zeta[i] + .15f * st->prior[i] + .075f * st->prior[i - 1] + .075f * st->prior[i + 1];
	for (i = N - 1; i < N + M; i++) st->zeta[i] = .7f * st->zeta[i] + .3f * st->prior[i];

	Zframe = 0;
	for (i = N; i < N + M; i++) Zframe = Zframe + st->zeta[i];
	Pframe = .1f + .899f * qcurve(Zframe / st->nbands);

	effective_echo_suppress = (int)((1.0f - Pframe) * st->echo_suppress + Pframe * st->echo_suppress_active);
	{
		float noise_floor = (float)exp(.2302585f * st->noise_suppress);
		float echo_floor = (float)exp(.2302585f * effective_echo_suppress);
		for (i = 0; i < M; i++)
			st->gain_floor[N + i] = (float)(sqrt(noise_floor * st->noise[N + i] + echo_floor * st->echo_noise[N + i]) /
			                                sqrt(1 + st->noise[N + i] + st->echo_noise[N + i]));
	}

	for (i = N; i < N + M; i++) {
		float theta, MM, prior_ratio, P1, q;
		prior_ratio = st->prior[i] / (st->prior[i] + 1.f);
		theta = prior_ratio * (1.f + st->post[i]);
		MM = hypergeom_gain(theta);
		st->gain[i] = prior_ratio * MM;
		if (st->gain[i] > 1.f) st->gain[i] = 1.f;
		st->old_ps[i] = .2f * st->old_ps[i] + (.8f * (st->gain[i] * st->gain[i])) * ps[i];
		P1 = .199f + .8f * qcurve(st->zeta[i]);
		q = 1.0f - Pframe * P1;
		st->gain2[i] = (float)(1 / (1.f + (q / (1.f - q)) * (1 + st->prior[i]) * exp(-theta)));
	}
	bank_compute_psd16(st->bank, st->gain2 + N, st->gain2);
	bank_compute_psd16(st->bank, st->gain + N, st->gain);
	bank_compute_psd16(st->bank, st->gain_floor + N, st->gain_floor);

	for (i = 0; i < N; i++) {
		float MM, theta, prior_ratio, tmp, p, g;
		prior_ratio = st->prior[i] / (st->prior[i] + 1.f);
		theta = prior_ratio * (1.f + st->post[i]);
		MM = hypergeom_gain(theta);
		g = prior_ratio * MM;
		if (g > 1.f) g = 1.f;
		p = st->gain2[i];
		if (.333f * g > st->gain[i]) g = 3 * st->gain[i];
		st->gain[i] = g;
		st->old_ps[i] = .2f * st->old_ps[i] + (.8f * (st->gain[i] * st->gain[i])) * ps[i];
		if (st->gain[i] < st->gain_floor[i]) st->gain[i] = st->gain_floor[i];
		tmp = p * (float)sqrt(st->gain[i]) + (1.0f - p) * (float)sqrt(st->gain_floor[i]);
		st->gain2[i] = tmp * tmp;
	}

	for (i = 1; i < N; i++) {
		st->ft[2 * i - 1] = st->gain2[i] * st->ft[2 * i - 1];
		st->ft[2 * i] = st->gain2[i] * st->ft[2 * i];
	}
	st->ft[0] = st->gain2[0] * st->ft[0];
	st->ft[2 * N - 1] = st->gain2[N - 1] * st->ft[2 * N - 1];

	orc_fft_inverse(st->fft, st->ft, st->frame);
	for (i = 0; i < 2 * N; i++) st->frame[i] = st->frame[i] * st->window[i];
	for (i = 0; i < N; i++) x[i] = word2int(st->outbuf[i] + st->frame[i]);
	for (i = 0; i < N; i++) st->outbuf[i] = st->frame[st->frame_size + i];
}
