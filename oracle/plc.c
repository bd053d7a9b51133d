/*
 * oracle/plc.c -- CPU oracle for MSGenericPLC: the FFT-based packet-loss
 * concealer the reference puts behind decoders without their own PLC (G.711, L16).
 *
 * TEST INFRASTRUCTURE ONLY (see ms2_oracle.h).  Parity unpinned: the sources restated here include
 * mediastreamer2 / bctoolbox headers and do not build in this image.
 *
 * Restates
 *   src/audiofilters/genericplc.h:26-44     constants
 *   src/audiofilters/genericplc.c:29-241    context, fftbf, generate_samples, the two buffer updates, transition mix
 *   src/audiofilters/msgenericplc.c:59-167  generic_plc_process (build without bcg729: comfort noise is silence)
 *   src/base/mscommon.c:315-366             MSConcealerContext
 * using the ms_fft / ms_ifft restatement of oracle/equalizer.c (kiss_fft with radix 2, 3, 4, 5 and generic stages).
 * Citations are relative to /root/reference.
 */
#include "ms2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TRANSITION_DELAY 5    /* ms, genericplc.h:27 */
#define PLC_DECREASE_START 100 /* ms, :34 */
#define MAX_PLC_LEN 150        /* ms, :35 */
#define ENERGY_ATTENUATION 0.85f /* :40 */

struct OrcPlc {
	int16_t *continuity; /* 2 * T samples */
	uint16_t nb;         /* plc_buffer_samples_nb */
	int16_t *hist;       /* plc_buffer: the last nb samples heard */
	float *window;
	int16_t *gen;        /* plc_out_buffer, 2 * nb samples */
	uint16_t index, used; /* plc_index, plc_samples_used: 16-bit counters as in genericplc.h:52-53 */
	OrcFft *fwd, *bwd;
	int rate;
};

/* generic_plc_create_context genericplc.c:29-69 */
OrcPlc *orc_plc_new(int rate) {
	OrcPlc *c = (OrcPlc *)calloc(1, sizeof(*c));
	int i;
	c->rate = rate;
	c->continuity = (int16_t *)calloc((size_t)(2 * rate * TRANSITION_DELAY / 1000), sizeof(int16_t));
	c->nb = (uint16_t)(((rate * 2 / 40) / 100) * 100); /* PLC_BUFFER_LEN is the token sequence 2 / 40 (genericplc.h:30) */
	c->hist = (int16_t *)calloc(c->nb, sizeof(int16_t));
	c->window = (float *)calloc(c->nb, sizeof(float));
	c->gen = (int16_t *)calloc(2 * (size_t)c->nb, sizeof(int16_t));
	c->fwd = orc_fft_new(c->nb);
	c->bwd = orc_fft_new(2 * c->nb);
	for (i = 0; i < c->nb; i++) c->window[i] = (float)(0.75 - 0.25 * cos(2 * 3.14159265 * i / (c->nb))); /* :63-65, PI as spelled there */
	return c;
}

void orc_plc_free(OrcPlc *c) {
	if (!c) return;
	free(c->continuity);
	free(c->hist);
	free(c->window);
	free(c->gen);
	orc_fft_free(c->fwd);
	orc_fft_free(c->bwd);
	free(c);
}

int orc_plc_info(const OrcPlc *c, int *nb, int *index, int *used) {
	if (nb) *nb = c->nb;
	if (index) *index = c->index;
	if (used) *used = c->used;
	return 0;
}

/* float -> int16 the way the x86-64 build does it: truncate to a 32-bit integer, keep the low half */
static int16_t to_i16(float v) { return (int16_t)(int32_t)v; }

/* generic_plc_transition_mix :233-241 */
void orc_plc_transition_mix(int16_t *inout, const int16_t *continuity, uint16_t n) {
	uint16_t i;
	for (i = 0; i < n; i++) {
		const float progress = ((float)i) / n;
		inout[i] = to_i16((float)continuity[i] * (1 - progress) + (float)inout[i] * progress);
	}
}

/* generic_plc_fftbf :83-121: window, forward FFT, every bin (as stored) moved to twice its index and scaled, inverse
 * FFT of twice the length, truncation to int16.  in and out may be the same array (out has 2 * n samples). */
static void fftbf(OrcPlc *c, const int16_t *in, int16_t *out, size_t n) {
	float *t = (float *)calloc(n, sizeof(float)), *f = (float *)calloc(n, sizeof(float));
	float *f2 = (float *)calloc(2 * n, sizeof(float)), *t2 = (float *)calloc(2 * n, sizeof(float));
	size_t i;
	for (i = 0; i < n; i++) t[i] = (float)in[i] * c->window[i];
	orc_fft_forward(c->fwd, t, f);
	for (i = 0; i < n; i++) {
		f2[2 * i] = f[i] * ENERGY_ATTENUATION;
		f2[2 * i + 1] = 0;
	}
	orc_fft_inverse(c->bwd, f2, t2);
	for (i = 0; i < 2 * n; i++) out[i] = to_i16(t2[i]);
	free(t);
	free(f);
	free(f2);
	free(t2);
}

/* generic_plc_update_plc_buffer :199-210 (lengths in samples here) */
void orc_plc_update_history(OrcPlc *c, const int16_t *data, size_t n) {
	if (n < c->nb) {
		memmove(c->hist, c->hist + n, (c->nb - n) * sizeof(int16_t));
		memcpy(c->hist + c->nb - n, data, n * sizeof(int16_t));
	} else {
		memcpy(c->hist, data + n - c->nb, c->nb * sizeof(int16_t));
	}
}

/* generic_plc_update_continuity_buffer :212-231: the block is delayed by TRANSITION_DELAY ms through the buffer */
void orc_plc_update_continuity(OrcPlc *c, int16_t *data, size_t n) {
	size_t T = (size_t)(c->rate * TRANSITION_DELAY / 1000);
	int16_t *tail;
	if (T > n) T = n;
	tail = (int16_t *)malloc(T * sizeof(int16_t) + 1);
	memcpy(tail, data + n - T, T * sizeof(int16_t));
	memmove(data + T, data, (n - T) * sizeof(int16_t));
	memcpy(data, c->continuity, T * sizeof(int16_t));
	memcpy(c->continuity, tail, T * sizeof(int16_t));
	free(tail);
}

/* generic_plc_generate_samples :123-197 */
void orc_plc_generate(OrcPlc *c, int16_t *data, uint16_t n) {
	const uint16_t T = (uint16_t)(c->rate * TRANSITION_DELAY / 1000);
	if (c->used >= MAX_PLC_LEN * c->rate / 1000) { /* :127-133: past 150 ms everything is silence */
		c->used = (uint16_t)(c->used + n);
		memset(data, 0, n * sizeof(int16_t));
		memset(c->continuity, 0, 2 * (size_t)T * sizeof(int16_t));
		return;
	}
	if (c->used == 0) { /* :136-144 first missing packet */
		fftbf(c, c->hist, c->gen, c->nb);
		orc_plc_transition_mix(c->gen, c->continuity, T);
	}
	if (c->index + n + T * 2 > 2 * c->nb) { /* :148-175 the generated signal runs out: extend it from itself */
		uint16_t ready = (uint16_t)(2 * c->nb - c->index - T);
		if (ready > n) ready = n;
		memcpy(data, c->gen + c->index, ready * sizeof(int16_t));
		memcpy(c->continuity, c->gen + c->index + ready, T * sizeof(int16_t));
		fftbf(c, c->gen, c->gen, c->nb);
		orc_plc_transition_mix(c->gen, c->continuity, T);
		if (n != ready) memcpy(data + ready, c->gen, (size_t)(n - ready) * sizeof(int16_t));
		c->index = (uint16_t)(n - ready);
		memcpy(c->continuity, c->gen + c->index, 2 * (size_t)T * sizeof(int16_t));
	} else { /* :176-184 */
		memcpy(data, c->gen + c->index, n * sizeof(int16_t));
		c->index = (uint16_t)(c->index + n);
		memcpy(c->continuity, c->gen + c->index, 2 * (size_t)T * sizeof(int16_t));
	}
	if (c->used + n > PLC_DECREASE_START * c->rate / 1000) { /* :187-203 fade to zero between 100 and 150 ms */
		int i = PLC_DECREASE_START * c->rate / 1000 - c->used;
		if (i < 0) i = 0;
		for (; i < n; i++) {
			if (c->used + i >= MAX_PLC_LEN * c->rate / 1000) data[i] = 0;
			else /* the literal 1.0 makes this a double expression, truncated straight to an integer */
				data[i] = (int16_t)(int32_t)((1.0 + ((float)(PLC_DECREASE_START * c->rate / 1000 - (c->used + i)) /
				                                     (float)((MAX_PLC_LEN - PLC_DECREASE_START) * c->rate / 1000))) *
				                             (float)data[i]);
		}
	}
	c->used = (uint16_t)(c->used + n);
}

/* The per-block body of generic_plc_process, msgenericplc.c:63-116, for a received block edited in place.
 * cng_resume: the filter was producing comfort noise (:76-89, silence without bcg729). */
void orc_plc_received(OrcPlc *c, int16_t *data, size_t n, int cng_resume) {
	const size_t T = (size_t)(c->rate * TRANSITION_DELAY / 1000);
	orc_plc_update_history(c, data, n);
	orc_plc_update_continuity(c, data, n);
	if (cng_resume && n >= 2 * T) { /* the reference's 80-sample zero array only covers rates up to 16 kHz; silence meant */
		int16_t *zeros = (int16_t *)calloc(T + 1, sizeof(int16_t));
		memcpy(data, zeros, T * sizeof(int16_t));
		orc_plc_transition_mix(data + T, zeros, (uint16_t)T);
		free(zeros);
	}
	if (c->used != 0) { /* :91-112 coming back from concealment */
		if (n >= 2 * T) orc_plc_transition_mix(data + T, c->continuity + T, (uint16_t)T);
		else orc_plc_transition_mix(c->continuity, c->continuity + T, (uint16_t)T);
	}
	c->index = 0;
	c->used = 0;
}

/* The concealment branch, msgenericplc.c:150-156: generate n samples and remember them as heard */
void orc_plc_conceal(OrcPlc *c, int16_t *data, uint16_t n) {
	orc_plc_generate(c, data, n);
	orc_plc_update_history(c, data, n);
}

/* -------------------------------------------------------------- MSConcealerContext, mscommon.c:315-366 */
void orc_concealer_init(OrcConcealer *o, uint32_t max_plc_time) {
	o->sample_time = -1;
	o->plc_start_time = -1;
	o->total_number_for_plc = 0;
	o->max_plc_time = max_plc_time;
}

uint32_t orc_concealer_inc_sample_time(OrcConcealer *o, uint64_t now, uint32_t increment, int got_packet) {
	uint32_t plc_duration = 0;
	if (o->sample_time == -1) o->sample_time = (int64_t)now;
	o->sample_time += increment;
	if (o->plc_start_time != -1 && got_packet) {
		plc_duration = (uint32_t)(now - (uint64_t)o->plc_start_time);
		o->plc_start_time = -1;
		if (plc_duration > o->max_plc_time) plc_duration = o->max_plc_time;
	}
	return plc_duration;
}

int orc_concealer_required(OrcConcealer *o, uint64_t now) {
	if (o->sample_time == -1) return 0;
	if ((uint64_t)o->sample_time <= now) {
		uint32_t plc_duration;
		if (o->plc_start_time == -1) o->plc_start_time = o->sample_time;
		plc_duration = (uint32_t)(now - (uint64_t)o->plc_start_time);
		if (plc_duration < o->max_plc_time) {
			o->total_number_for_plc++;
			return 1;
		}
		o->sample_time = -1;
		return 0;
	}
	return 0;
}
