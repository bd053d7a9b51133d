/*
 * wav_echo_canceller.c -- "the same WAV inputs" through the kernel library, in plain C: the scene of the reference's
 * echo-canceller tester (tester/mediastreamer2_aec3_tester.c:380-440) without a filter graph.
 *
 *   wav_echo_canceller far.wav near.wav echo.wav out.wav [rate] [delay_ms] [tail_ms]
 *
 * far.wav is played to the canceller's reference input; near.wav + echo.wav, both started delay_ms later, are summed
 * with the mixer's symmetric saturation (src/audiofilters/audiomixer.c:33-44) into its microphone input; the cleaned
 * microphone signal is written to out.wav at the files' rate.  With rate = 48000 and 16 kHz files every input goes
 * through the resampler first and the output back (the tester's 48 kHz case, :743-758).  Frames are fed back to back,
 * as MSSpeexEC does once both of its pins run (src/audiofilters/speexec.c:256-305).
 *
 * Build: gcc -std=c99 -Iinclude examples/wav_echo_canceller.c -Lmediastreamer2_amd -lmsmi355x -o wav_echo_canceller
 * No GPU -> exits with the library's error (there is no CPU fallback).
 */
#include "ms2_mediaio.h"
#include "msmi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MUST(call)                                                       \
	do {                                                                 \
		if ((call) != MI_OK) {                                           \
			fprintf(stderr, "%s: %s\n", #call, mi_last_error());         \
			return 1;                                                    \
		}                                                                \
	} while (0)

static int16_t sat(int v) { return (int16_t)(v > 32767 ? 32767 : (v < -32767 ? -32767 : v)); }

/* a whole track through a resampler in 10 ms blocks; returns the produced length */
static int resample_track(mi_resampler *r, const int16_t *in, int n, int in_rate, int16_t *out, int out_cap) {
	const int blk = in_rate / 100, cap = mi_resampler_out_capacity(r, blk);
	int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)cap);
	int32_t got = 0;
	int produced = 0;
	for (int at = 0; at + blk <= n; at += blk) {
		if (mi_resampler_process_host(r, in + at, blk, blk, tmp, cap, &got) != MI_OK || produced + got > out_cap) {
			free(tmp);
			return -1;
		}
		memcpy(out + produced, tmp, sizeof(int16_t) * (size_t)got);
		produced += got;
	}
	free(tmp);
	return produced;
}

int main(int argc, char **argv) {
	if (argc < 5) {
		fprintf(stderr, "usage: %s far.wav near.wav echo.wav out.wav [rate=file rate] [delay_ms=100] [tail_ms=250]\n", argv[0]);
		return 2;
	}
	ms2_wav far, near, echo;
	if (ms2_wav_read(argv[1], &far) || ms2_wav_read(argv[2], &near) || ms2_wav_read(argv[3], &echo)) {
		fprintf(stderr, "cannot read the input recordings (PCM16 WAV expected)\n");
		return 2;
	}
	if (far.nchannels != 1 || near.nchannels != 1 || echo.nchannels != 1 || far.rate != near.rate || far.rate != echo.rate) {
		fprintf(stderr, "mono recordings of one rate expected\n");
		return 2;
	}
	const int frate = far.rate;
	const int rate = argc > 5 ? atoi(argv[5]) : frate;
	const int delay_ms = argc > 6 ? atoi(argv[6]) : 100;
	const int tail_ms = argc > 7 ? atoi(argv[7]) : 250; /* speexec.c:82 */
	const int d = delay_ms * frate / 1000;

	/* the three tracks on one time line, padded to whole 10 ms blocks */
	int n = far.nsamples;
	if (near.nsamples + d > n) n = near.nsamples + d;
	if (echo.nsamples + d > n) n = echo.nsamples + d;
	n = (n + frate / 100 - 1) / (frate / 100) * (frate / 100);
	int16_t *tfar = (int16_t *)calloc((size_t)n, 2), *tnear = (int16_t *)calloc((size_t)n, 2), *techo = (int16_t *)calloc((size_t)n, 2);
	memcpy(tfar, far.samples, 2 * (size_t)far.nsamples);
	memcpy(tnear + d, near.samples, 2 * (size_t)near.nsamples);
	memcpy(techo + d, echo.samples, 2 * (size_t)echo.nsamples);

	mi_ctx *ctx = NULL;
	MUST(mi_ctx_create(0, NULL, &ctx));

	/* at the canceller's rate */
	int m = n;
	int16_t *cfar = tfar, *cnear = tnear, *cecho = techo;
	if (rate != frate) {
		const int cap = (int)((long long)n * rate / frate) + 1024;
		int16_t **src[3] = {&cfar, &cnear, &cecho};
		for (int k = 0; k < 3; ++k) {
			mi_resampler *r = NULL;
			MUST(mi_resampler_create(ctx, 1, (uint32_t)frate, (uint32_t)rate, 3, &r));
			int16_t *o = (int16_t *)calloc((size_t)cap, 2);
			m = resample_track(r, *src[k], n, frate, o, cap);
			mi_resampler_destroy(r);
			if (m < 0) {
				fprintf(stderr, "resampler: %s\n", mi_last_error());
				return 1;
			}
			*src[k] = o;
		}
	}
	int16_t *mic = (int16_t *)malloc(2 * (size_t)m);
	for (int i = 0; i < m; ++i) mic[i] = sat((int)cnear[i] + (int)cecho[i]);

	/* the canceller: speexec.c:171-180 frame size, tail_ms of filter, post-filter on (:297-298) */
	const int F = mi_aec_framesize(64, rate);
	mi_aec *aec = NULL;
	MUST(mi_aec_create(ctx, 1, rate, F, tail_ms * rate / 1000, &aec));
	int16_t *clean = (int16_t *)calloc((size_t)m + (size_t)F, 2);
	const int nframes = m / F;
	for (int k = 0; k < nframes; ++k)
		MUST(mi_aec_process_host(aec, mic + (size_t)k * F, cfar + (size_t)k * F, clean + (size_t)k * F, F, NULL, MI_AEC_POSTFILTER));
	mi_aec_destroy(aec);

	/* back at the files' rate */
	int16_t *out = clean;
	int nout = nframes * F;
	if (rate != frate) {
		mi_resampler *r = NULL;
		MUST(mi_resampler_create(ctx, 1, (uint32_t)rate, (uint32_t)frate, 3, &r));
		const int cap = (int)((long long)nout * frate / rate) + 1024;
		out = (int16_t *)calloc((size_t)cap, 2);
		nout = resample_track(r, clean, nout, rate, out, cap);
		mi_resampler_destroy(r);
		if (nout < 0) {
			fprintf(stderr, "resampler: %s\n", mi_last_error());
			return 1;
		}
	}
	if (ms2_wav_write(argv[4], frate, 1, out, nout) != 0) {
		fprintf(stderr, "cannot write %s\n", argv[4]);
		return 1;
	}
	mi_ctx_destroy(ctx);
	printf("ok %d samples at %d Hz (canceller at %d Hz, frame %d, tail %d ms)\n", nout, frate, rate, F, tail_ms);
	return 0;
}
