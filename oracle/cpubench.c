/* oracle/cpubench.c -- TEST INFRASTRUCTURE (see ms2_oracle.h).
 * CPU-baseline loops for bench.py's `cpu_baseline` leg: one filter object per
 * stream, driven tick by tick like an MSTicker would call process()
 * (/root/reference/src/base/msticker.c:244-259), without sleeping.  Single
 * thread.  Returns elapsed seconds; *sink defeats dead-code elimination. */
#include "ms2_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* nstreams resamplers, nticks blocks of in_len samples each; input [nstreams][in_len] reused */
double orc_bench_resample(int nstreams, int in_len, int nticks, uint32_t in_rate, uint32_t out_rate,
                          const int16_t *in, long long *sink) {
	OrcResampler **r = (OrcResampler **)malloc(sizeof(*r) * (size_t)nstreams);
	uint32_t cap = orc_msresample_outcap((uint32_t)in_len, in_rate, out_rate);
	int16_t *out = (int16_t *)malloc(sizeof(int16_t) * cap);
	long long acc = 0;
	double t0, t1;
	int s, t;
	for (s = 0; s < nstreams; ++s) r[s] = orc_resampler_new(in_rate, out_rate, 3);
	t0 = now_s();
	for (t = 0; t < nticks; ++t) {
		for (s = 0; s < nstreams; ++s) {
			uint32_t il = (uint32_t)in_len, ol = cap;
			orc_resampler_process(r[s], in + (size_t)s * in_len, &il, out, &ol);
			acc += out[ol / 2];
		}
	}
	t1 = now_s();
	for (s = 0; s < nstreams; ++s) orc_resampler_free(r[s]);
	free(r);
	free(out);
	if (sink) *sink = acc;
	return t1 - t0;
}

double orc_bench_mixer(int nconf, int nmembers, int nsamples, int nticks, const int16_t *in, long long *sink) {
	uint8_t *ones = (uint8_t *)malloc((size_t)nmembers);
	float *gain = (float *)malloc(sizeof(float) * (size_t)nmembers);
	int16_t *out = (int16_t *)malloc(sizeof(int16_t) * (size_t)nmembers * nsamples);
	long long acc = 0;
	double t0, t1;
	int c, t, i;
	for (i = 0; i < nmembers; ++i) {
		ones[i] = 1;
		gain[i] = 1.0f;
	}
	t0 = now_s();
	for (t = 0; t < nticks; ++t)
		for (c = 0; c < nconf; ++c) {
			orc_mixer_tick(in + (size_t)c * nmembers * nsamples, ones, gain, ones, ones, nmembers, nsamples, 1, out,
			               NULL);
			acc += out[nsamples / 2];
		}
	t1 = now_s();
	free(ones);
	free(gain);
	free(out);
	if (sink) *sink = acc;
	return t1 - t0;
}

double orc_bench_volume(int nstreams, int nsamples, int nticks, int rate, int agc, const int16_t *in,
                        long long *sink) {
	OrcVolume *v = (OrcVolume *)malloc(sizeof(OrcVolume) * (size_t)nstreams);
	int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)nsamples);
	long long acc = 0;
	double t0, t1;
	int s, t;
	for (s = 0; s < nstreams; ++s) {
		orc_volume_init(&v[s]);
		orc_volume_set_rate(&v[s], rate);
		v[s].agc_enabled = agc;
	}
	t0 = now_s();
	for (t = 0; t < nticks; ++t)
		for (s = 0; s < nstreams; ++s) {
			memcpy(buf, in + (size_t)s * nsamples, sizeof(int16_t) * (size_t)nsamples);
			orc_volume_chunk(&v[s], buf, nsamples, 0.f);
			acc += buf[nsamples / 2];
		}
	t1 = now_s();
	free(v);
	free(buf);
	if (sink) *sink = acc;
	return t1 - t0;
}

double orc_bench_equalizer(int nstreams, int nsamples, int nticks, int rate, const int16_t *in, long long *sink) {
	OrcEqualizer **e = (OrcEqualizer **)malloc(sizeof(*e) * (size_t)nstreams);
	int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)nsamples);
	long long acc = 0;
	double t0, t1;
	int s, t;
	for (s = 0; s < nstreams; ++s) {
		e[s] = orc_equalizer_new(rate);
		orc_equalizer_set_gain(e[s], 1000, 2.0f, 500);
		orc_equalizer_design(e[s]);
	}
	t0 = now_s();
	for (t = 0; t < nticks; ++t)
		for (s = 0; s < nstreams; ++s) {
			memcpy(buf, in + (size_t)s * nsamples, sizeof(int16_t) * (size_t)nsamples);
			orc_equalizer_run(e[s], buf, nsamples);
			acc += buf[nsamples / 2];
		}
	t1 = now_s();
	for (s = 0; s < nstreams; ++s) orc_equalizer_free(e[s]);
	free(e);
	free(buf);
	if (sink) *sink = acc;
	return t1 - t0;
}

double orc_bench_scaler(int nframes, int sw, int sh, int dw, int dh, const uint8_t *src, long long *sink) {
	uint8_t *rgb = (uint8_t *)malloc((size_t)dw * dh * 3);
	size_t fb = (size_t)sw * (sh + (sh & 1)) * 3 / 2;
	long long acc = 0;
	double t0, t1;
	int f;
	(void)fb;
	t0 = now_s();
	for (f = 0; f < nframes; ++f) {
		orc_i420_scale_to_rgb24(src, sw, sh, rgb, dw, dh);
		acc += rgb[(size_t)dw * dh];
	}
	t1 = now_s();
	free(rgb);
	if (sink) *sink = acc;
	return t1 - t0;
}

/* nstreams cancellers + post-filters (speexec.c:297-298), nframes frames each; mic/ref [nstreams][frame] reused */
double orc_bench_aec(int nstreams, int frame, int filter_length, int rate, int nframes, const int16_t *mic,
                     const int16_t *ref, long long *sink) {
	OrcEcho **e = (OrcEcho **)malloc(sizeof(*e) * (size_t)nstreams);
	OrcPreproc **p = (OrcPreproc **)malloc(sizeof(*p) * (size_t)nstreams);
	int16_t *out = (int16_t *)malloc(sizeof(int16_t) * (size_t)frame);
	long long acc = 0;
	double t0, t1;
	int s, t;
	for (s = 0; s < nstreams; ++s) {
		e[s] = orc_echo_new(frame, filter_length, rate);
		p[s] = orc_preproc_new(frame, rate, e[s]);
	}
	t0 = now_s();
	for (t = 0; t < nframes; ++t)
		for (s = 0; s < nstreams; ++s) {
			orc_echo_cancel(e[s], mic + (size_t)s * frame, ref + (size_t)s * frame, out);
			orc_preproc_run(p[s], out);
			acc += out[frame / 2];
		}
	t1 = now_s();
	for (s = 0; s < nstreams; ++s) {
		orc_preproc_free(p[s]);
		orc_echo_free(e[s]);
	}
	free(e);
	free(p);
	free(out);
	if (sink) *sink = acc;
	return t1 - t0;
}

/* The same loop on `nthreads` host threads, streams split evenly (each thread owns its streams' filter objects, as one
 * MSTicker thread per group of calls would).  Returns the wall time of the slowest thread's span. */
#include <pthread.h>
typedef struct {
	int first, count, in_len, nticks;
	uint32_t in_rate, out_rate;
	const int16_t *in;
	long long acc;
	pthread_barrier_t *bar;
	double t0, t1;
} RsJob;

static void *rs_worker(void *arg) {
	RsJob *j = (RsJob *)arg;
	OrcResampler **r = (OrcResampler **)malloc(sizeof(*r) * (size_t)(j->count > 0 ? j->count : 1));
	uint32_t cap = orc_msresample_outcap((uint32_t)j->in_len, j->in_rate, j->out_rate);
	int16_t *out = (int16_t *)malloc(sizeof(int16_t) * cap);
	int s, t;
	long long acc = 0; /* thread-local: no shared cache line is written inside the timed loop */
	for (s = 0; s < j->count; ++s) r[s] = orc_resampler_new(j->in_rate, j->out_rate, 3);
	pthread_barrier_wait(j->bar);
	j->t0 = now_s();
	for (t = 0; t < j->nticks; ++t)
		for (s = 0; s < j->count; ++s) {
			uint32_t il = (uint32_t)j->in_len, ol = cap;
			orc_resampler_process(r[s], j->in + (size_t)(j->first + s) * j->in_len, &il, out, &ol);
			acc += out[ol / 2];
		}
	j->t1 = now_s();
	j->acc = acc;
	for (s = 0; s < j->count; ++s) orc_resampler_free(r[s]);
	free(r);
	free(out);
	return NULL;
}

double orc_bench_resample_mt(int nstreams, int in_len, int nticks, uint32_t in_rate, uint32_t out_rate, const int16_t *in,
                             int nthreads, long long *sink) {
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
	RsJob *jobs = (RsJob *)calloc((size_t)nthreads, sizeof(RsJob));
	pthread_barrier_t bar;
	double t0 = 1e300, t1 = 0;
	long long acc = 0;
	int i;
	pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
	for (i = 0; i < nthreads; ++i) {
		const int lo = (int)((long long)nstreams * i / nthreads), hi = (int)((long long)nstreams * (i + 1) / nthreads);
		jobs[i].first = lo;
		jobs[i].count = hi - lo;
		jobs[i].in_len = in_len;
		jobs[i].nticks = nticks;
		jobs[i].in_rate = in_rate;
		jobs[i].out_rate = out_rate;
		jobs[i].in = in;
		jobs[i].bar = &bar;
		pthread_create(&th[i], NULL, rs_worker, &jobs[i]);
	}
	for (i = 0; i < nthreads; ++i) {
		pthread_join(th[i], NULL);
		if (jobs[i].t0 < t0) t0 = jobs[i].t0;
		if (jobs[i].t1 > t1) t1 = jobs[i].t1;
		acc += jobs[i].acc;
	}
	pthread_barrier_destroy(&bar);
	free(th);
	free(jobs);
	if (sink) *sink = acc;
	return t1 - t0;
}

/* G.711 sample loops of MSAlawDec / MSAlawEnc (alaw.c:213-217, :77-82) timed through a function pointer, so that the
 * conversion can be the reference's own Snack_* from oracle/_ref/libg711_ref.so ("reference") or this oracle's ("port"):
 * either way one external call per sample, as in the reference.  Returns seconds for `reps` passes over n samples. */
double orc_bench_g711_decode(short (*fn)(unsigned char), const uint8_t *codes, size_t n, int reps, long long *sink) {
	long long acc = 0;
	const double t0 = now_s();
	for (int r = 0; r < reps; ++r)
		for (size_t i = 0; i < n; ++i) acc += fn(codes[i]);
	const double t1 = now_s();
	if (sink) *sink = acc;
	return t1 - t0;
}
double orc_bench_g711_encode(unsigned char (*fn)(short), const int16_t *pcm, size_t n, int reps, long long *sink) {
	long long acc = 0;
	const double t0 = now_s();
	for (int r = 0; r < reps; ++r)
		for (size_t i = 0; i < n; ++i) acc += fn(pcm[i]);
	const double t1 = now_s();
	if (sink) *sink = acc;
	return t1 - t0;
}
