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

/* ---- the north_star chain on the CPU: what one MSTicker thread does per 10 ms tick for a group of conferences.
 * Per call leg: MSResample 16k -> 48k (msresample.c:122-179) -> MSSpeexEC at 48 kHz (256-sample frames through the
 * filter's bufferizers, zero reference when the far end is short: speexec.c:223-305; canceller + post-filter) ->
 * MSVolume with AGC on 10 ms chunks (msvolume.c:471-514); per conference of `members` legs: MSAudioMixer in conference
 * mode (audiomixer.c:288-346).  mic16 [nstreams][160] and ref48 [nstreams][480] are reused every tick. */
typedef struct {
	int first_conf, nconf, members, nticks, tail_ms;
	const int16_t *mic16, *ref48;
	pthread_barrier_t *bar;
	long long acc;
	double t0, t1;
} ChainJob;

typedef struct {
	int16_t buf[2048];
	int n;
} ChainFifo;

static void cf_push(ChainFifo *f, const int16_t *x, int n) {
	if (f->n + n > 2048) return;
	memcpy(f->buf + f->n, x, sizeof(int16_t) * (size_t)n);
	f->n += n;
}
static int cf_pop(ChainFifo *f, int16_t *x, int n) { /* all-or-nothing, msqueue.c:83 */
	if (f->n < n) return 0;
	memcpy(x, f->buf, sizeof(int16_t) * (size_t)n);
	f->n -= n;
	memmove(f->buf, f->buf + n, sizeof(int16_t) * (size_t)f->n);
	return 1;
}

static void *chain_worker(void *arg) {
	ChainJob *j = (ChainJob *)arg;
	const int rate = 48000, F = 256, ns = 480, mm = j->members;
	const int nst = j->nconf * mm;
	OrcResampler **rs = (OrcResampler **)malloc(sizeof(*rs) * (size_t)(nst > 0 ? nst : 1));
	OrcEcho **ec = (OrcEcho **)malloc(sizeof(*ec) * (size_t)(nst > 0 ? nst : 1));
	OrcPreproc **pp = (OrcPreproc **)malloc(sizeof(*pp) * (size_t)(nst > 0 ? nst : 1));
	OrcVolume *vol = (OrcVolume *)malloc(sizeof(OrcVolume) * (size_t)(nst > 0 ? nst : 1));
	ChainFifo *fm = (ChainFifo *)calloc((size_t)(nst > 0 ? nst : 1), sizeof(ChainFifo));
	ChainFifo *fr = (ChainFifo *)calloc((size_t)(nst > 0 ? nst : 1), sizeof(ChainFifo));
	ChainFifo *fo = (ChainFifo *)calloc((size_t)(nst > 0 ? nst : 1), sizeof(ChainFifo));
	int16_t *tick = (int16_t *)malloc(sizeof(int16_t) * (size_t)mm * ns);
	int16_t *mixed = (int16_t *)malloc(sizeof(int16_t) * (size_t)mm * ns);
	uint8_t *ones = (uint8_t *)malloc((size_t)mm);
	float *gain = (float *)malloc(sizeof(float) * (size_t)mm);
	int16_t up[488], mf[256], rf[256], cl[256];
	long long acc = 0;
	int s, t, c, m;
	for (m = 0; m < mm; ++m) ones[m] = 1, gain[m] = 1.0f;
	for (s = 0; s < nst; ++s) {
		rs[s] = orc_resampler_new(16000, (uint32_t)rate, 3);
		ec[s] = orc_echo_new(F, j->tail_ms * rate / 1000, rate);
		pp[s] = orc_preproc_new(F, rate, ec[s]);
		orc_volume_init(&vol[s]);
		orc_volume_set_rate(&vol[s], rate);
		vol[s].agc_enabled = 1;
	}
	pthread_barrier_wait(j->bar);
	j->t0 = now_s();
	for (t = 0; t < j->nticks; ++t)
		for (c = 0; c < j->nconf; ++c) {
			for (m = 0; m < mm; ++m) {
				const int ls = c * mm + m;
				const size_t gs = (size_t)(j->first_conf + c) * mm + m;
				uint32_t il = 160, ol = 488;
				orc_resampler_process(rs[ls], j->mic16 + gs * 160, &il, up, &ol);
				cf_push(&fm[ls], up, (int)ol);
				cf_push(&fr[ls], j->ref48 + gs * ns, ns);
				while (cf_pop(&fm[ls], mf, F)) {
					if (!cf_pop(&fr[ls], rf, F)) memset(rf, 0, sizeof(rf)); /* speexec.c:261-272 */
					orc_echo_cancel(ec[ls], mf, rf, cl);
					orc_preproc_run(pp[ls], cl);
					cf_push(&fo[ls], cl, F);
				}
				if (!cf_pop(&fo[ls], tick + (size_t)m * ns, ns)) memset(tick + (size_t)m * ns, 0, sizeof(int16_t) * ns);
				orc_volume_chunk(&vol[ls], tick + (size_t)m * ns, ns, 0.f);
			}
			orc_mixer_tick(tick, ones, gain, ones, ones, mm, ns, 1, mixed, NULL);
			acc += mixed[ns / 2];
		}
	j->t1 = now_s();
	j->acc = acc;
	for (s = 0; s < nst; ++s) {
		orc_preproc_free(pp[s]);
		orc_echo_free(ec[s]);
		orc_resampler_free(rs[s]);
	}
	free(rs), free(ec), free(pp), free(vol), free(fm), free(fr), free(fo), free(tick), free(mixed), free(ones), free(gain);
	return NULL;
}

/* nconf conferences of `members` legs for nticks ticks on nthreads threads (whole conferences per thread).
 * Returns the wall time from the first thread's start to the last thread's end. */
double orc_bench_chain_mt(int nconf, int members, int nticks, int tail_ms, const int16_t *mic16, const int16_t *ref48,
                          int nthreads, long long *sink) {
	pthread_t *th;
	ChainJob *jobs;
	pthread_barrier_t bar;
	double t0 = 1e300, t1 = 0;
	long long acc = 0;
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nconf) nthreads = nconf;
	th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
	jobs = (ChainJob *)calloc((size_t)nthreads, sizeof(ChainJob));
	pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
	for (i = 0; i < nthreads; ++i) {
		const int lo = (int)((long long)nconf * i / nthreads), hi = (int)((long long)nconf * (i + 1) / nthreads);
		jobs[i].first_conf = lo;
		jobs[i].nconf = hi - lo;
		jobs[i].members = members;
		jobs[i].nticks = nticks;
		jobs[i].tail_ms = tail_ms;
		jobs[i].mic16 = mic16;
		jobs[i].ref48 = ref48;
		jobs[i].bar = &bar;
		pthread_create(&th[i], NULL, chain_worker, &jobs[i]);
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
