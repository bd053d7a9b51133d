/* plugin_bench.c -- TEST / MEASUREMENT INFRASTRUCTURE (not part of the product).
 *
 * How many full call legs does a mediastreamer2-shaped process carry through the DROP-IN PLUGIN?  N legs of
 *
 *     mic source (16 kHz) -> MSResample 16k->48k -> MSSpeexEC (128 ms tail) pin 1 -> MSVolume (AGC) -> MSAudioMixer pin k
 *     far-end source (48 kHz) --------------------> MSSpeexEC pin 0 -> speaker sink;   mixer pin k -> sink
 *
 * (the sending side of src/voip/audiostream.c:1798-1810 in front of a conference mixer, conferences of 32) on T tickers of
 * the test runtime (tests/host/ms2shim.c: the reference's one-thread-per-MSTicker model, src/base/msticker.c:448-524), the
 * filters created by id through the factory after libmsmi355xfilters_init() registered the plugin's descriptors -- exactly
 * what a mediastreamer2 process does.  Every tick every ticker thread runs its postponed tasks (the plugin's flush) and
 * walks its graphs; all tickers tick together (a barrier stands in for the wall clock), a tick costs what the slowest
 * ticker needs.  Prints one JSON object.
 *
 *   plugin_bench <plugin.so> <legs> <tickers> <ticks> <warmup> [members=32]
 */
#define _GNU_SOURCE /* RUSAGE_THREAD */
#include "../../include/ms2_plugin_abi.h"

#include <dlfcn.h>
#include <execinfo.h>
#include <malloc.h>
#include <signal.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

MSFactory *ms_factory_new(void);
int ms_factory_load_plugin(MSFactory *f, const char *path);
MSFilter *ms_factory_create_filter(MSFactory *f, MSFilterId id);
int ms_filter_link(MSFilter *f1, int pin1, MSFilter *f2, int pin2);
int ms_filter_call_method(MSFilter *f, unsigned int id, void *arg);
MSTicker *ms_ticker_new(void);
int ms_ticker_attach(MSTicker *t, MSFilter *f);
int ms_ticker_detach(MSTicker *t, MSFilter *f);
void ms_ticker_step(MSTicker *t);
void ms2shim_register_test_filters(MSFactory *f);
MSFilter *ms2shim_new_source(MSFactory *f);
MSFilter *ms2shim_new_sink(MSFactory *f);
MSFilter *ms2shim_new_pass(MSFactory *f);
void ms2shim_sink_set_discard(MSFilter *f, int on);
void ms2shim_source_set_loop(MSFilter *src, const void *ring, size_t block_bytes, int nblocks, int phase);
void ms2shim_ticker_last_step(MSTicker *t, uint64_t *tasks_ns, uint64_t *step_ns);
uint64_t ms2shim_sink_sum(MSFilter *sink);
int ms2shim_ticker_profile(MSTicker *t, int *ids, uint64_t *ns, int cap, int *max_id, uint64_t *max_ns);
size_t ms2shim_sink_size(MSFilter *sink);
int ms2shim_sink_blocks(MSFilter *sink);
int ms2shim_equalizer_set_active(MSFilter *eq, int active);

#define RING 16
static int16_t g_mic[RING][160], g_far[RING][480], g_pcm8[RING][80];
static uint8_t g_codes8[RING][80];

typedef struct {
	MSTicker *ticker;
	MSFilter **mixers;
	MSFilter *probe_out; /* one mixer output sink: did audio arrive? */
	int nconf, index;
	double *late_ms;                     /* paced: how far behind its schedule the step STARTED */
	double *step_ms, *task_ms, *cpu_ms; /* per tick: wall time of the step, of its postponed tasks, CPU time of the thread */
	MSFilter **heads;        /* PLUGIN_BENCH_SHAPE nomixer: every leg's source, the root its graph is attached by */
	MSFilter **outs, **spks; /* PLUGIN_BENCH_CHECKSUM=1: every leg's two sinks (mix back to the leg, speaker pin) */
	double *warm_ms;                     /* the steps from the attach on (the warm-up): reported apart as `from_attach` */
	double slowest_ms;
	int slowest_tick, prof_n, prof_ids[16], max_id;
	uint64_t prof_ns[16], max_ns;
	int tot_ids[16];        /* MS2SHIM_PROFILE=1: every timed step's process() time by filter id, summed */
	uint64_t tot_ns[16];
	int *nvcsw, *nivcsw, *minflt;        /* per tick: voluntary / involuntary context switches, minor page faults of the thread */
	double *churn_ms;                    /* PLUGIN_BENCH_CHURN: what the re-plumbing in front of this tick's step took (0: none) */
	double attach_ms;                    /* what ms_ticker_attach of this ticker's graphs took (every filter's preprocess: the plugin fuses and opens its banks there) */
	int churn_next;                      /* ... the conference (or leg) re-plumbed next */
} TickerJob;

static MSFactory *g_fac;
static int g_members = 32, g_ticks, g_warmup, g_tickers;
static int g_paced;            /* PLUGIN_BENCH_PACED=1: every ticker fires at t0 + k x 10 ms of wall time, as an MSTicker does */
static volatile uint64_t g_t0; /* ... the schedule's origin (ns, CLOCK_MONOTONIC) */
static volatile uint64_t g_w0; /* ... and the warm-up's */
static pthread_barrier_t g_bar;

static int g_profile, g_checksum;
static int g_nors, g_noagc, g_nomixer; /* PLUGIN_BENCH_SHAPE: words of "nors noagc nomixer" -- the leg without MSResample / without AGC / without a conference mixer */
static int g_server; /* ... "server": a conference server's REMOTE members (audioconference.c:121-179,209-257): 8 kHz source (decoder .. dtmfgen) -> MSVolume -> in_resampler -> pin -> out_resampler -> MSUlawEnc -> sink, no canceller */
static int g_dec; /* ... with "server": "dec" -- the sources hand over G.711 PACKETS (rtprecv) and MSUlawDec of the plugin heads every leg */
static int g_astream; /* ... "astream": full-duplex narrow-band AudioStreams as audiostream.c:1798-1832 plumbs them, the card at 8 kHz: PCMU packets ->
                        MSUlawDec -> MSGenericPLC -> dtmfgen (the application's) -> volrecv -> recv_tee -> MSSpeexEC pin 0 -> speaker;  microphone -> MSSpeexEC
                        pin 1 -> volsend -> dtmfgen_rtp -> MSUlawEnc -> packets.  The sending side fuses leg by leg, the receiving side runs as facades. */
static int g_default; /* ... "astream default": the reference's DEFAULT features (AUDIO_STREAM_FEATURE_ALL, audiostream.c:1585-1588,1754-1772,1807,1815) with a telephone-event payload
                        negotiated: packets -> MSUlawDec -> local_mixer -> MSGenericPLC -> MSAudioFlowControl -> dtmfgen -> volrecv -> recv_tee -> MSSpeexEC pin 0;  microphone ->
                        MSSpeexEC pin 1 -> volsend -> outbound_mixer -> MSUlawEnc (no dtmfgen_rtp, :1396-1404): both directions device-resident, the encoder in the leg's batch;
                        and BOTH equalizers of AUDIO_STREAM_FEATURE_EQUALIZER, neither active (:1623-1640): mic_equalizer in front of pin 1, spk_equalizer behind recv_tee */
static int g_wb; /* ... "server wb": the server's conference runs at 16 kHz, its G.711 endpoints at 8 kHz -- both resamplers of every member work (audioconference.c:209-257) */
static int g_eq; /* ... "eq": a mic_equalizer between MSResample and MSSpeexEC (audiostream.c:1801), a response of its own per leg */
static int g_el; /* ... "el": the echo limiter on (audiostream.c:2236-2240): volrecv upstream of the canceller's far end, volsend's peer (with nomixer) */
static int g_eprs; /* ... "eprs": every pin behind an in_resampler, in front of an out_resampler, as MSAudioConference plumbs its endpoints (audioconference.c:209-257) */
static double now_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static double thread_cpu_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static void call_int(MSFilter *f, unsigned id, int v) { ms_filter_call_method(f, id, &v); }

static void build(TickerJob *j) {
	j->ticker = ms_ticker_new();
	j->mixers = (MSFilter **)calloc((size_t)j->nconf, sizeof(MSFilter *));
	j->outs = (MSFilter **)calloc((size_t)j->nconf * (size_t)g_members, sizeof(MSFilter *));
	j->heads = (MSFilter **)calloc((size_t)j->nconf * (size_t)g_members, sizeof(MSFilter *));
	j->spks = (MSFilter **)calloc((size_t)j->nconf * (size_t)g_members, sizeof(MSFilter *));
	for (int c = 0; c < j->nconf; ++c) {
		MSFilter *mx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
		call_int(mx, MS_FILTER_SET_SAMPLE_RATE, 48000);
		call_int(mx, MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
		j->mixers[c] = mx;
		if (g_server) { /* the conference at the G.711 endpoints' rate: both resamplers forward (msresample.c:126-135) */
			call_int(mx, MS_FILTER_SET_SAMPLE_RATE, g_wb ? 16000 : 8000);
			for (int k = 0; k < g_members; ++k) {
				MSFilter *src = ms2shim_new_source(g_fac), *out = ms2shim_new_sink(g_fac), *vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
				MSFilter *in_rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID), *out_rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
				MSFilter *enc = ms_factory_create_filter(g_fac, MS_ULAW_ENC_ID);
				const int leg = (j->index * j->nconf + c) * g_members + k;
				MSFilter *dec = g_dec ? ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID) : NULL;
				if (dec) ms2shim_source_set_loop(src, g_codes8, sizeof(g_codes8[0]), RING, leg);
				else ms2shim_source_set_loop(src, g_pcm8, sizeof(g_pcm8[0]), RING, leg);
				ms2shim_sink_set_discard(out, g_checksum ? 2 : 1);
				j->outs[c * g_members + k] = out;
				j->spks[c * g_members + k] = NULL;
				call_int(vol, MS_FILTER_SET_SAMPLE_RATE, 8000);
				call_int(in_rs, MS_FILTER_SET_SAMPLE_RATE, 8000), call_int(in_rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, g_wb ? 16000 : 8000);
				call_int(out_rs, MS_FILTER_SET_SAMPLE_RATE, g_wb ? 16000 : 8000), call_int(out_rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 8000);
				if (dec) ms_filter_link(src, 0, dec, 0), ms_filter_link(dec, 0, vol, 0);
				else ms_filter_link(src, 0, vol, 0);
				ms_filter_link(vol, 0, in_rs, 0), ms_filter_link(in_rs, 0, mx, k);
				ms_filter_link(mx, k, out_rs, 0), ms_filter_link(out_rs, 0, enc, 0), ms_filter_link(enc, 0, out, 0);
				if (c == 0 && k == 0) j->probe_out = out;
			}
			continue;
		}
		for (int k = 0; k < g_members && g_astream; ++k) {
			MSFilter *mic = ms2shim_new_source(g_fac), *far = ms2shim_new_source(g_fac), *spk = ms2shim_new_sink(g_fac), *out = ms2shim_new_sink(g_fac);
			MSFilter *ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID), *vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID), *volrecv = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			MSFilter *dec = ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID), *plc = ms_factory_create_filter(g_fac, MS_GENERIC_PLC_ID), *enc = ms_factory_create_filter(g_fac, MS_ULAW_ENC_ID);
			MSFilter *dtmfgen = ms2shim_new_pass(g_fac), *recv_tee = ms2shim_new_pass(g_fac), *dtmfgen_rtp = g_default ? NULL : ms2shim_new_pass(g_fac);
			MSFilter *local_mixer = g_default ? ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID) : NULL, *outbound_mixer = g_default ? ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID) : NULL;
			MSFilter *flowctl = g_default ? ms_factory_create_filter(g_fac, MS_AUDIO_FLOW_CONTROL_ID) : NULL;
			const int leg = (j->index * j->nconf + c) * g_members + k;
			ms2shim_source_set_loop(mic, g_pcm8, sizeof(g_pcm8[0]), RING, leg);
			ms2shim_source_set_loop(far, g_codes8, sizeof(g_codes8[0]), RING, leg * 7);
			ms2shim_sink_set_discard(spk, g_checksum ? 2 : 1);
			ms2shim_sink_set_discard(out, g_checksum ? 2 : 1);
			j->outs[c * g_members + k] = out;
			j->spks[c * g_members + k] = spk;
			j->heads[c * g_members + k] = mic;
			call_int(ec, MS_FILTER_SET_SAMPLE_RATE, 8000);
			call_int(ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 128);
			call_int(vol, MS_FILTER_SET_SAMPLE_RATE, 8000);
			call_int(volrecv, MS_FILTER_SET_SAMPLE_RATE, 8000);
			call_int(plc, MS_FILTER_SET_SAMPLE_RATE, 8000);
			if (g_default) {
				call_int(local_mixer, MS_FILTER_SET_SAMPLE_RATE, 8000), call_int(outbound_mixer, MS_FILTER_SET_SAMPLE_RATE, 8000);
				call_int(flowctl, MS_FILTER_SET_SAMPLE_RATE, 8000), call_int(flowctl, MS_FILTER_SET_NCHANNELS, 1);
				ms_filter_link(far, 0, dec, 0), ms_filter_link(dec, 0, local_mixer, 0), ms_filter_link(local_mixer, 0, plc, 0), ms_filter_link(plc, 0, flowctl, 0);
				ms_filter_link(flowctl, 0, dtmfgen, 0), ms_filter_link(dtmfgen, 0, volrecv, 0);
			} else {
				ms_filter_link(far, 0, dec, 0), ms_filter_link(dec, 0, plc, 0), ms_filter_link(plc, 0, dtmfgen, 0), ms_filter_link(dtmfgen, 0, volrecv, 0);
			}
			if (g_default) {
				MSFilter *mic_eq = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID), *spk_eq = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID);
				call_int(mic_eq, MS_FILTER_SET_SAMPLE_RATE, 8000), call_int(spk_eq, MS_FILTER_SET_SAMPLE_RATE, 8000);
				ms2shim_equalizer_set_active(mic_eq, 0), ms2shim_equalizer_set_active(spk_eq, 0);
				ms_filter_link(volrecv, 0, recv_tee, 0), ms_filter_link(recv_tee, 0, spk_eq, 0), ms_filter_link(spk_eq, 0, ec, 0), ms_filter_link(ec, 0, spk, 0);
				ms_filter_link(mic, 0, mic_eq, 0), ms_filter_link(mic_eq, 0, ec, 1), ms_filter_link(ec, 1, vol, 0);
			} else {
				ms_filter_link(volrecv, 0, recv_tee, 0), ms_filter_link(recv_tee, 0, ec, 0), ms_filter_link(ec, 0, spk, 0);
				ms_filter_link(mic, 0, ec, 1), ms_filter_link(ec, 1, vol, 0);
			}
			if (g_default) ms_filter_link(vol, 0, outbound_mixer, 0), ms_filter_link(outbound_mixer, 0, enc, 0);
			else ms_filter_link(vol, 0, dtmfgen_rtp, 0), ms_filter_link(dtmfgen_rtp, 0, enc, 0);
			ms_filter_link(enc, 0, out, 0);
			if (c == 0 && k == 0) j->probe_out = out;
		}
		for (int k = 0; k < g_members && !g_astream; ++k) {
			MSFilter *mic = ms2shim_new_source(g_fac), *far = ms2shim_new_source(g_fac), *spk = ms2shim_new_sink(g_fac), *out = ms2shim_new_sink(g_fac);
			MSFilter *rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID), *ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID);
			MSFilter *vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			const int leg = (j->index * j->nconf + c) * g_members + k;
			if (g_nors) ms2shim_source_set_loop(mic, g_far, sizeof(g_far[0]), RING, leg * 3 + 1); /* a 48 kHz microphone: the card runs at the canceller's rate */
			else ms2shim_source_set_loop(mic, g_mic, sizeof(g_mic[0]), RING, leg);
			ms2shim_source_set_loop(far, g_far, sizeof(g_far[0]), RING, leg * 7);
			ms2shim_sink_set_discard(spk, g_checksum ? 2 : 1);
			ms2shim_sink_set_discard(out, g_checksum ? 2 : 1);
			j->outs[c * g_members + k] = out;
			j->spks[c * g_members + k] = spk;
			call_int(rs, MS_FILTER_SET_SAMPLE_RATE, 16000);
			call_int(rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
			call_int(ec, MS_FILTER_SET_SAMPLE_RATE, 48000);
			call_int(ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 128);
			call_int(vol, MS_FILTER_SET_SAMPLE_RATE, 48000);
			if (!g_noagc) call_int(vol, MS_VOLUME_ENABLE_AGC, 1);
			if (g_nors) {
				ms_filter_link(mic, 0, ec, 1);
			} else if (g_eq) {
				MSFilter *eq = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID);
				MSEqualizerGain eg;
				call_int(eq, MS_FILTER_SET_SAMPLE_RATE, 48000);
				eg.frequency = 800.f + 50.f * (float)(leg % 32), eg.gain = 2.0f, eg.width = 500.f;
				ms_filter_call_method(eq, MS_EQUALIZER_SET_GAIN, &eg);
				ms_filter_link(mic, 0, rs, 0);
				ms_filter_link(rs, 0, eq, 0);
				ms_filter_link(eq, 0, ec, 1);
			} else {
				ms_filter_link(mic, 0, rs, 0);
				ms_filter_link(rs, 0, ec, 1);
			}
			ms_filter_link(ec, 1, vol, 0);
			if (g_nomixer) { /* an AudioStream's sending side: MSVolume's blocks go straight on (to the encoder; here a sink) */
				ms_filter_link(vol, 0, out, 0);
				j->heads[c * g_members + k] = mic;
			} else if (g_eprs) { /* both at the conference's rate: they forward (msresample.c:126-135) */
				MSFilter *in_rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID), *out_rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
				call_int(in_rs, MS_FILTER_SET_SAMPLE_RATE, 48000), call_int(in_rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
				call_int(out_rs, MS_FILTER_SET_SAMPLE_RATE, 48000), call_int(out_rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
				ms_filter_link(vol, 0, in_rs, 0), ms_filter_link(in_rs, 0, mx, k);
				ms_filter_link(mx, k, out_rs, 0), ms_filter_link(out_rs, 0, out, 0);
			} else {
				ms_filter_link(vol, 0, mx, k);
				ms_filter_link(mx, k, out, 0);
			}
			if (g_el) {
				MSFilter *volrecv = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
				float thres = 0.002f, force = 20.f;
				call_int(volrecv, MS_FILTER_SET_SAMPLE_RATE, 48000);
				ms_filter_call_method(vol, MS_VOLUME_SET_PEER, volrecv);
				ms_filter_call_method(vol, MS_VOLUME_SET_EA_THRESHOLD, &thres);
				ms_filter_call_method(vol, MS_VOLUME_SET_EA_FORCE, &force);
				ms_filter_link(far, 0, volrecv, 0);
				ms_filter_link(volrecv, 0, ec, 0);
			} else ms_filter_link(far, 0, ec, 0);
			ms_filter_link(ec, 0, spk, 0);
			if (c == 0 && k == 0) j->probe_out = out;
		}
	}
}

/* PLUGIN_BENCH_STACKS=<ms>: a watchdog signals a ticker thread whose step has been running for that long; the handler prints
 * where the thread is (backtrace: exported symbols of the HIP / HSA runtime included) -- what a rare long step is waiting in */
static volatile uint64_t g_step_start[256]; /* per ticker: start of the running step (ns), 0 = not in a step */
static pthread_t g_threads[256];
static double g_stack_ms;
static volatile int g_sampling;
static volatile int g_stack_dumps, g_done;
static uint64_t mono_ns(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
static void stack_handler(int sig) {
	(void)sig;
	void *buf[48];
	const int n = backtrace(buf, 48);
	static const char head[] = "---- plugin_bench: a step is taking long; this thread is in:\n";
	if (write(2, head, sizeof(head) - 1) < 0) return;
	backtrace_symbols_fd(buf, n, 2);
}
static void *watchdog(void *arg) {
	(void)arg;
	uint64_t dumped[256] = {0};
	while (!g_done && g_stack_dumps < 8) {
		const uint64_t now = mono_ns();
		for (int i = 0; i < g_tickers; ++i) {
			const uint64_t st = g_step_start[i];
			if (st && st != dumped[i] && (double)(now - st) * 1e-6 > g_stack_ms) {
				dumped[i] = st;
				++g_stack_dumps;
				pthread_kill(g_threads[i], SIGUSR1);
			}
		}
		usleep(300);
	}
	return NULL;
}

/* PLUGIN_BENCH_SAMPLE=<Hz>: a profiler of the ticker threads' steps where the box has none (no perf, no gdb): every ticker thread inside a
 * step is signalled <Hz> times a second, the handler keeps the innermost frames' return addresses; at exit they go to stderr as "module+offset" chains with
 * their counts (scripts/walk_profile.py names them with llvm-symbolizer).  Timed region only. */
#define SAMPLE_DEPTH 6
static void *(*g_samples)[SAMPLE_DEPTH];
static volatile int g_nsamples;
static pthread_t g_sampler;
static int g_sample_cap;
static void sample_handler(int sig) {
	(void)sig;
	if (!g_sampling) return;
	void *buf[SAMPLE_DEPTH + 2];
	const int n = backtrace(buf, SAMPLE_DEPTH + 2);
	const int at = __sync_fetch_and_add(&g_nsamples, 1);
	if (at >= g_sample_cap) return;
	for (int i = 0; i < SAMPLE_DEPTH; ++i) g_samples[at][i] = i + 2 < n ? buf[i + 2] : NULL; /* (past the handler and the signal trampoline) */
}
/* ITIMER_PROF ticks with the kernel's jiffies (250 a second for the whole process): a thread of its own signals every ticker thread that
 * is inside a step, `hz` times a second each */
static void *sampler(void *arg) {
	const double hz = atof(getenv("PLUGIN_BENCH_SAMPLE"));
	const useconds_t gap = (useconds_t)(1e6 / (hz > 0 ? hz : 1000));
	(void)arg;
	while (g_sampling) {
		for (int i = 0; i < g_tickers && i < 256; ++i)
			if (g_step_start[i]) pthread_kill(g_threads[i], SIGPROF);
		usleep(gap);
	}
	return NULL;
}
static void sample_start(void) {
	g_sample_cap = 400000;
	g_samples = calloc((size_t)g_sample_cap, sizeof(*g_samples));
	void *warm[4];
	backtrace(warm, 4);
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_handler = sample_handler;
	sa.sa_flags = SA_RESTART;
	sigaction(SIGPROF, &sa, NULL);
	g_sampling = 1;
	pthread_create(&g_sampler, NULL, sampler, NULL);
}
static void sample_report(void) {
	pthread_join(g_sampler, NULL);
	const int n = g_nsamples < g_sample_cap ? g_nsamples : g_sample_cap;
	fprintf(stderr, "== plugin_bench samples: %d\n", n);
	for (int i = 0; i < n; ++i) {
		fprintf(stderr, "S");
		for (int k = 0; k < SAMPLE_DEPTH && g_samples[i][k]; ++k) {
			Dl_info info;
			if (dladdr(g_samples[i][k], &info) && info.dli_fname)
				fprintf(stderr, " %s+0x%lx", info.dli_fname, (unsigned long)((char *)g_samples[i][k] - (char *)info.dli_fbase));
			else fprintf(stderr, " ?+%p", g_samples[i][k]);
		}
		fprintf(stderr, "\n");
	}
}

/* Paced: ticker k fires at origin + k x (10 ms / T) + n x 10 ms.  A server's MSTickers are created one by one with their conferences
 * (audioconference.c:70-73) and each paces itself from its own start (msticker.c:419-443,496-515): their phases are spread over the
 * interval, they do not all fire in the same instant.  PLUGIN_BENCH_ALIGNED=1: all at once (the worst case; every tick before round 5) */
static int g_aligned;
/* PLUGIN_BENCH_CHURN=<n>: n times a second every ticker has ONE conference (or, without mixers, one leg) re-plumbed the way
 * ms_audio_conference_add_member / remove_member do around every join and leave (src/voip/audioconference.c:322-374): the whole conference
 * graph detached and attached again -- every filter's postprocess and preprocess, the fused batch left and joined -- while the ticker
 * carries its full load -- ON AN APPLICATION THREAD, as the reference's callers do: ms_ticker_detach takes the ticker's lock to take the
 * graph's sources out (it waits for the tick in progress), runs the postprocess calls with the lock released; ms_ticker_attach runs the
 * preprocess calls (this plugin's fusing) without the lock, while the ticker walks its other graphs, and takes it to splice the sources in
 * (msticker.c:153-221,:462-493).  One such thread serves all the tickers in turn.  What is recorded: how long a re-plumbing took (waiting
 * for the ticker included) -- and, in the ticks' own figures, what it cost the tickers.  PLUGIN_BENCH_CHURN_ON_TICKER=1: on the ticker's
 * own thread in front of its step instead (counted in that tick: round 6's first measurement). */
static int g_churn, g_churn_on_ticker;
static volatile int g_churn_run;
static double *g_churn_ops;
static volatile int g_churn_nops;
static int g_churn_cap;
static void *churner(void *arg);
static uint64_t phase_ns(int index) { return g_aligned ? 0 : (uint64_t)index * (10000000ull / (uint64_t)g_tickers); }

static void *run(void *arg) {
	TickerJob *j = (TickerJob *)arg;
	/* attach on the ticker's own thread: the hub's device context and its banks belong to the thread that ticks them */
	const double a0 = now_ms();
	if (g_sampling) g_step_start[j->index] = mono_ns(); /* (PLUGIN_BENCH_SAMPLE_ATTACH: the attach is what is sampled) */
	if (g_nomixer)
		for (int k = 0; k < j->nconf * g_members; ++k) ms_ticker_attach(j->ticker, j->heads[k]);
	else
		for (int c = 0; c < j->nconf; ++c) ms_ticker_attach(j->ticker, j->mixers[c]);
	j->attach_ms = now_ms() - a0;
	g_step_start[j->index] = 0;
	/* the steps from the attach on are timed too (a start-up stall -- fusing, banks opening, slabs, a cold device -- must be visible,
	 * not folded into capacity): paced like the rest when PLUGIN_BENCH_PACED (a schedule of its own, origin g_w0) */
	if (g_paced) pthread_barrier_wait(&g_bar); /* (g_w0 is set) */
	for (int t = 0; t < g_warmup; ++t) {
		if (!g_paced) {
			pthread_barrier_wait(&g_bar);
		} else {
			const uint64_t sched = g_w0 + phase_ns(j->index) + (uint64_t)t * 10000000ull;
			if (mono_ns() < sched) {
				struct timespec ts = {(time_t)(sched / 1000000000ull), (long)(sched % 1000000000ull)};
				clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL);
			}
		}
		const double w0 = now_ms();
		if ((g_stack_ms > 0 && getenv("PLUGIN_BENCH_STACKS_WARMUP")) || g_sampling) g_step_start[j->index] = mono_ns(); /* (the first ticks after the attach too) */
		ms_ticker_step(j->ticker);
		g_step_start[j->index] = 0;
		j->warm_ms[t] = now_ms() - w0;
	}
	if (g_paced) pthread_barrier_wait(&g_bar); /* (the warm-up is through on every ticker) */
	if (g_paced) pthread_barrier_wait(&g_bar); /* (the schedule's origin is set; from here on the wall clock fires the ticks) */
	for (int t = 0; t < g_ticks; ++t) {
		if (!g_paced) {
			pthread_barrier_wait(&g_bar); /* back to back: all tickers fire together as soon as the slowest is done */
		} else { /* msticker.c:419-443,496-515: sleep until the tick's time; no sleep while behind (the late ticks are caught up) */
			const uint64_t sched = g_t0 + phase_ns(j->index) + (uint64_t)t * 10000000ull;
			uint64_t now = mono_ns();
			if (now < sched) {
				struct timespec ts = {(time_t)(sched / 1000000000ull), (long)(sched % 1000000000ull)};
				clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL);
				now = mono_ns();
			}
			j->late_ms[t] = now > sched ? (double)(now - sched) * 1e-6 : 0.0;
		}
		struct rusage ru0;
		getrusage(RUSAGE_THREAD, &ru0);
		const double c0 = thread_cpu_ms();
		const double t0 = now_ms();
		if ((g_stack_ms > 0 && t > 50) || g_sampling) g_step_start[j->index] = mono_ns();
		if (g_churn > 0 && g_churn_on_ticker && t % (100 / g_churn > 0 ? 100 / g_churn : 1) == (j->index % (100 / g_churn > 0 ? 100 / g_churn : 1))) {
			const int units = g_nomixer ? j->nconf * g_members : j->nconf;
			MSFilter *root = g_nomixer ? j->heads[j->churn_next % units] : j->mixers[j->churn_next % units];
			j->churn_next++;
			ms_ticker_detach(j->ticker, root);
			ms_ticker_attach(j->ticker, root);
			j->churn_ms[t] = now_ms() - t0;
		}
		ms_ticker_step(j->ticker);
		g_step_start[j->index] = 0;
		j->step_ms[t] = now_ms() - t0;
		struct rusage ru1;
		getrusage(RUSAGE_THREAD, &ru1);
		j->cpu_ms[t] = thread_cpu_ms() - c0;
		j->nvcsw[t] = (int)(ru1.ru_nvcsw - ru0.ru_nvcsw);
		j->nivcsw[t] = (int)(ru1.ru_nivcsw - ru0.ru_nivcsw);
		j->minflt[t] = (int)(ru1.ru_minflt - ru0.ru_minflt);
		uint64_t tasks_ns = 0;
		ms2shim_ticker_last_step(j->ticker, &tasks_ns, NULL);
		j->task_ms[t] = (double)tasks_ns * 1e-6;
		if (g_profile) {
			int ids[16], mid = 0;
			uint64_t ns[16], mns = 0;
			const int n = ms2shim_ticker_profile(j->ticker, ids, ns, 16, &mid, &mns);
			for (int k = 0; k < n; ++k)
				for (int q = 0; q < 16; ++q)
					if (j->tot_ids[q] == ids[k] || j->tot_ids[q] == 0) {
						j->tot_ids[q] = ids[k];
						j->tot_ns[q] += ns[k];
						break;
					}
			if (j->step_ms[t] > j->slowest_ms) { /* ... and where this thread's slowest step went */
				j->slowest_ms = j->step_ms[t];
				j->slowest_tick = t;
				j->prof_n = n, j->max_id = mid, j->max_ns = mns;
				memcpy(j->prof_ids, ids, sizeof(ids)), memcpy(j->prof_ns, ns, sizeof(ns));
			}
		}
	}
	pthread_barrier_wait(&g_bar);
	return NULL;
}

static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }
static double pct(const double *sorted, int n, double p) { return sorted[(int)((double)(n - 1) * p)]; }

static TickerJob *g_jobs;
static void *churner(void *arg) { /* the application's thread */
	(void)arg;
	const double gap_ms = 1000.0 / ((double)g_churn * g_tickers);
	double next = now_ms() + gap_ms;
	for (int n = 0; g_churn_run; ++n) {
		const double now = now_ms();
		if (now < next) {
			usleep((useconds_t)((next - now) * 1000.0));
			continue;
		}
		next += gap_ms;
		if (now - next > 50.0) next = now + gap_ms; /* (far behind: do not burst) */
		TickerJob *j = &g_jobs[n % g_tickers];
		const int units = g_nomixer ? j->nconf * g_members : j->nconf;
		MSFilter *root = g_nomixer ? j->heads[j->churn_next % units] : j->mixers[j->churn_next % units];
		j->churn_next++;
		const double t0 = now_ms();
		ms_ticker_detach(j->ticker, root);
		ms_ticker_attach(j->ticker, root);
		if (g_churn_nops < g_churn_cap) g_churn_ops[g_churn_nops++] = now_ms() - t0;
	}
	return NULL;
}

int main(int argc, char **argv) {
	if (argc < 6) {
		fprintf(stderr, "usage: plugin_bench <plugin.so> <legs> <tickers> <ticks> <warmup> [members]\n");
		return 2;
	}
	g_profile = getenv("MS2SHIM_PROFILE") != NULL;
	g_paced = getenv("PLUGIN_BENCH_PACED") != NULL;
	g_aligned = getenv("PLUGIN_BENCH_ALIGNED") != NULL;
	g_churn = getenv("PLUGIN_BENCH_CHURN") ? atoi(getenv("PLUGIN_BENCH_CHURN")) : 0;
	if (g_churn > 100) g_churn = 100;
	g_churn_on_ticker = getenv("PLUGIN_BENCH_CHURN_ON_TICKER") != NULL;
	if (getenv("PLUGIN_BENCH_SHAPE")) {
		const char *sh = getenv("PLUGIN_BENCH_SHAPE");
		g_nors = strstr(sh, "nors") != NULL, g_noagc = strstr(sh, "noagc") != NULL, g_nomixer = strstr(sh, "nomixer") != NULL, g_eprs = strstr(sh, "eprs") != NULL, g_server = strstr(sh, "server") != NULL, g_dec = strstr(sh, "dec") != NULL;
		g_astream = strstr(sh, "astream") != NULL;
		g_default = g_astream && strstr(sh, "default") != NULL;
		g_wb = strstr(sh, " wb") != NULL;
		if (g_astream) g_nomixer = 1;
		g_eq = strstr(sh, "eq") != NULL, g_el = strstr(sh, " el") != NULL || strncmp(sh, "el", 2) == 0;
	}
	g_checksum = getenv("PLUGIN_BENCH_CHECKSUM") != NULL; /* (costs the walk ~2 us per leg-tick: for parity runs, not for timing) */
	/* glibc's per-thread arenas grow 128 KB at a time, each step an mprotect under the process's mmap lock: sixteen tickers allocating their first
	 * ticks' blocks at once spent 65 % of those ticks there (scripts/r06_first_ticks_profile.sh).  The application's own choice, as a media
	 * server would make it (or link another allocator): grow in 32 MB steps, do not give the top back */
	if (!getenv("PLUGIN_BENCH_DEFAULT_MALLOC")) { /* (set: glibc's defaults, for an A/B) */
		mallopt(M_TOP_PAD, 32 << 20);
		mallopt(M_TRIM_THRESHOLD, 512 << 20);
	}
	const char *plugin = argv[1];
	int legs = atoi(argv[2]);
	g_tickers = atoi(argv[3]);
	g_ticks = atoi(argv[4]);
	g_warmup = atoi(argv[5]);
	if (argc > 6) g_members = atoi(argv[6]);
	if (legs <= 0 || g_tickers <= 0 || g_ticks <= 0 || g_members <= 0 || g_members > 50) return 2;
	const int nconf_total = legs / g_members, nconf = nconf_total / g_tickers;
	if (nconf <= 0) {
		fprintf(stderr, "plugin_bench: %d legs do not fill a conference of %d on each of %d tickers\n", legs, g_members, g_tickers);
		return 2;
	}
	legs = nconf * g_tickers * g_members;
	if (!getenv("MSMI355X_SLOTS")) { /* first bank of a hub sized for the hub's legs (a server that knows its size would do the same) */
		char v[32];
		int slots = nconf * g_members / 4;
		snprintf(v, sizeof(v), "%d", slots < 16 ? 16 : slots);
		setenv("MSMI355X_SLOTS", v, 1);
	}
	unsigned seed = 12345u;
	for (int r = 0; r < RING; ++r) {
		for (int i = 0; i < 160; ++i) g_mic[r][i] = (int16_t)((int)((seed = seed * 1664525u + 1013904223u) >> 19) - 4096);
		for (int i = 0; i < 480; ++i) g_far[r][i] = (int16_t)((int)((seed = seed * 1664525u + 1013904223u) >> 18) - 8192);
		for (int i = 0; i < 80; ++i) g_pcm8[r][i] = (int16_t)((int)((seed = seed * 1664525u + 1013904223u) >> 19) - 4096);
		for (int i = 0; i < 80; ++i) g_codes8[r][i] = (uint8_t)((seed = seed * 1664525u + 1013904223u) >> 24);
	}
	g_fac = ms_factory_new();
	ms2shim_register_test_filters(g_fac);
	if (ms_factory_load_plugin(g_fac, plugin) != 0) {
		fprintf(stderr, "plugin_bench: could not load %s\n", plugin);
		return 1;
	}
	if (ms_factory_create_filter(g_fac, MS_RESAMPLE_ID) == NULL) {
		fprintf(stderr, "plugin_bench: the plugin registered no filters (no HIP device?)\n");
		return 1;
	}
	void *ph = dlopen(plugin, RTLD_NOW | RTLD_NOLOAD);
	void (*fused_stats)(int *, int *, unsigned long long *, unsigned long long *) =
	    ph ? (void (*)(int *, int *, unsigned long long *, unsigned long long *))dlsym(ph, "ms_mi355x_fused_stats") : NULL;
	unsigned long long (*late_events)(void) = ph ? (unsigned long long (*)(void))dlsym(ph, "ms_mi355x_late_events") : NULL;

	const double t_build0 = now_ms();
	TickerJob *jobs = (TickerJob *)calloc((size_t)g_tickers, sizeof(TickerJob));
	for (int i = 0; i < g_tickers; ++i) {
		jobs[i].index = i;
		jobs[i].nconf = nconf;
		jobs[i].step_ms = (double *)calloc((size_t)g_ticks, sizeof(double));
		jobs[i].task_ms = (double *)calloc((size_t)g_ticks, sizeof(double));
		jobs[i].cpu_ms = (double *)calloc((size_t)g_ticks, sizeof(double));
		jobs[i].late_ms = (double *)calloc((size_t)g_ticks, sizeof(double));
		jobs[i].warm_ms = (double *)calloc((size_t)g_warmup + 1, sizeof(double));
		jobs[i].nvcsw = (int *)calloc((size_t)g_ticks, sizeof(int));
		jobs[i].nivcsw = (int *)calloc((size_t)g_ticks, sizeof(int));
		jobs[i].minflt = (int *)calloc((size_t)g_ticks, sizeof(int));
		jobs[i].churn_ms = (double *)calloc((size_t)g_ticks, sizeof(double));
		build(&jobs[i]);
	}
	const double build_ms = now_ms() - t_build0;
	pthread_barrier_init(&g_bar, NULL, (unsigned)g_tickers + 1);
	pthread_t *th = (pthread_t *)calloc((size_t)g_tickers, sizeof(pthread_t));
	const int sample_attach = getenv("PLUGIN_BENCH_SAMPLE") && getenv("PLUGIN_BENCH_SAMPLE_ATTACH") && g_tickers <= 256;
	if (sample_attach) g_sampling = 1; /* (the threads mark themselves from their first line on; the sampler starts once they exist) */
	for (int i = 0; i < g_tickers; ++i) pthread_create(&th[i], NULL, run, &jobs[i]);
	if (sample_attach) {
		for (int i = 0; i < g_tickers; ++i) g_threads[i] = th[i];
		sample_start();
	}
	pthread_t wd;
	if (getenv("PLUGIN_BENCH_STACKS") && g_tickers <= 256) {
		g_stack_ms = atof(getenv("PLUGIN_BENCH_STACKS"));
		void *warm[4];
		backtrace(warm, 4); /* (loads libgcc now, not inside the handler) */
		signal(SIGUSR1, stack_handler);
		for (int i = 0; i < g_tickers; ++i) g_threads[i] = th[i];
		pthread_create(&wd, NULL, watchdog, NULL);
	}
	int fc0 = 0, fl0 = 0, fc1 = 0, fl1 = 0;
	unsigned long long la0 = 0, fr0 = 0, la1 = 0, fr1 = 0;
	const double t_warm0 = now_ms();
	const int sample_warmup = getenv("PLUGIN_BENCH_SAMPLE") && getenv("PLUGIN_BENCH_SAMPLE_WARMUP") && !sample_attach && g_tickers <= 256; /* the ticks from the attach on instead of the timed ones */
	if (sample_warmup) {
		for (int i = 0; i < g_tickers; ++i) g_threads[i] = th[i];
		sample_start();
	}
	if (g_paced) {
		g_w0 = mono_ns() + 20000000ull;
		pthread_barrier_wait(&g_bar);
		pthread_barrier_wait(&g_bar); /* every ticker has done its warm-up steps */
	} else {
		for (int t = 0; t < g_warmup; ++t) pthread_barrier_wait(&g_bar);
	}
	if (sample_warmup) g_sampling = 0, sample_report();
	if (sample_attach && g_sampling) g_sampling = 0, sample_report(); /* (the attach and the warm-up ticks behind it) */
	/* the warm-up's last step is running; the first timed barrier releases when it is done */
	double t_first = 0;
	pthread_t churn_th;
	if (g_churn > 0 && !g_churn_on_ticker) {
		g_jobs = jobs;
		g_churn_cap = 1 << 16;
		g_churn_ops = (double *)calloc((size_t)g_churn_cap, sizeof(double));
		g_churn_run = 1;
		pthread_create(&churn_th, NULL, churner, NULL);
	}
	if (getenv("PLUGIN_BENCH_SAMPLE") && !sample_warmup && !sample_attach && g_tickers <= 256) {
		for (int i = 0; i < g_tickers; ++i) g_threads[i] = th[i];
		sample_start();
	}
	if (g_paced) {
		g_t0 = mono_ns() + 20000000ull; /* the first tick fires 20 ms from now */
		pthread_barrier_wait(&g_bar);
		t_first = now_ms() + 20.0;
		if (fused_stats) fused_stats(&fc0, &fl0, &la0, &fr0);
	} else {
		for (int t = 0; t < g_ticks; ++t) {
			pthread_barrier_wait(&g_bar);
			if (t == 0) {
				t_first = now_ms();
				if (fused_stats) fused_stats(&fc0, &fl0, &la0, &fr0); /* (racing with the first timed step by a launch or two: negligible over the run) */
			}
		}
	}
	pthread_barrier_wait(&g_bar);
	const double wall_ms = now_ms() - t_first;
	if (g_churn_run) {
		g_churn_run = 0;
		pthread_join(churn_th, NULL);
	}
	if (g_sampling) g_sampling = 0, sample_report();
	if (fused_stats) fused_stats(&fc1, &fl1, &la1, &fr1);
	for (int i = 0; i < g_tickers; ++i) pthread_join(th[i], NULL);
	g_done = 1;

	/* a tick costs what the slowest ticker needs */
	double *tick = (double *)calloc((size_t)g_ticks, sizeof(double)), *task = (double *)calloc((size_t)g_ticks, sizeof(double));
	double sum_step = 0, sum_task = 0;
	for (int t = 0; t < g_ticks; ++t) {
		for (int i = 0; i < g_tickers; ++i) {
			if (jobs[i].step_ms[t] > tick[t]) tick[t] = jobs[i].step_ms[t], task[t] = jobs[i].task_ms[t];
			sum_step += jobs[i].step_ms[t];
			sum_task += jobs[i].task_ms[t];
		}
	}
	double *sorted = (double *)malloc(sizeof(double) * (size_t)g_ticks);
	memcpy(sorted, tick, sizeof(double) * (size_t)g_ticks);
	qsort(sorted, (size_t)g_ticks, sizeof(double), cmp_d);
	int late = 0, worst_t = 0, worst_i = 0;
	for (int t = 0; t < g_ticks; ++t) {
		late += tick[t] >= 10.0;
		if (tick[t] > tick[worst_t]) worst_t = t;
	}
	for (int i = 0; i < g_tickers; ++i)
		if (jobs[i].step_ms[worst_t] > jobs[worst_i].step_ms[worst_t]) worst_i = i;
	/* what an MSTicker makes of these ticks (src/base/msticker.c:419-443,496-515): it does not sleep while it is behind, so a long
	 * tick is caught up by the short ones after it; it reports a late event once it is more than 5 intervals behind */
	double backlog = 0, max_backlog = 0;
	int ref_late_events = 0;
	if (g_paced) { /* measured, not modelled: how far behind its schedule a ticker's step started */
		for (int i = 0; i < g_tickers; ++i)
			for (int t = 0; t < g_ticks; ++t) {
				if (jobs[i].late_ms[t] > max_backlog) max_backlog = jobs[i].late_ms[t];
				ref_late_events += jobs[i].late_ms[t] > 50.0;
			}
	} else {
		for (int t = 0; t < g_ticks; ++t) {
			backlog += tick[t] - 10.0;
			if (backlog < 0) backlog = 0;
			if (backlog > max_backlog) max_backlog = backlog;
			ref_late_events += backlog > 50.0;
		}
	}
	/* the ticks that took longest: was the slowest thread running (cpu_ms ~ ms), blocked (voluntary switches) or pushed off its core? */
	char slow[1024];
	int so = 0;
	{
		double *copy = (double *)malloc(sizeof(double) * (size_t)g_ticks);
		memcpy(copy, tick, sizeof(double) * (size_t)g_ticks);
		for (int n = 0; n < 5 && n < g_ticks; ++n) {
			int bt = 0, bi = 0;
			for (int t = 0; t < g_ticks; ++t)
				if (copy[t] > copy[bt]) bt = t;
			for (int i = 0; i < g_tickers; ++i)
				if (jobs[i].step_ms[bt] > jobs[bi].step_ms[bt]) bi = i;
			int others = 0; /* how many OTHER tickers were also over 8 ms in that tick: one thread's mishap or everybody's */
			for (int i = 0; i < g_tickers; ++i) others += (i != bi && jobs[i].step_ms[bt] > 8.0);
			so += snprintf(slow + so, sizeof(slow) - (size_t)so, "%s{\"index\": %d, \"ticker\": %d, \"ms\": %.2f, \"cpu_ms\": %.2f, \"flush_ms\": %.2f, \"nvcsw\": %d, \"nivcsw\": %d, \"minflt\": %d, \"others_over_8ms\": %d}",
			               n ? ", " : "", bt, bi, jobs[bi].step_ms[bt], jobs[bi].cpu_ms[bt], jobs[bi].task_ms[bt], jobs[bi].nvcsw[bt], jobs[bi].nivcsw[bt], jobs[bi].minflt[bt], others);
			copy[bt] = 0;
		}
		free(copy);
	}
	double sum_cpu = 0;
	long sum_flt = 0, sum_nv = 0, sum_niv = 0;
	for (int i = 0; i < g_tickers; ++i)
		for (int t = 0; t < g_ticks; ++t) sum_cpu += jobs[i].cpu_ms[t], sum_flt += jobs[i].minflt[t], sum_nv += jobs[i].nvcsw[t], sum_niv += jobs[i].nivcsw[t];
	if (g_profile) { /* stderr: the slowest step of the slowest thread, by filter id */
		int bi = 0;
		for (int i = 0; i < g_tickers; ++i)
			if (jobs[i].slowest_ms > jobs[bi].slowest_ms) bi = i;
		fprintf(stderr, "plugin_bench profile: ticker %d tick %d took %.2f ms; longest single process(): id %d %.3f ms; by id:", bi, jobs[bi].slowest_tick,
		        jobs[bi].slowest_ms, jobs[bi].max_id, (double)jobs[bi].max_ns * 1e-6);
		for (int k = 0; k < jobs[bi].prof_n; ++k) fprintf(stderr, " %d=%.3fms", jobs[bi].prof_ids[k], (double)jobs[bi].prof_ns[k] * 1e-6);
		fprintf(stderr, "\n");
	}
	/* from the attach on: the warm-up's steps (the slowest ticker's per step), their five longest */
	char fa[640];
	{
		double *w = (double *)calloc((size_t)g_warmup + 1, sizeof(double)), *ws = (double *)calloc((size_t)g_warmup + 1, sizeof(double));
		int over = 0, fo = 0;
		for (int t = 0; t < g_warmup; ++t) {
			for (int i = 0; i < g_tickers; ++i)
				if (jobs[i].warm_ms[t] > w[t]) w[t] = jobs[i].warm_ms[t];
			over += w[t] >= 10.0;
			ws[t] = w[t];
		}
		qsort(ws, (size_t)g_warmup, sizeof(double), cmp_d);
		fo += snprintf(fa, sizeof(fa), "{\"ticks\": %d, \"paced\": %s, \"p50_ms\": %.3f, \"max_ms\": %.3f, \"ticks_over_10ms\": %d, \"first_ms\": [", g_warmup, g_paced ? "true" : "false",
		               g_warmup ? pct(ws, g_warmup, 0.5) : 0.0, g_warmup ? ws[g_warmup - 1] : 0.0, over);
		for (int t = 0; t < g_warmup && t < 8; ++t) fo += snprintf(fa + fo, sizeof(fa) - (size_t)fo, "%s%.2f", t ? ", " : "", w[t]);
		fo += snprintf(fa + fo, sizeof(fa) - (size_t)fo, "], \"slowest\": [");
		for (int n = 0; n < 5 && n < g_warmup; ++n) {
			int bt = 0;
			for (int t = 0; t < g_warmup; ++t)
				if (w[t] > w[bt]) bt = t;
			fo += snprintf(fa + fo, sizeof(fa) - (size_t)fo, "%s[%d, %.2f]", n ? ", " : "", bt, w[bt]);
			w[bt] = 0;
		}
		snprintf(fa + fo, sizeof(fa) - (size_t)fo, "]}");
		free(w), free(ws);
	}
	/* MS2SHIM_PROFILE=1: the graph walk by filter id, us per leg and tick over all tickers and timed steps (9001 / 9002: the test
	 * runtime's sources / sinks -- the HARNESS's share; the rest are the plugin's facades; the timer itself costs ~0.05 us per call) */
	char byid[512] = "";
	if (g_profile) {
		int ids[16] = {0}, bo = 0;
		double us[16] = {0};
		for (int i = 0; i < g_tickers; ++i)
			for (int q = 0; q < 16 && jobs[i].tot_ids[q]; ++q)
				for (int k = 0; k < 16; ++k)
					if (ids[k] == jobs[i].tot_ids[q] || ids[k] == 0) {
						ids[k] = jobs[i].tot_ids[q];
						us[k] += (double)jobs[i].tot_ns[q] * 1e-3;
						break;
					}
		for (int k = 0; k < 16 && ids[k]; ++k)
			bo += snprintf(byid + bo, sizeof(byid) - (size_t)bo, "%s\"%d\": %.4f", k ? ", " : "", ids[k], us[k] / ((double)g_ticks * legs));
	}
	/* PLUGIN_BENCH_CHECKSUM=1: the whole run's output as two numbers -- every leg's mix and speaker audio, byte for byte and in
	 * order, folded per sink (FNV-1a) and summed over the legs: equal between two runs iff (to 2^-64) every leg heard the same */
	unsigned long long mix_sum = 0, spk_sum = 0, out_bytes = 0;
	if (g_checksum)
		for (int i = 0; i < g_tickers; ++i)
			for (int k = 0; k < jobs[i].nconf * g_members; ++k) {
				mix_sum += ms2shim_sink_sum(jobs[i].outs[k]) * (unsigned long long)(2 * (i * jobs[i].nconf * g_members + k) + 1);
				if (jobs[i].spks[k]) spk_sum += ms2shim_sink_sum(jobs[i].spks[k]) * (unsigned long long)(2 * (i * jobs[i].nconf * g_members + k) + 1);
				out_bytes += ms2shim_sink_size(jobs[i].outs[k]);
			}
	double attach_max = 0;
	for (int i = 0; i < g_tickers; ++i)
		if (jobs[i].attach_ms > attach_max) attach_max = jobs[i].attach_ms;
	char churn[384] = "null";
	if (g_churn > 0) { /* the re-plumbings: how many, what one took (median, longest) */
		double *ops = (double *)calloc((size_t)g_ticks * (size_t)g_tickers + (size_t)g_churn_nops + 1, sizeof(double));
		int nops = 0;
		for (int i = 0; i < g_tickers && g_churn_on_ticker; ++i)
			for (int t = 0; t < g_ticks; ++t)
				if (jobs[i].churn_ms[t] > 0) ops[nops++] = jobs[i].churn_ms[t];
		for (int k = 0; k < g_churn_nops && !g_churn_on_ticker; ++k) ops[nops++] = g_churn_ops[k];
		qsort(ops, (size_t)nops, sizeof(double), cmp_d);
		snprintf(churn, sizeof(churn), "{\"per_second_and_ticker\": %d, \"replumbings\": %d, \"p50_ms\": %.3f, \"p99_ms\": %.3f, \"max_ms\": %.3f, \"counted_in_the_tick\": %s, \"thread\": \"%s\"}", g_churn, nops,
		         nops ? pct(ops, nops, 0.5) : 0.0, nops ? pct(ops, nops, 0.99) : 0.0, nops ? ops[nops - 1] : 0.0, g_churn_on_ticker ? "true" : "false", g_churn_on_ticker ? "the ticker's" : "the application's");
		free(ops);
	}
	const double mean_step = sum_step / ((double)g_ticks * g_tickers), mean_task = sum_task / ((double)g_ticks * g_tickers);
	printf("{\"paced\": %s, \"phases\": \"%s\", \"legs\": %d, \"members\": %d, \"conferences\": %d, \"tickers\": %d, \"ticks\": %d, \"warmup\": %d, "
	       "\"p50_ms\": %.4f, \"p99_ms\": %.4f, \"max_ms\": %.4f, \"late\": %d, \"wall_ms_per_tick\": %.4f, "
	       "\"ticker_mean_ms\": %.4f, \"ticker_flush_ms\": %.4f, \"ticker_graph_walk_ms\": %.4f, \"us_per_leg_tick\": %.4f, "
	       "\"fused_conferences\": %d, \"fused_legs\": %d, \"launches_per_tick\": %.2f, \"launches_per_tick_and_ticker\": %.2f, "
	       "\"flush_rounds_per_tick_and_ticker\": %.2f, \"late_events\": %llu, \"probe_sink_blocks\": %d, \"probe_sink_bytes\": %zu, "
	       "\"build_ms\": %.1f, \"warmup_ms\": %.1f, \"worst_tick\": {\"index\": %d, \"ticker\": %d, \"ms\": %.3f, \"flush_ms\": %.3f}, "
	       "\"p99_9_ms\": %.4f, \"mean_ms\": %.4f, \"max_backlog_ms\": %.3f, \"msticker_late_events\": %d, "
	       "\"ticker_cpu_ms\": %.4f, \"minflt_per_tick_and_ticker\": %.2f, \"nvcsw_per_tick_and_ticker\": %.2f, \"nivcsw_per_tick_and_ticker\": %.3f, \"slow_ticks\": [%s], "
	       "\"mix_checksum\": \"%016llx\", \"speaker_checksum\": \"%016llx\", \"mix_bytes\": %llu, \"walk_us_per_leg_tick_by_filter_id\": {%s}, \"from_attach\": %s, \"churn\": %s, \"attach_ms_slowest_ticker\": %.1f}\n",
	       g_paced ? "true" : "false", !g_paced ? "barrier" : (g_aligned ? "aligned" : "spread"), legs, g_members, nconf * g_tickers, g_tickers, g_ticks, g_warmup, pct(sorted, g_ticks, 0.5), pct(sorted, g_ticks, 0.99), sorted[g_ticks - 1], late,
	       wall_ms / g_ticks, mean_step, mean_task, mean_step - mean_task, mean_step * 1e3 * g_tickers / legs, fc1, fl1,
	       (double)(la1 - la0) / g_ticks, (double)(la1 - la0) / g_ticks / g_tickers, (double)(fr1 - fr0) / g_ticks / g_tickers,
	       late_events ? late_events() : 0ull, ms2shim_sink_blocks(jobs[0].probe_out), ms2shim_sink_size(jobs[0].probe_out), build_ms, t_first - t_warm0,
	       worst_t, worst_i, jobs[worst_i].step_ms[worst_t], jobs[worst_i].task_ms[worst_t], pct(sorted, g_ticks, 0.999), wall_ms / g_ticks,
	       max_backlog, ref_late_events, sum_cpu / ((double)g_ticks * g_tickers), (double)sum_flt / ((double)g_ticks * g_tickers),
	       (double)sum_nv / ((double)g_ticks * g_tickers), (double)sum_niv / ((double)g_ticks * g_tickers), slow, mix_sum, spk_sum, out_bytes, byid, fa, churn, attach_max);
	fflush(stdout);
	/* the graphs are left as they are: the process ends here (tearing 10^5 filters down is not what is measured) */
	if (getenv("PLUGIN_BENCH_CLEAN_EXIT")) exit(0); /* (under rocprofv3: its summary is written by an exit handler) */
	_exit(0);
}
