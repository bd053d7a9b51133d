/*
 * san_stress.c -- TEST INFRASTRUCTURE: the plugin's threaded host runtime (mediastreamer2_amd/host/filters.cpp: one hub
 * per ticker with its own lock, a registry under a shared mutex, reference-counted banks that grow and die) driven hard
 * from several threads, to be run under ThreadSanitizer and AddressSanitizer + UBSan.  The kernel library behind the
 * plugin is the host-memory double (mi_double.cpp), so this runs on a box without a GPU; the test runtime (ms2shim.c) is
 * compiled into this executable and exports the ms_* symbols the plugin resolves (-rdynamic).
 *
 *   san_stress <plugin.so> [rounds]
 *
 * Threads:
 *   callers x3   each: a ticker of its own, a few source -> MSResample -> MSVolume -> sink chains (plus an MSSpeexEC and a
 *                conference mixer now and then), some ticks, detach, destroy everything -- the hub dies with its last bank;
 *   grower       one ticker, 150 filters attached one after the other (banks of 16, 64, 256 slots open), ticks, half of
 *                them detached and destroyed, ticks, the rest destroyed, all over again;
 *   conferences  one ticker, two conferences of  source -> MSResample -> MSSpeexEC -> MSVolume (AGC) -> MSAudioMixer  legs: the
 *                plugin fuses each into its device-resident batch (filters/leg_chain.inl); ticks, a volume method from this
 *                thread between ticks, one leg's canceller switched to bypass (the conference leaves the batch and goes on
 *                facade by facade), detach, re-attach (fuses again), ticks, everything destroyed;
 *   walker       ms_mi355x_runtime_stats / ms_mi355x_hub_devices over every hub, all the time (ms_mi355x_flush is for a
 *                stopped graph: it emits into the filters' queues and is called once, at the end).
 * Ends with the runtime where it started (no hub, no bank, no slot) and no late events; prints "ok".
 */
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/ms2_plugin_abi.h"

MSFilter *ms2shim_new_source(MSFactory *f);
MSFilter *ms2shim_new_sink(MSFactory *f);
MSFilter *ms2shim_new_pass(MSFactory *f);
void ms2shim_register_test_filters(MSFactory *f);
void ms2shim_source_push(MSFilter *src, const void *data, size_t nbytes);
void ms2shim_sink_set_discard(MSFilter *f, int on);

static MSFactory *g_fac;
static int g_rounds = 12;
static volatile int g_stop;
static int g_fail;
static void (*p_stats)(int *, int *, int *);
static void (*p_flush)(void);
static int (*p_hub_devices)(int *, int);
static unsigned long long (*p_late)(void);
static int (*p_in_batch)(MSFilter *);

#define CHECK(c)                                                        \
	do {                                                                \
		if (!(c)) {                                                     \
			fprintf(stderr, "san_stress: %s:%d: %s\n", __FILE__, __LINE__, #c); \
			__atomic_store_n(&g_fail, 1, __ATOMIC_SEQ_CST);             \
		}                                                               \
	} while (0)

static int set_int(MSFilter *f, unsigned int id, int v) { return ms_filter_call_method(f, id, &v); }

typedef struct {
	MSFilter *src, *rs, *vol, *snk;
} chain_t;

static void chain_make(chain_t *c, int in_rate) {
	float g = 0.5f;
	c->src = ms2shim_new_source(g_fac);
	c->rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
	c->vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
	c->snk = ms2shim_new_sink(g_fac);
	CHECK(c->src && c->rs && c->vol && c->snk);
	ms2shim_sink_set_discard(c->snk, 1);
	CHECK(set_int(c->rs, MS_FILTER_SET_SAMPLE_RATE, in_rate) == 0);
	CHECK(set_int(c->rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000) == 0);
	CHECK(set_int(c->vol, MS_FILTER_SET_SAMPLE_RATE, 48000) == 0);
	CHECK(ms_filter_call_method(c->vol, MS_VOLUME_SET_GAIN, &g) == 0);
	ms_filter_link(c->src, 0, c->rs, 0);
	ms_filter_link(c->rs, 0, c->vol, 0);
	ms_filter_link(c->vol, 0, c->snk, 0);
}
static void chain_destroy(chain_t *c) {
	ms_filter_unlink(c->src, 0, c->rs, 0);
	ms_filter_unlink(c->rs, 0, c->vol, 0);
	ms_filter_unlink(c->vol, 0, c->snk, 0);
	ms_filter_destroy(c->src);
	ms_filter_destroy(c->rs);
	ms_filter_destroy(c->vol);
	ms_filter_destroy(c->snk);
}

static void *caller(void *arg) {
	const int seed = (int)(intptr_t)arg;
	int16_t block[160];
	for (int i = 0; i < 160; ++i) block[i] = (int16_t)((i * 37 + seed * 101) % 2000 - 1000);
	for (int rep = 0; rep < g_rounds; ++rep) {
		MSTicker *tk = ms_ticker_new();
		int n = 3 + (seed + rep) % 4;
		chain_t ch[8];
		MSFilter *ec = NULL, *ecs[2] = {NULL, NULL}, *eck[2] = {NULL, NULL};
		for (int i = 0; i < n; ++i) {
			chain_make(&ch[i], 16000);
			CHECK(ms_ticker_attach(tk, ch[i].src) == 0);
		}
		/* an echo-limiter pair (audiostream.c:2240): chain 0's MSVolume (volsend) names chain 1's (volrecv) as its peer */
		const int peered = (seed + rep) % 3 == 0;
		if (peered) CHECK(ms_filter_call_method(ch[0].vol, MS_VOLUME_SET_PEER, ch[1].vol) == 0);
		if ((seed + rep) % 2 == 0) { /* an echo canceller: far end -> pin 0, microphone -> pin 1 */
			ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID);
			CHECK(ec != NULL);
			set_int(ec, MS_FILTER_SET_SAMPLE_RATE, 16000);
			for (int k = 0; k < 2; ++k) {
				ecs[k] = ms2shim_new_source(g_fac);
				eck[k] = ms2shim_new_sink(g_fac);
				ms2shim_sink_set_discard(eck[k], 1);
				ms_filter_link(ecs[k], 0, ec, k);
				ms_filter_link(ec, k, eck[k], 0);
			}
			CHECK(ms_ticker_attach(tk, ec) == 0);
		}
		MSFilter *mx = NULL, *mxs[3] = {NULL, NULL, NULL}, *mxk[3] = {NULL, NULL, NULL};
		if ((seed + rep) % 2 == 1) { /* a three-party conference: the mixer is a pump facade */
			mx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
			CHECK(mx != NULL);
			set_int(mx, MS_FILTER_SET_SAMPLE_RATE, 16000);
			set_int(mx, MS_FILTER_SET_NCHANNELS, 1);
			set_int(mx, MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
			for (int k = 0; k < 3; ++k) {
				mxs[k] = ms2shim_new_source(g_fac);
				mxk[k] = ms2shim_new_sink(g_fac);
				ms2shim_sink_set_discard(mxk[k], 1);
				ms_filter_link(mxs[k], 0, mx, k);
				ms_filter_link(mx, k, mxk[k], 0);
			}
			CHECK(ms_ticker_attach(tk, mx) == 0);
		}
		for (int t = 0; t < 6; ++t) {
			for (int i = 0; i < n; ++i) ms2shim_source_push(ch[i].src, block, sizeof block);
			if (ec)
				for (int k = 0; k < 2; ++k) ms2shim_source_push(ecs[k], block, sizeof block);
			if (mx)
				for (int k = 0; k < 3; ++k) ms2shim_source_push(mxs[k], block, sizeof block);
			ms_ticker_step(tk);
			if (t == 3 && n > 3) { /* one call ends mid-way */
				ms_ticker_detach(tk, ch[n - 1].src);
				chain_destroy(&ch[n - 1]);
				--n;
			}
		}
		for (int i = 0; i < n; ++i) ms_ticker_detach(tk, ch[i].src);
		if (peered) { /* audio_stream_free's order (audiostream.c:357-358): the PEER (volrecv) is destroyed before the filter that named it */
			chain_destroy(&ch[1]);
			chain_destroy(&ch[0]);
			ch[0] = ch[n - 1], --n;
			if (n > 1) ch[1] = ch[n - 1], --n;
			else n = 0;
		}
		if (ec) {
			char *state = NULL;
			ms_ticker_detach(tk, ec);
			ms_filter_call_method(ec, MS_ECHO_CANCELLER_GET_STATE_STRING, &state); /* on a detached filter: a hub that never gets a bank */
			for (int k = 0; k < 2; ++k) {
				ms_filter_unlink(ecs[k], 0, ec, k);
				ms_filter_unlink(ec, k, eck[k], 0);
				ms_filter_destroy(ecs[k]);
				ms_filter_destroy(eck[k]);
			}
			ms_filter_destroy(ec);
		}
		if (mx) {
			ms_ticker_detach(tk, mx);
			for (int k = 0; k < 3; ++k) {
				ms_filter_unlink(mxs[k], 0, mx, k);
				ms_filter_unlink(mx, k, mxk[k], 0);
				ms_filter_destroy(mxs[k]);
				ms_filter_destroy(mxk[k]);
			}
			ms_filter_destroy(mx);
		}
		for (int i = 0; i < n; ++i) chain_destroy(&ch[i]);
		ms_ticker_destroy(tk);
	}
	return NULL;
}

static void *grower(void *arg) {
	enum { NF = 150 };
	int16_t block[480];
	(void)arg;
	for (int i = 0; i < 480; ++i) block[i] = (int16_t)(i * 13 % 3000 - 1500);
	for (int rep = 0; rep < (g_rounds + 3) / 4; ++rep) {
		MSTicker *tk = ms_ticker_new();
		MSFilter *src[NF], *vol[NF], *snk[NF];
		for (int i = 0; i < NF; ++i) { /* banks of 16, 64 and 256 slots open as the filters come */
			src[i] = ms2shim_new_source(g_fac);
			vol[i] = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			snk[i] = ms2shim_new_sink(g_fac);
			ms2shim_sink_set_discard(snk[i], 1);
			set_int(vol[i], MS_FILTER_SET_SAMPLE_RATE, 48000);
			ms_filter_link(src[i], 0, vol[i], 0);
			ms_filter_link(vol[i], 0, snk[i], 0);
			CHECK(ms_ticker_attach(tk, src[i]) == 0);
			if (i % 25 == 24) {
				for (int k = 0; k <= i; ++k) ms2shim_source_push(src[k], block, sizeof block);
				ms_ticker_step(tk);
			}
		}
		for (int t = 0; t < 3; ++t) {
			for (int i = 0; i < NF; ++i) ms2shim_source_push(src[i], block, sizeof block);
			ms_ticker_step(tk);
		}
		for (int i = 0; i < NF; i += 2) { /* every other call ends: slots return, a bank may empty */
			ms_ticker_detach(tk, src[i]);
			ms_filter_unlink(src[i], 0, vol[i], 0);
			ms_filter_unlink(vol[i], 0, snk[i], 0);
			ms_filter_destroy(src[i]), ms_filter_destroy(vol[i]), ms_filter_destroy(snk[i]);
		}
		for (int t = 0; t < 3; ++t) {
			for (int i = 1; i < NF; i += 2) ms2shim_source_push(src[i], block, sizeof block);
			ms_ticker_step(tk);
		}
		for (int i = 1; i < NF; i += 2) {
			ms_ticker_detach(tk, src[i]);
			ms_filter_unlink(src[i], 0, vol[i], 0);
			ms_filter_unlink(vol[i], 0, snk[i], 0);
			ms_filter_destroy(src[i]), ms_filter_destroy(vol[i]), ms_filter_destroy(snk[i]);
		}
		ms_ticker_destroy(tk);
	}
	return NULL;
}

typedef struct {
	MSFilter *mic, *far, *rs, *ec, *vol, *spk, *out;
} leg_t;

/* Methods from the APPLICATION's thread while the ticker thread walks (d->leg / s->leg must only be read under the
 * hub's lock -- the walking thread un-fuses and deletes legs).  Topology changes (detach / attach / link) are the application's
 * too and never overlap its own method calls: g_topo serialises the two, the walk itself runs free. */
typedef struct {
	pthread_mutex_t topo;
	volatile int stop;
	MSFilter *vol_conf, *vol_other, *ec_conf, *ec_bypass, *vol_solo, *volrecv, *mic_eq;
} meddle_t;
static void *meddler(void *arg) {
	meddle_t *m = (meddle_t *)arg;
	int k = 0;
	while (!__atomic_load_n(&m->stop, __ATOMIC_SEQ_CST)) {
		float g = 0.4f + 0.05f * (float)(k % 8), v = 0;
		char *state = NULL;
		int off = 0, on = 1;
		bool_t byp = (bool_t)(k % 5 == 4);
		pthread_mutex_lock(&m->topo);
		ms_filter_call_method(m->vol_conf, MS_VOLUME_SET_GAIN, &g);
		ms_filter_call_method(m->vol_other, MS_VOLUME_GET, &v);
		ms_filter_call_method(m->ec_conf, MS_ECHO_CANCELLER_GET_STATE_STRING, &state);
		ms_filter_call_method(m->vol_solo, MS_VOLUME_ENABLE_AGC, k % 2 ? &on : &off); /* the leg leaves its batch ... */
		ms_filter_call_method(m->vol_solo, MS_VOLUME_SET_GAIN, &g);                   /* ... and is addressed again at once */
		ms_filter_call_method(m->vol_solo, MS_VOLUME_GET_LINEAR, &v);
		ms_filter_call_method(m->volrecv, MS_VOLUME_GET, &v); /* the metered echo-limiter peer of a fused leg */
		ms_filter_call_method(m->volrecv, MS_VOLUME_GET_MAX, &v);
		{ /* the mic_equalizer of a fused leg: gains and the active switch from the application's thread, its state read back */
			MSEqualizerGain eg;
			float dump[256];
			int act = k % 3 != 0;
			eg.frequency = 1500.f + 100.f * (float)(k % 5), eg.gain = 0.5f + 0.25f * (float)(k % 4), eg.width = 700.f;
			ms_filter_call_method(m->mic_eq, MS_EQUALIZER_SET_GAIN, &eg);
			ms_filter_call_method(m->mic_eq, MS_EQUALIZER_SET_ACTIVE, &act);
			ms_filter_call_method(m->mic_eq, MS_EQUALIZER_DUMP_STATE, dump);
		}
		ms_filter_call_method(m->ec_bypass, MS_ECHO_CANCELLER_SET_BYPASS_MODE, &byp);
		pthread_mutex_unlock(&m->topo);
		++k;
		sched_yield();
	}
	return NULL;
}

static void *conferences(void *arg) {
	enum { NC = 2, NM = 5 };
	int16_t mic[160], far[480];
	void (*p_fused)(int *, int *, unsigned long long *, unsigned long long *) = (void (*)(int *, int *, unsigned long long *, unsigned long long *))arg;
	for (int i = 0; i < 160; ++i) mic[i] = (int16_t)(i * 91 % 4000 - 2000);
	for (int i = 0; i < 480; ++i) far[i] = (int16_t)(i * 57 % 6000 - 3000);
	for (int rep = 0; rep < (g_rounds + 1) / 2; ++rep) {
		MSTicker *tk = ms_ticker_new();
		MSFilter *mx[NC];
		leg_t leg[NC][NM];
		for (int c = 0; c < NC; ++c) {
			mx[c] = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
			set_int(mx[c], MS_FILTER_SET_SAMPLE_RATE, 48000);
			set_int(mx[c], MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
			for (int k = 0; k < NM; ++k) {
				leg_t *l = &leg[c][k];
				l->mic = ms2shim_new_source(g_fac), l->far = ms2shim_new_source(g_fac);
				l->spk = ms2shim_new_sink(g_fac), l->out = ms2shim_new_sink(g_fac);
				l->rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
				l->ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID);
				l->vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
				ms2shim_sink_set_discard(l->spk, 1), ms2shim_sink_set_discard(l->out, 1);
				set_int(l->rs, MS_FILTER_SET_SAMPLE_RATE, 16000), set_int(l->rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
				set_int(l->ec, MS_FILTER_SET_SAMPLE_RATE, 48000), set_int(l->ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 64);
				set_int(l->vol, MS_FILTER_SET_SAMPLE_RATE, 48000), set_int(l->vol, MS_VOLUME_ENABLE_AGC, 1);
				ms_filter_link(l->mic, 0, l->rs, 0), ms_filter_link(l->rs, 0, l->ec, 1), ms_filter_link(l->ec, 1, l->vol, 0);
				ms_filter_link(l->vol, 0, mx[c], k), ms_filter_link(mx[c], k, l->out, 0);
				ms_filter_link(l->far, 0, l->ec, 0), ms_filter_link(l->ec, 0, l->spk, 0);
			}
			CHECK(ms_ticker_attach(tk, mx[c]) == 0);
		}
		/* three call legs WITHOUT a mixer on the same ticker (an AudioStream's sending side): fused leg by leg */
		leg_t solo[3];
		/* leg 1 is a default AudioStream with the echo limiter on (audiostream.c:2236-2240): volrecv upstream of the canceller's far end,
		 * named as volsend's peer -- metered beside the fused leg, its blocks handed on in the walk */
		MSFilter *volrecv = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
		set_int(volrecv, MS_FILTER_SET_SAMPLE_RATE, 48000);
		/* leg 2 carries a mic_equalizer between its MSResample and its canceller (audiostream.c:1801): it moves into the leg's bank */
		/* leg 0's far end comes through a plain volrecv and the application's recv_tee (audiostream.c:1826-1828): a meter only, it hands
		 * its blocks on in the walk -- until the meddler gives it a gain, which sends the leg back to its facades */
		MSFilter *volrecv0 = ms_factory_create_filter(g_fac, MS_VOLUME_ID), *tee0 = ms2shim_new_pass(g_fac);
		set_int(volrecv0, MS_FILTER_SET_SAMPLE_RATE, 48000);
		MSFilter *mic_eq = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID);
		set_int(mic_eq, MS_FILTER_SET_SAMPLE_RATE, 48000);
		for (int k = 0; k < 3; ++k) {
			leg_t *l = &solo[k];
			l->mic = ms2shim_new_source(g_fac), l->far = ms2shim_new_source(g_fac);
			l->spk = ms2shim_new_sink(g_fac), l->out = ms2shim_new_sink(g_fac);
			l->rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
			l->ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID);
			l->vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			ms2shim_sink_set_discard(l->spk, 1), ms2shim_sink_set_discard(l->out, 1);
			set_int(l->rs, MS_FILTER_SET_SAMPLE_RATE, 16000), set_int(l->rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
			set_int(l->ec, MS_FILTER_SET_SAMPLE_RATE, 48000), set_int(l->ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 64);
			set_int(l->vol, MS_FILTER_SET_SAMPLE_RATE, 48000);
			if (k != 1) set_int(l->vol, MS_VOLUME_ENABLE_AGC, 1); /* leg 1: MSVolume without AGC (the default): fused into a bank of the other kind */
			if (k == 0) ms_filter_link(l->mic, 0, l->ec, 1); /* no MSResample in front: MSSpeexEC is this leg's head (48 kHz microphone) */
			else if (k == 2) ms_filter_link(l->mic, 0, l->rs, 0), ms_filter_link(l->rs, 0, mic_eq, 0), ms_filter_link(mic_eq, 0, l->ec, 1);
			else ms_filter_link(l->mic, 0, l->rs, 0), ms_filter_link(l->rs, 0, l->ec, 1);
			ms_filter_link(l->ec, 1, l->vol, 0);
			ms_filter_link(l->vol, 0, l->out, 0), ms_filter_link(l->ec, 0, l->spk, 0);
			if (k == 1) {
				ms_filter_link(l->far, 0, volrecv, 0), ms_filter_link(volrecv, 0, l->ec, 0);
				CHECK(ms_filter_call_method(l->vol, MS_VOLUME_SET_PEER, volrecv) == 0);
			} else if (k == 0) {
				ms_filter_link(l->far, 0, volrecv0, 0), ms_filter_link(volrecv0, 0, tee0, 0), ms_filter_link(tee0, 0, l->ec, 0);
			} else ms_filter_link(l->far, 0, l->ec, 0);
			CHECK(ms_ticker_attach(tk, l->mic) == 0);
		}
		meddle_t med;
		pthread_t med_th;
		pthread_mutex_init(&med.topo, NULL);
		med.stop = 0;
		med.vol_conf = leg[0][1].vol, med.vol_other = leg[1][0].vol, med.ec_conf = leg[0][0].ec, med.ec_bypass = leg[1][3].ec, med.vol_solo = solo[2].vol, med.volrecv = volrecv, med.mic_eq = mic_eq;
		for (int t = 0; t < 14; ++t) {
			if (t == 3) CHECK(pthread_create(&med_th, NULL, meddler, &med) == 0); /* (after the census at t == 2) */
			for (int k = 0; k < 3; ++k) {
				if (k == 0) ms2shim_source_push(solo[k].mic, far, sizeof far); /* (a 48 kHz block) */
				else ms2shim_source_push(solo[k].mic, mic, sizeof mic);
				if (t != 5 || k != 1) ms2shim_source_push(solo[k].far, far, sizeof far); /* a far end that skips a tick */
			}
			if (t == 7) { /* that leg goes back to its facades */
				int off = 0;
				pthread_mutex_lock(&med.topo);
				ms_filter_call_method(solo[2].vol, MS_VOLUME_ENABLE_AGC, &off);
				pthread_mutex_unlock(&med.topo);
			}
			for (int c = 0; c < NC; ++c)
				for (int k = 0; k < NM; ++k) {
					ms2shim_source_push(leg[c][k].mic, mic, sizeof mic);
					ms2shim_source_push(leg[c][k].far, far, sizeof far);
				}
			ms_ticker_step(tk);
			pthread_mutex_lock(&med.topo); /* (the application's own calls from here to the end of the tick) */
			if (t == 2) { /* both conferences live in the fused batch by now */
				int nc = 0, nl = 0;
				p_fused(&nc, &nl, NULL, NULL);
				CHECK(nc >= NC && nl >= NC * NM);
			}
			if (t == 4) { float g = 0.7f; ms_filter_call_method(leg[0][1].vol, MS_VOLUME_SET_GAIN, &g); } /* a method between ticks */
			if (t == 6) { bool_t on = TRUE; ms_filter_call_method(leg[1][2].ec, MS_ECHO_CANCELLER_SET_BYPASS_MODE, &on); } /* conference 1 leaves the batch */
			if (t == 9) { /* the whole graph of conference 0 detached and attached again: fuses again */
				ms_ticker_detach(tk, mx[0]);
				CHECK(ms_ticker_attach(tk, mx[0]) == 0);
			}
			/* ms_audio_conference_remove_member / add_member (audioconference.c:322-374): the conference graph detached, a member from
			 * the middle un-plumbed, attached again (MSVolume's state and queued samples move from slot to slot, the tick in flight is
			 * delivered); a tick later it joins again on the pin it gave up */
			if (t == 10) {
				ms_ticker_detach(tk, mx[0]);
				ms_filter_unlink(leg[0][2].vol, 0, mx[0], 2), ms_filter_unlink(mx[0], 2, leg[0][2].out, 0);
				CHECK(ms_ticker_attach(tk, mx[0]) == 0);
			}
			if (t == 11) {
				ms_ticker_detach(tk, mx[0]);
				ms_filter_link(leg[0][2].vol, 0, mx[0], 2), ms_filter_link(mx[0], 2, leg[0][2].out, 0);
				CHECK(ms_ticker_attach(tk, mx[0]) == 0);
			}
			if (t == 6) { float g = 0.8f; ms_filter_call_method(volrecv0, MS_VOLUME_SET_GAIN, &g); } /* the speaker's volume: leg 0's far end now comes with the flush */
			if (t == 8) { float g = 0.5f; ms_filter_call_method(volrecv, MS_VOLUME_SET_GAIN, &g); } /* no longer a meter only: its leg goes back to the facades */
			if (t == 12) { /* a leg without a mixer detached and attached again (its chunks and speaker frames in flight are handed on) */
				ms_ticker_detach(tk, solo[0].mic);
				CHECK(ms_ticker_attach(tk, solo[0].mic) == 0);
			}
			pthread_mutex_unlock(&med.topo);
		}
		__atomic_store_n(&med.stop, 1, __ATOMIC_SEQ_CST);
		pthread_join(med_th, NULL);
		pthread_mutex_destroy(&med.topo);
		for (int k = 0; k < 3; ++k) {
			leg_t *l = &solo[k];
			ms_ticker_detach(tk, l->mic);
			if (k == 0) ms_filter_unlink(l->mic, 0, l->ec, 1);
			else if (k == 2) {
				ms_filter_unlink(l->mic, 0, l->rs, 0), ms_filter_unlink(l->rs, 0, mic_eq, 0), ms_filter_unlink(mic_eq, 0, l->ec, 1);
				ms_filter_destroy(mic_eq);
			} else ms_filter_unlink(l->mic, 0, l->rs, 0), ms_filter_unlink(l->rs, 0, l->ec, 1);
			ms_filter_unlink(l->ec, 1, l->vol, 0);
			ms_filter_unlink(l->vol, 0, l->out, 0), ms_filter_unlink(l->ec, 0, l->spk, 0);
			if (k == 1) {
				ms_filter_unlink(l->far, 0, volrecv, 0), ms_filter_unlink(volrecv, 0, l->ec, 0);
				ms_filter_destroy(volrecv); /* (audio_stream_free destroys volrecv before volsend, audiostream.c:357-358) */
			} else if (k == 0) {
				ms_filter_unlink(l->far, 0, volrecv0, 0), ms_filter_unlink(volrecv0, 0, tee0, 0), ms_filter_unlink(tee0, 0, l->ec, 0);
				ms_filter_destroy(volrecv0), ms_filter_destroy(tee0);
			} else ms_filter_unlink(l->far, 0, l->ec, 0);
			ms_filter_destroy(l->mic), ms_filter_destroy(l->far), ms_filter_destroy(l->rs), ms_filter_destroy(l->ec);
			ms_filter_destroy(l->vol), ms_filter_destroy(l->spk), ms_filter_destroy(l->out);
		}
		for (int c = 0; c < NC; ++c) {
			ms_ticker_detach(tk, mx[c]);
			for (int k = 0; k < NM; ++k) {
				leg_t *l = &leg[c][k];
				ms_filter_unlink(l->mic, 0, l->rs, 0), ms_filter_unlink(l->rs, 0, l->ec, 1), ms_filter_unlink(l->ec, 1, l->vol, 0);
				ms_filter_unlink(l->vol, 0, mx[c], k), ms_filter_unlink(mx[c], k, l->out, 0);
				ms_filter_unlink(l->far, 0, l->ec, 0), ms_filter_unlink(l->ec, 0, l->spk, 0);
				ms_filter_destroy(l->mic), ms_filter_destroy(l->far), ms_filter_destroy(l->rs), ms_filter_destroy(l->ec);
				ms_filter_destroy(l->vol), ms_filter_destroy(l->spk), ms_filter_destroy(l->out);
			}
			ms_filter_destroy(mx[c]);
		}
		ms_ticker_destroy(tk);
	}
	return NULL;
}

/* A conference SERVER's members (filters/server_leg.inl): source -> MSVolume -> in_resampler -> mixer pin -> out_resampler -> MSUlawEnc ->
 * sink at 8 kHz, one PCM listener; fused into a ServerBank, methods between ticks, a burst of blocks beyond the launch rounds, a
 * re-plumbing in mid-packet, AGC switched on (the conference leaves its batch), teardown with the encoders destroyed first. */
static void *server_conferences(void *arg) {
	enum { NM = 6 };
	int16_t pcm[80];
	void (*p_fused)(int *, int *, unsigned long long *, unsigned long long *) = (void (*)(int *, int *, unsigned long long *, unsigned long long *))arg;
	for (int i = 0; i < 80; ++i) pcm[i] = (int16_t)(i * 131 % 5000 - 2500);
	for (int rep = 0; rep < (g_rounds + 1) / 2; ++rep) {
		MSTicker *tk = ms_ticker_new();
		MSFilter *mx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
		MSFilter *src[NM], *vol[NM], *irs[NM], *ors[NM], *enc[NM], *snk[NM], *dec[NM], *tap = ms2shim_new_sink(g_fac);
		uint8_t codes[80];
		for (int i = 0; i < 80; ++i) codes[i] = (uint8_t)(i * 37 + rep);
		const int crate = (rep % 2) ? 16000 : 8000; /* every other round the conference runs at 16 kHz: the G.711 endpoints' resamplers work, in the batch */
		set_int(mx, MS_FILTER_SET_SAMPLE_RATE, crate);
		set_int(mx, MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
		ms2shim_sink_set_discard(tap, 1);
		for (int k = 0; k < NM; ++k) {
			src[k] = ms2shim_new_source(g_fac), snk[k] = ms2shim_new_sink(g_fac);
			vol[k] = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			irs[k] = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID), ors[k] = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID);
			enc[k] = ms_factory_create_filter(g_fac, k % 2 ? MS_ALAW_ENC_ID : MS_ULAW_ENC_ID);
			dec[k] = (k == 0 || k == 3) ? ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID) : NULL; /* these members' packets are decoded in the batch (the decoder heads the leg) */
			CHECK(src[k] && snk[k] && vol[k] && irs[k] && ors[k] && enc[k]);
			ms2shim_sink_set_discard(snk[k], 1);
			set_int(vol[k], MS_FILTER_SET_SAMPLE_RATE, 8000);
			set_int(irs[k], MS_FILTER_SET_SAMPLE_RATE, 8000), set_int(irs[k], MS_FILTER_SET_OUTPUT_SAMPLE_RATE, crate);
			set_int(ors[k], MS_FILTER_SET_SAMPLE_RATE, crate), set_int(ors[k], MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 8000);
			if (dec[k]) ms_filter_link(src[k], 0, dec[k], 0), ms_filter_link(dec[k], 0, vol[k], 0);
			else ms_filter_link(src[k], 0, vol[k], 0);
			ms_filter_link(vol[k], 0, irs[k], 0), ms_filter_link(irs[k], 0, mx, k);
			ms_filter_link(mx, k, ors[k], 0), ms_filter_link(ors[k], 0, enc[k], 0), ms_filter_link(enc[k], 0, snk[k], 0);
		}
		ms_filter_link(mx, NM + 1, tap, 0); /* a listener above the members */
		CHECK(ms_ticker_attach(tk, mx) == 0);
		for (int t = 0; t < 16; ++t) {
			for (int k = 0; k < NM; ++k) {
				const int blocks = (t == 6 && k == 2) ? 7 : 1; /* a burst: more blocks in one tick than the bank has launch rounds */
				for (int b = 0; b < blocks; ++b) {
					if (dec[k]) ms2shim_source_push(src[k], codes, sizeof codes);
					else ms2shim_source_push(src[k], pcm, sizeof pcm);
				}
			}
			ms_ticker_step(tk);
			if (t == 2) {
				int nc = 0, nl = 0;
				p_fused(&nc, &nl, NULL, NULL);
				CHECK(nc >= 1 && nl >= NM);
			}
			if (t == 4) { float g = 0.5f; ms_filter_call_method(vol[1], MS_VOLUME_SET_GAIN, &g); }
			if (t == 5) { MSAudioMixerCtl ctl; ctl.pin = 3; ctl.param.active = 0; ms_filter_call_method(mx, MS_AUDIO_MIXER_SET_ACTIVE, &ctl); }
			if (t == 9) { /* re-plumbed with half a packet filled: fuses again */
				ms_ticker_detach(tk, mx);
				CHECK(ms_ticker_attach(tk, mx) == 0);
			}
			if (t == 12) { int on = 1; ms_filter_call_method(vol[4], MS_VOLUME_ENABLE_AGC, &on); } /* the conference leaves its batch */
		}
		ms_ticker_detach(tk, mx);
		for (int k = 0; k < NM; ++k) { /* the encoders go first */
			ms_filter_unlink(ors[k], 0, enc[k], 0), ms_filter_unlink(enc[k], 0, snk[k], 0);
			ms_filter_destroy(enc[k]);
		}
		for (int k = 0; k < NM; ++k) {
			if (dec[k]) {
				ms_filter_unlink(src[k], 0, dec[k], 0), ms_filter_unlink(dec[k], 0, vol[k], 0);
				ms_filter_destroy(dec[k]);
			} else ms_filter_unlink(src[k], 0, vol[k], 0);
			ms_filter_unlink(vol[k], 0, irs[k], 0), ms_filter_unlink(irs[k], 0, mx, k), ms_filter_unlink(mx, k, ors[k], 0);
			ms_filter_destroy(src[k]), ms_filter_destroy(vol[k]), ms_filter_destroy(irs[k]), ms_filter_destroy(ors[k]), ms_filter_destroy(snk[k]);
		}
		ms_filter_unlink(mx, NM + 1, tap, 0);
		ms_filter_destroy(tap);
		ms_filter_destroy(mx);
		ms_ticker_destroy(tk);
	}
	return NULL;
}

/* Full-duplex G.711 AudioStreams as audiostream.c:1796-1832 plumbs them with the default features (filters/recv_leg.inl, the encoder in the leg's
 * batch): packets -> MSUlawDec -> local_mixer (one input) -> MSGenericPLC -> MSAudioFlowControl -> dtmfgen (the application's) -> volrecv ->
 * MSSpeexEC pin 0 -> speaker;  microphone -> MSSpeexEC pin 1 -> volsend -> outbound_mixer (one input) -> MSUlawEnc -> packets.  Fused at the attach
 * (on THIS thread, while other tickers run); both equalizers of AUDIO_STREAM_FEATURE_EQUALIZER in place and inactive (mic_equalizer in front of pin 1,
 * spk_equalizer in front of pin 0: transparent, no lock taken); lost packets, a drop request, a PLC rate change, an equalizer switched on and a
 * mixer's output disabled from ANOTHER thread during the walk, a re-plumbing,
 * teardown with the decoder / the PLC / the encoder destroyed first in turn. */
typedef struct {
	MSFilter *fc, *plc, *vol, *spk_eq, *omx;
	volatile int stop;
} stream_meddle_t;
static void *stream_meddler(void *arg) {
	stream_meddle_t *m = (stream_meddle_t *)arg;
	int n = 0;
	while (!__atomic_load_n(&m->stop, __ATOMIC_SEQ_CST)) {
		MSAudioFlowControlDropEvent ev;
		float g = (n & 1) ? 0.8f : 1.0f;
		ev.flow_control_interval_ms = 200, ev.drop_ms = 10;
		ms_filter_call_method(m->fc, MS_AUDIO_FLOW_CONTROL_DROP, &ev);
		ms_filter_call_method(m->vol, MS_VOLUME_SET_GAIN, &g);
		if (n == 15) { /* an equalizer that has handed its blocks on without a lock since the attach is switched on in mid-walk: its leg goes back to its facades */
			int on = 1;
			ms_filter_call_method(m->spk_eq, MS_EQUALIZER_SET_ACTIVE, &on);
		}
		if (n == 25 || n == 30) { /* the same for a one-input mixer: an output disabled and enabled again -- under the locks from then on */
			MSAudioMixerCtl ctl;
			ctl.pin = 0, ctl.param.enabled = n == 30;
			ms_filter_call_method(m->omx, MS_AUDIO_MIXER_ENABLE_OUTPUT, &ctl);
		}
		if (++n == 40) { /* another rate and back: the stream's receiving side leaves its batch at the next walk and carries on at 8 kHz on its facades */
			set_int(m->plc, MS_FILTER_SET_SAMPLE_RATE, 16000);
			set_int(m->plc, MS_FILTER_SET_SAMPLE_RATE, 8000);
		}
		usleep(200);
	}
	return NULL;
}
static void *audiostreams(void *arg) {
	enum { NS_ = 5 };
	void (*p_fused)(int *, int *, unsigned long long *, unsigned long long *) = (void (*)(int *, int *, unsigned long long *, unsigned long long *))arg;
	int16_t pcm[80];
	uint8_t codes[80];
	for (int i = 0; i < 80; ++i) pcm[i] = (int16_t)(i * 211 % 6000 - 3000), codes[i] = (uint8_t)(i * 29 + 7);
	for (int rep = 0; rep < (g_rounds + 1) / 2; ++rep) {
		MSTicker *tk = ms_ticker_new();
		MSFilter *mic[NS_], *far[NS_], *spk[NS_], *out[NS_], *dec[NS_], *lmx[NS_], *plc[NS_], *fc[NS_], *dtmf[NS_], *vr[NS_], *ec[NS_], *vs[NS_], *omx[NS_], *enc[NS_], *meq[NS_], *seq[NS_];
		for (int k = 0; k < NS_; ++k) {
			mic[k] = ms2shim_new_source(g_fac), far[k] = ms2shim_new_source(g_fac), spk[k] = ms2shim_new_sink(g_fac), out[k] = ms2shim_new_sink(g_fac);
			dec[k] = ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID), lmx[k] = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
			plc[k] = ms_factory_create_filter(g_fac, MS_GENERIC_PLC_ID), fc[k] = ms_factory_create_filter(g_fac, MS_AUDIO_FLOW_CONTROL_ID);
			dtmf[k] = ms2shim_new_pass(g_fac), vr[k] = ms_factory_create_filter(g_fac, MS_VOLUME_ID), ec[k] = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID);
			vs[k] = ms_factory_create_filter(g_fac, MS_VOLUME_ID), omx[k] = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID), enc[k] = ms_factory_create_filter(g_fac, MS_ULAW_ENC_ID);
			meq[k] = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID), seq[k] = ms_factory_create_filter(g_fac, MS_EQUALIZER_ID);
			CHECK(dec[k] && lmx[k] && plc[k] && fc[k] && vr[k] && ec[k] && vs[k] && omx[k] && enc[k] && meq[k] && seq[k]);
			for (MSFilter **f = (MSFilter *[]){meq[k], seq[k], NULL}; *f; ++f) {
				int off = 0;
				set_int(*f, MS_FILTER_SET_SAMPLE_RATE, 8000);
				ms_filter_call_method(*f, MS_EQUALIZER_SET_ACTIVE, &off);
			}
			ms2shim_sink_set_discard(spk[k], 1), ms2shim_sink_set_discard(out[k], 1);
			for (MSFilter **f = (MSFilter *[]){lmx[k], plc[k], fc[k], vr[k], ec[k], vs[k], omx[k], NULL}; *f; ++f) set_int(*f, MS_FILTER_SET_SAMPLE_RATE, 8000);
			set_int(ec[k], MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 64);
			ms_filter_link(far[k], 0, dec[k], 0), ms_filter_link(dec[k], 0, lmx[k], 0), ms_filter_link(lmx[k], 0, plc[k], 0), ms_filter_link(plc[k], 0, fc[k], 0);
			ms_filter_link(fc[k], 0, dtmf[k], 0), ms_filter_link(dtmf[k], 0, vr[k], 0), ms_filter_link(vr[k], 0, seq[k], 0), ms_filter_link(seq[k], 0, ec[k], 0), ms_filter_link(ec[k], 0, spk[k], 0);
			ms_filter_link(mic[k], 0, meq[k], 0), ms_filter_link(meq[k], 0, ec[k], 1), ms_filter_link(ec[k], 1, vs[k], 0), ms_filter_link(vs[k], 0, omx[k], 0), ms_filter_link(omx[k], 0, enc[k], 0);
			ms_filter_link(enc[k], 0, out[k], 0);
			CHECK(ms_ticker_attach(tk, mic[k]) == 0);
		}
		stream_meddle_t md = {fc[1], plc[2], vs[3], seq[4], omx[2], 0}; /* (not stream 0: that one is re-plumbed by THIS thread at t == 11, and an application does not call a filter while it attaches it) */
		pthread_t mt;
		for (int t = 0; t < 24; ++t) {
			for (int k = 0; k < NS_; ++k) {
				ms2shim_source_push(mic[k], pcm, sizeof pcm);
				if ((t + k) % 7 != 3) ms2shim_source_push(far[k], codes, sizeof codes); /* (a packet lost now and then: concealed in the batch) */
			}
			ms_ticker_step(tk);
			if (t == 3) {
				int nl = 0;
				p_fused(NULL, &nl, NULL, NULL);
				CHECK(nl >= NS_);
				for (int k = 0; k < NS_; ++k) /* (the count above is the process's: these are THIS thread's streams, both directions) */
					CHECK(p_in_batch(ec[k]) == 1 && p_in_batch(enc[k]) == 1 && p_in_batch(dec[k]) == 1 && p_in_batch(plc[k]) == 1 && p_in_batch(fc[k]) == 1);
				pthread_create(&mt, NULL, stream_meddler, &md); /* (from here on legs leave their batch as the meddler's methods land) */
			}
			if (t == 11) { /* one stream re-plumbed under the others' feet */
				ms_ticker_detach(tk, mic[0]);
				CHECK(ms_ticker_attach(tk, mic[0]) == 0);
			}
		}
		__atomic_store_n(&md.stop, 1, __ATOMIC_SEQ_CST);
		pthread_join(mt, NULL);
		for (int k = 0; k < NS_; ++k) ms_ticker_detach(tk, mic[k]);
		for (int k = 0; k < NS_; ++k) {
			MSFilter *all[] = {mic[k], far[k], spk[k], out[k], dec[k], lmx[k], plc[k], fc[k], dtmf[k], vr[k], ec[k], vs[k], omx[k], enc[k], meq[k], seq[k]};
			MSFilter *first = (k % 3 == 0) ? dec[k] : ((k % 3 == 1) ? plc[k] : enc[k]); /* (whoever goes first, nobody reaches into a freed filter) */
			ms_filter_unlink(far[k], 0, dec[k], 0), ms_filter_unlink(dec[k], 0, lmx[k], 0), ms_filter_unlink(lmx[k], 0, plc[k], 0), ms_filter_unlink(plc[k], 0, fc[k], 0);
			ms_filter_unlink(fc[k], 0, dtmf[k], 0), ms_filter_unlink(dtmf[k], 0, vr[k], 0), ms_filter_unlink(vr[k], 0, seq[k], 0), ms_filter_unlink(seq[k], 0, ec[k], 0), ms_filter_unlink(ec[k], 0, spk[k], 0);
			ms_filter_unlink(mic[k], 0, meq[k], 0), ms_filter_unlink(meq[k], 0, ec[k], 1), ms_filter_unlink(ec[k], 1, vs[k], 0), ms_filter_unlink(vs[k], 0, omx[k], 0), ms_filter_unlink(omx[k], 0, enc[k], 0);
			ms_filter_unlink(enc[k], 0, out[k], 0);
			ms_filter_destroy(first);
			for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i)
				if (all[i] != first) ms_filter_destroy(all[i]);
		}
		ms_ticker_destroy(tk);
	}
	return NULL;
}

/* A ticker that is re-plumbed BY THE APPLICATION while it runs, as the reference's callers do it (ms_audio_conference_add_member / remove_member,
 * audioconference.c:322-374; audio_stream_stop): ms_ticker_detach and ms_ticker_attach on a thread of their own -- the detach takes the ticker's
 * lock to take the graph's sources out and runs the postprocess calls with the lock released, the attach runs the preprocess calls (this plugin's
 * fusing: banks joined, queues moved to the device) while the ticker walks its other graphs and takes the lock to splice the sources in
 * (msticker.c:153-221,:462-493).  Three conferences of four legs, two AudioStreams and a server's bridge of four G.711 members on one ticker; the
 * application takes them out and puts them back in turn, as fast as it can, while the ticker ticks. */
void ms2shim_source_set_loop(MSFilter *src, const void *ring, size_t block_bytes, int nblocks, int phase);
typedef struct {
	MSTicker *tk;
	MSFilter *roots[6];
	volatile int stop;
	int done;
} replumber_t;
static void *replumber(void *arg) {
	replumber_t *r = (replumber_t *)arg;
	for (int n = 0; !__atomic_load_n(&r->stop, __ATOMIC_SEQ_CST); ++n) {
		/* the conferences in turn; each AudioStream twice (its PLC makes up for every tick the stream was away with a block of its own,
		 * msgenericplc.c:117-166 -- re-plumbed every millisecond its far end would outrun the microphone and the canceller's delay line
		 * drop, and count, the excess) */
		MSFilter *root = (n == 6 || n == 21) ? r->roots[1] : ((n == 11 || n == 31) ? r->roots[3] : (n % 4 == 3 ? r->roots[5] : r->roots[(n % 4) * 2]));
		ms_ticker_detach(r->tk, root);
		CHECK(ms_ticker_attach(r->tk, root) == 0);
		r->done++;
		usleep(150);
	}
	return NULL;
}
static void *replumbed_by_the_application(void *arg) {
	enum { NC2 = 3, NM2 = 4, NS2 = 2 };
	static int16_t ring16[4][160], ring8[4][80];
	static uint8_t codes[4][80];
	(void)arg;
	for (int b = 0; b < 4; ++b) {
		for (int i = 0; i < 160; ++i) ring16[b][i] = (int16_t)((i * 97 + b * 911) % 5000 - 2500);
		for (int i = 0; i < 80; ++i) ring8[b][i] = (int16_t)((i * 211 + b * 577) % 6000 - 3000), codes[b][i] = (uint8_t)(i * 29 + b * 7);
	}
	for (int rep = 0; rep < (g_rounds + 2) / 3; ++rep) {
		MSTicker *tk = ms_ticker_new();
		MSFilter *mx[NC2], *all[NC2 * NM2 * 7 + NC2 + NS2 * 14 + 1 + NM2 * 5], *smx;
		int nall = 0;
		for (int c = 0; c < NC2; ++c) {
			mx[c] = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
			all[nall++] = mx[c];
			set_int(mx[c], MS_FILTER_SET_SAMPLE_RATE, 48000), set_int(mx[c], MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
			for (int k = 0; k < NM2; ++k) {
				MSFilter *mic = ms2shim_new_source(g_fac), *far = ms2shim_new_source(g_fac), *spk = ms2shim_new_sink(g_fac), *out = ms2shim_new_sink(g_fac);
				MSFilter *rs = ms_factory_create_filter(g_fac, MS_RESAMPLE_ID), *ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID), *vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
				MSFilter *seven[] = {mic, far, spk, out, rs, ec, vol};
				for (int i = 0; i < 7; ++i) all[nall++] = seven[i];
				ms2shim_source_set_loop(mic, ring16, sizeof ring16[0], 4, c + k), ms2shim_source_set_loop(far, ring16, sizeof ring16[0], 4, c + k + 2);
				ms2shim_sink_set_discard(spk, 1), ms2shim_sink_set_discard(out, 1);
				set_int(rs, MS_FILTER_SET_SAMPLE_RATE, 16000), set_int(rs, MS_FILTER_SET_OUTPUT_SAMPLE_RATE, 48000);
				set_int(ec, MS_FILTER_SET_SAMPLE_RATE, 48000), set_int(ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 64);
				set_int(vol, MS_FILTER_SET_SAMPLE_RATE, 48000);
				if (c != 1) set_int(vol, MS_VOLUME_ENABLE_AGC, 1); /* (conference 1 without AGC: the other kind of bank) */
				ms_filter_link(mic, 0, rs, 0), ms_filter_link(rs, 0, ec, 1), ms_filter_link(ec, 1, vol, 0), ms_filter_link(vol, 0, mx[c], k), ms_filter_link(mx[c], k, out, 0);
				ms_filter_link(far, 0, ec, 0), ms_filter_link(ec, 0, spk, 0);
			}
			CHECK(ms_ticker_attach(tk, mx[c]) == 0);
		}
		MSFilter *heads[NS2], *ecs[NS2];
		for (int k = 0; k < NS2; ++k) { /* full-duplex G.711 AudioStreams, the default features' filters of this plugin */
			MSFilter *mic = ms2shim_new_source(g_fac), *far = ms2shim_new_source(g_fac), *spk = ms2shim_new_sink(g_fac), *out = ms2shim_new_sink(g_fac);
			MSFilter *dec = ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID), *lmx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID), *plc = ms_factory_create_filter(g_fac, MS_GENERIC_PLC_ID);
			MSFilter *fc = ms_factory_create_filter(g_fac, MS_AUDIO_FLOW_CONTROL_ID), *dtmf = ms2shim_new_pass(g_fac), *vr = ms_factory_create_filter(g_fac, MS_VOLUME_ID);
			MSFilter *ec = ms_factory_create_filter(g_fac, MS_SPEEX_EC_ID), *vs = ms_factory_create_filter(g_fac, MS_VOLUME_ID), *omx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
			MSFilter *enc = ms_factory_create_filter(g_fac, MS_ULAW_ENC_ID);
			MSFilter *fourteen[] = {mic, far, spk, out, dec, lmx, plc, fc, dtmf, vr, ec, vs, omx, enc};
			for (int i = 0; i < 14; ++i) all[nall++] = fourteen[i];
			heads[k] = mic, ecs[k] = ec;
			ms2shim_source_set_loop(mic, ring8, sizeof ring8[0], 4, k), ms2shim_source_set_loop(far, codes, sizeof codes[0], 4, k + 1);
			ms2shim_sink_set_discard(spk, 1), ms2shim_sink_set_discard(out, 1);
			for (MSFilter **f = (MSFilter *[]){lmx, plc, fc, vr, ec, vs, omx, NULL}; *f; ++f) set_int(*f, MS_FILTER_SET_SAMPLE_RATE, 8000);
			set_int(ec, MS_ECHO_CANCELLER_SET_TAIL_LENGTH, 64);
			ms_filter_link(far, 0, dec, 0), ms_filter_link(dec, 0, lmx, 0), ms_filter_link(lmx, 0, plc, 0), ms_filter_link(plc, 0, fc, 0), ms_filter_link(fc, 0, dtmf, 0);
			ms_filter_link(dtmf, 0, vr, 0), ms_filter_link(vr, 0, ec, 0), ms_filter_link(ec, 0, spk, 0);
			ms_filter_link(mic, 0, ec, 1), ms_filter_link(ec, 1, vs, 0), ms_filter_link(vs, 0, omx, 0), ms_filter_link(omx, 0, enc, 0), ms_filter_link(enc, 0, out, 0);
			CHECK(ms_ticker_attach(tk, mic) == 0);
		}
		/* ... and a conference SERVER's bridge of four G.711 members (decoder -> volrecv -> pin -> encoder: server_leg.inl) */
		smx = ms_factory_create_filter(g_fac, MS_AUDIO_MIXER_ID);
		all[nall++] = smx;
		set_int(smx, MS_FILTER_SET_SAMPLE_RATE, 8000), set_int(smx, MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, 1);
		for (int k = 0; k < NM2; ++k) {
			MSFilter *src = ms2shim_new_source(g_fac), *snk = ms2shim_new_sink(g_fac), *dec = ms_factory_create_filter(g_fac, MS_ULAW_DEC_ID);
			MSFilter *vol = ms_factory_create_filter(g_fac, MS_VOLUME_ID), *enc = ms_factory_create_filter(g_fac, MS_ULAW_ENC_ID);
			MSFilter *five[] = {src, snk, dec, vol, enc};
			for (int i = 0; i < 5; ++i) all[nall++] = five[i];
			ms2shim_source_set_loop(src, codes, sizeof codes[0], 4, k);
			ms2shim_sink_set_discard(snk, 1);
			set_int(vol, MS_FILTER_SET_SAMPLE_RATE, 8000);
			ms_filter_link(src, 0, dec, 0), ms_filter_link(dec, 0, vol, 0), ms_filter_link(vol, 0, smx, k), ms_filter_link(smx, k, enc, 0), ms_filter_link(enc, 0, snk, 0);
		}
		CHECK(ms_ticker_attach(tk, smx) == 0);
		replumber_t rp = {tk, {mx[0], heads[0], mx[1], heads[1], mx[2], smx}, 0, 0};
		pthread_t rt;
		for (int t = 0; t < 4; ++t) ms_ticker_step(tk);
		for (int c = 0; c < NC2; ++c) CHECK(p_in_batch(mx[c]) == 1);
		for (int k = 0; k < NS2; ++k) CHECK(p_in_batch(ecs[k]) == 1);
		CHECK(p_in_batch(smx) == 1);
		CHECK(pthread_create(&rt, NULL, replumber, &rp) == 0);
		for (int t = 0; t < 60; ++t) { /* the ticker ticks while the application re-plumbs */
			ms_ticker_step(tk);
			usleep(1500); /* (a re-plumbing takes about a tick's time, as in service: a stream that stays away for many ticks has its PLC make up for
			               * all of them afterwards, msgenericplc.c:117-166, and the canceller's delay line drops -- and counts -- what the far end then runs ahead) */
		}
		__atomic_store_n(&rp.stop, 1, __ATOMIC_SEQ_CST);
		pthread_join(rt, NULL);
		CHECK(rp.done >= 5);
		for (int t = 0; t < 3; ++t) ms_ticker_step(tk);
		for (int c = 0; c < NC2; ++c) CHECK(p_in_batch(mx[c]) == 1); /* every one of them found its way back into its batch */
		for (int k = 0; k < NS2; ++k) CHECK(p_in_batch(ecs[k]) == 1);
		CHECK(p_in_batch(smx) == 1);
		ms_ticker_detach(tk, smx);
		for (int c = 0; c < NC2; ++c) ms_ticker_detach(tk, mx[c]);
		for (int k = 0; k < NS2; ++k) ms_ticker_detach(tk, heads[k]);
		for (int i = 0; i < nall; ++i) { /* (ms_filter_destroy unlinks nothing: take the links down first) */
			MSFilter *f = all[i];
			for (int pin = 0; pin < f->desc->noutputs; ++pin)
				if (f->outputs[pin]) ms_filter_unlink(f, pin, f->outputs[pin]->next.filter, f->outputs[pin]->next.pin);
		}
		for (int i = 0; i < nall; ++i) ms_filter_destroy(all[i]);
		ms_ticker_destroy(tk);
	}
	return NULL;
}

static void *walker(void *arg) {
	long walks = 0;
	(void)arg;
	while (!__atomic_load_n(&g_stop, __ATOMIC_SEQ_CST)) {
		int h = -1, b = -1, s = -1, dev[64];
		p_stats(&h, &b, &s);
		CHECK(h >= 0 && b >= 0 && s >= 0);
		CHECK(p_hub_devices(dev, 64) >= 0);
		++walks;
	}
	return (void *)(intptr_t)walks;
}

int main(int argc, char **argv) {
	pthread_t th[9];
	void *p_fused = NULL;
	void *walks = NULL;
	int h, b, s;
	if (argc < 2) {
		fprintf(stderr, "usage: san_stress <libmsmi355xfilters.so> [rounds]\n");
		return 2;
	}
	if (argc > 2) g_rounds = atoi(argv[2]);
	g_fac = ms_factory_new();
	ms2shim_register_test_filters(g_fac);
	if (ms_factory_load_plugin(g_fac, argv[1]) != 0) {
		fprintf(stderr, "san_stress: cannot load %s\n", argv[1]);
		return 2;
	}
	{
		void *so = dlopen(argv[1], RTLD_NOW | RTLD_NOLOAD);
		if (!so) so = dlopen(argv[1], RTLD_NOW);
		p_stats = (void (*)(int *, int *, int *))dlsym(so, "ms_mi355x_runtime_stats");
		p_flush = (void (*)(void))dlsym(so, "ms_mi355x_flush");
		p_hub_devices = (int (*)(int *, int))dlsym(so, "ms_mi355x_hub_devices");
		p_late = (unsigned long long (*)(void))dlsym(so, "ms_mi355x_late_events");
		p_fused = dlsym(so, "ms_mi355x_fused_stats");
		p_in_batch = (int (*)(MSFilter *))dlsym(so, "ms_mi355x_filter_in_batch");
		if (!p_stats || !p_flush || !p_hub_devices || !p_late || !p_fused || !p_in_batch) {
			fprintf(stderr, "san_stress: plugin entry points missing\n");
			return 2;
		}
	}
	pthread_create(&th[3], NULL, walker, NULL);
	for (int i = 0; i < 3; ++i) pthread_create(&th[i], NULL, caller, (void *)(intptr_t)(i + 1));
	pthread_create(&th[4], NULL, grower, NULL);
	pthread_create(&th[5], NULL, conferences, p_fused);
	pthread_create(&th[6], NULL, server_conferences, p_fused);
	pthread_create(&th[7], NULL, audiostreams, p_fused);
	pthread_create(&th[8], NULL, replumbed_by_the_application, NULL);
	for (int i = 0; i < 3; ++i) pthread_join(th[i], NULL);
	pthread_join(th[4], NULL);
	pthread_join(th[5], NULL);
	pthread_join(th[6], NULL);
	pthread_join(th[7], NULL);
	pthread_join(th[8], NULL);
	__atomic_store_n(&g_stop, 1, __ATOMIC_SEQ_CST);
	pthread_join(th[3], &walks);
	p_flush(); /* nothing is running any more */
	p_stats(&h, &b, &s);
	CHECK(h == 0 && b == 0 && s == 0);
	CHECK(p_late() == 0);
	CHECK((long)(intptr_t)walks > 10);
	if (g_fail) return 1;
	printf("ok %ld walks, %d rounds\n", (long)(intptr_t)walks, g_rounds);
	return 0;
}
