// session.hip -- the chained hot path for a batch of call legs, fed from host buffers, one 10 ms tick per call:
//   MSResample in_rate->rate -> FIFO (ticks -> canceller frames) -> MSSpeexEC (+post-filter) -> FIFO (frames -> ticks)
//   -> MSVolume (AGC) -> MSAudioMixer (conferences of `members`, conference mode)
// i.e. what BASELINE.json's north_star counts per stream, as ONE object for a media server that does not run the
// reference's per-filter ticker (a batch-aware runtime).  It owns nothing but plumbing: every
// stage is the C ABI object of its filter (mi_resampler, mi_aec, ...), so its output is by construction the output of
// tests/test_gpu_pipeline.py's chain.
//
// Three HIP streams: uploads, kernels (the context's stream), downloads.  A tick's inputs go up while the previous
// tick computes and the one before comes down; events order the three, hipGraphs (one per buffer slot) replay the
// kernel sequence.  Pinned host buffers belong to the session: the caller fills / reads them in place.
#include "common.hpp"

#include <cmath>

namespace {
constexpr int SLOTS = 3; // upload | compute | download can each hold a different tick
}

struct mi_session {
	mi_ctx *ctx = nullptr;
	mi_session_config cfg;
	int n = 0, nconf = 0, in_len = 0, len = 0, frame = 0, up_stride = 0;
	int out_len = 0, down_stride = 0;               // samples per tick leaving, row pitch of the down-sampled mix
	size_t mic_bytes = 0, ref_bytes = 0, out_bytes = 0; // per stream and tick, on the host side
	mi_resampler *rs = nullptr, *rs_out = nullptr;
	mi_plc *plc = nullptr;
	uint8_t *h_ev[SLOTS] = {}, *d_ev[SLOTS] = {};
	int32_t *d_evlen = nullptr;
	int16_t *d_pcm = nullptr, *d_down = nullptr, *d_zero = nullptr;
	int16_t *d_mix[SLOTS] = {}; // the mix at `rate` per slot when it is not what leaves (out_rate / out_codec / loopback)
	mi_aec *aec = nullptr;
	mi_volume *vol = nullptr;
	mi_mixer *mix = nullptr;
	mi_fifo *f_mic = nullptr, *f_ref = nullptr, *f_out = nullptr;
	hipStream_t s_up = nullptr, s_down = nullptr;
	int16_t *h_mic[SLOTS] = {}, *h_ref[SLOTS] = {}, *h_out[SLOTS] = {};
	int16_t *d_mic[SLOTS] = {}, *d_ref[SLOTS] = {}, *d_out[SLOTS] = {};
	int16_t *d_up = nullptr, *d_tick = nullptr;
	bool fold_resampler = true; // the canceller's launch runs the up-sampler too (until it says it cannot)
	bool fuse_mix = true;       // volume + conference mix in one launch (likewise)
	// the canceller's frames of a tick (up to ROUNDS_MAX per leg, back to back in one row): frames in, cleaned frames out,
	// frames each leg had ready (buffers [0] only; the arrays are kept for the reset helpers)
	static constexpr int ROUNDS_MAX = MI_AEC_MAX_TICK_FRAMES;
	int rounds = 0;
	int16_t *d_micf[ROUNDS_MAX] = {}, *d_reff[ROUNDS_MAX] = {}, *d_clean[ROUNDS_MAX] = {};
	uint8_t *d_ok[ROUNDS_MAX] = {};
	hipEvent_t ev_up[SLOTS] = {}, ev_done[SLOTS] = {}, ev_down[SLOTS] = {};
	bool used[SLOTS] = {};
	mi_graph *graph[SLOTS] = {};
	long long submitted = 0, collected = 0;
	bool acquired = false;
	// conference membership / active-speaker election (MSAudioConference, src/voip/audioconference.c)
	std::vector<uint8_t> flags;      // MI_MIX_* per stream as last set (default: every pin linked, active, output on)
	std::vector<uint32_t> joined;    // the conference's member LIST is in joining order (bctbx_list_append, audioconference.c:328): of two
	uint32_t join_seq = 0;           // equally loud members the election takes the one that joined first (:449 compares strictly)
};

namespace {

int run_tick_kernels(mi_session *s, int slot) { // everything on the context's stream
	int rc;
	const mi_session_config &cf = s->cfg;
	const int16_t *mic = s->d_mic[slot];
	if (cf.mic_codec) { // MSAlawDec / MSUlawDec
		if ((rc = mi_g711_decode(s->ctx, cf.mic_codec == MI_SESSION_PCMA ? MI_LAW_PCMA : MI_LAW_PCMU, (const uint8_t *)s->d_mic[slot],
		                         (size_t)s->in_len, s->d_pcm, (size_t)s->in_len, nullptr, s->in_len, (size_t)s->n)) != MI_OK)
			return rc;
		mic = s->d_pcm;
	}
	if (s->plc) { // MSGenericPLC behind the decoder: lost legs are concealed, the others delayed by 5 ms (edits the rows in place)
		int16_t *rows = cf.mic_codec ? s->d_pcm : s->d_mic[slot];
		if ((rc = mi_plc_process(s->plc, rows, (size_t)s->in_len, s->d_evlen, s->d_ev[slot])) != MI_OK) return rc;
	}
	// far end: from the host, or what this leg was sent one tick ago
	const int16_t *ref = cf.ref_loopback ? s->d_mix[(slot + SLOTS - 1) % SLOTS] : s->d_ref[slot];
	// MSResample + MSSpeexEC for the tick, FIFOs included: the microphone block is up-sampled, both blocks are queued, the one
	// or two whole frames (480 / 256) a leg then holds are cancelled + post-filtered, the cleaned frames queued for the mixer
	// side -- ONE launch where the canceller's kernel can run the up-sampler itself (integer ratios: 16k -> 48k, 8k -> 48k,
	// 8k -> 16k), else the resampler's launch and then the canceller's
	rc = MI_ENOTSUP;
	if (s->rs && s->fold_resampler)
		rc = mi_aec_process_fifos_resampled(s->aec, s->rs, mic, s->in_len, s->in_len, s->f_mic, s->f_ref, ref, s->len, nullptr, s->f_out,
		                                    s->rounds, MI_AEC_POSTFILTER, nullptr);
	if (rc == MI_ENOTSUP) {
		s->fold_resampler = false; // decided once: the shapes do not change
		const int16_t *mic_tick = mic; // the microphone block at the processing rate
		int mic_stride = s->len;
		if (s->rs) {
			if ((rc = mi_resampler_process(s->rs, mic, s->in_len, s->in_len, s->d_up, s->up_stride, nullptr)) != MI_OK) return rc;
			mic_tick = s->d_up;
			mic_stride = s->up_stride;
		}
		rc = mi_aec_process_fifos(s->aec, s->f_mic, mic_tick, mic_stride, s->f_ref, ref, s->len, nullptr, s->len, s->f_out, s->rounds,
		                          MI_AEC_POSTFILTER, nullptr);
	}
	if (rc != MI_OK) return rc;
	// MSVolume on the tick the mixer side reads: popped from the canceller's output FIFO inside the kernel where the sizes
	// allow 16-byte groups (every rate in use), else pop + process
	int16_t *mix = s->d_mix[slot] ? s->d_mix[slot] : s->d_out[slot];
	// MSVolume + MSAudioMixer: one launch (every leg's tick popped from the canceller's output FIFO, levelled, the conference
	// mixed from the levelled ticks on the chip) where the sizes allow 16-byte groups (every rate in use); else pop, level, mix
	rc = s->fuse_mix ? mi_mixer_process_volume_fifo(s->mix, s->vol, 0, s->f_out, mix) : MI_ENOTSUP;
	if (rc == MI_ENOTSUP) {
		s->fuse_mix = false;
		if ((s->len & 7) == 0) {
			if ((rc = mi_volume_process_fifo(s->vol, s->f_out, s->d_tick, s->len, s->len)) != MI_OK) return rc;
		} else {
			if ((rc = mi_fifo_pop(s->f_out, s->len, s->d_tick, s->len, nullptr, nullptr, 1)) != MI_OK) return rc;
			if ((rc = mi_volume_process(s->vol, s->d_tick, s->len, s->len, nullptr)) != MI_OK) return rc;
		}
		rc = mi_mixer_process(s->mix, s->d_tick, nullptr, 1, mix);
	}
	if (rc != MI_OK) return rc;
	const int16_t *leaving = mix;
	int pitch = s->len;
	if (s->rs_out) { // MSResample rate -> out_rate
		if ((rc = mi_resampler_process(s->rs_out, mix, s->len, s->len, s->d_down, s->down_stride, nullptr)) != MI_OK) return rc;
		leaving = s->d_down;
		pitch = s->down_stride;
	}
	if (cf.out_codec) // MSAlawEnc / MSUlawEnc
		return mi_g711_encode(s->ctx, cf.out_codec == MI_SESSION_PCMA ? MI_LAW_PCMA : MI_LAW_PCMU, leaving, (size_t)pitch,
		                      (uint8_t *)s->d_out[slot], (size_t)s->out_len, nullptr, s->out_len, (size_t)s->n);
	if (leaving != s->d_out[slot]) // 16-bit output that went through the down-sampler (or a kept mix): pack the rows
		MI_HIP(hipMemcpy2DAsync(s->d_out[slot], (size_t)s->out_len * 2, leaving, (size_t)pitch * 2, (size_t)s->out_len * 2, (size_t)s->n,
		                        hipMemcpyDeviceToDevice, s->ctx->stream));
	return MI_OK;
}

} // namespace

extern "C" {

void mi_session_default_config(mi_session_config *c) {
	if (!c) return;
	memset(c, 0, sizeof(*c));
	c->nstreams = 32;
	c->members_per_conference = 32;
	c->in_rate = 16000;
	c->rate = 48000;
	c->tail_ms = 128;
	c->agc = 1;
	c->stagger = 1;
	c->use_graphs = 0; // per-tick graphs cost more to launch than ~17 kernels do (0.99 vs 0.68 ms at 4096 streams); no gain at 65536
}

void mi_session_destroy(mi_session *s) {
	if (!s) return;
	mi_ctx *c = s->ctx;
	(void)c->activate();
	(void)hipStreamSynchronize(c->stream);
	if (s->s_up) (void)hipStreamSynchronize(s->s_up);
	if (s->s_down) (void)hipStreamSynchronize(s->s_down);
	for (int i = 0; i < SLOTS; ++i) {
		if (s->graph[i]) mi_graph_destroy(s->graph[i]);
		if (s->h_mic[i]) mi_host_free(c, s->h_mic[i]);
		if (s->h_ref[i]) mi_host_free(c, s->h_ref[i]);
		if (s->h_out[i]) mi_host_free(c, s->h_out[i]);
		if (s->d_mic[i]) mi_dev_free(c, s->d_mic[i]);
		if (s->d_ref[i]) mi_dev_free(c, s->d_ref[i]);
		if (s->d_out[i]) mi_dev_free(c, s->d_out[i]);
		if (s->d_mix[i]) mi_dev_free(c, s->d_mix[i]);
		if (s->h_ev[i]) mi_host_free(c, s->h_ev[i]);
		if (s->d_ev[i]) mi_dev_free(c, s->d_ev[i]);
		if (s->ev_up[i]) (void)hipEventDestroy(s->ev_up[i]);
		if (s->ev_done[i]) (void)hipEventDestroy(s->ev_done[i]);
		if (s->ev_down[i]) (void)hipEventDestroy(s->ev_down[i]);
	}
	void *dv[] = {s->d_up, s->d_tick, s->d_pcm, s->d_down, s->d_zero, s->d_evlen};
	for (void *p : dv)
		if (p) mi_dev_free(c, p);
	for (int r = 0; r < mi_session::ROUNDS_MAX; ++r)
		for (void *p : {(void *)s->d_micf[r], (void *)s->d_reff[r], (void *)s->d_clean[r], (void *)s->d_ok[r]})
			if (p) mi_dev_free(c, p);
	if (s->rs) mi_resampler_destroy(s->rs);
	if (s->rs_out) mi_resampler_destroy(s->rs_out);
	if (s->plc) mi_plc_destroy(s->plc);
	if (s->aec) mi_aec_destroy(s->aec);
	if (s->vol) mi_volume_destroy(s->vol);
	if (s->mix) mi_mixer_destroy(s->mix);
	if (s->f_mic) mi_fifo_destroy(s->f_mic);
	if (s->f_ref) mi_fifo_destroy(s->f_ref);
	if (s->f_out) mi_fifo_destroy(s->f_out);
	if (s->s_up) (void)hipStreamDestroy(s->s_up);
	if (s->s_down) (void)hipStreamDestroy(s->s_down);
	delete s;
}

int mi_session_create(mi_ctx *ctx, const mi_session_config *cfg, mi_session **out) {
	MI_CHECK_ARG(ctx && cfg && out);
	*out = nullptr;
	MI_CHECK_ARG(cfg->nstreams > 0 && cfg->members_per_conference > 0 && cfg->nstreams % cfg->members_per_conference == 0);
	MI_CHECK_ARG(cfg->rate > 0 && cfg->in_rate > 0 && cfg->rate % 100 == 0 && cfg->in_rate % 100 == 0 && cfg->tail_ms > 0);
	MI_CHECK_ARG(cfg->mic_codec >= MI_SESSION_PCM16 && cfg->mic_codec <= MI_SESSION_PCMU && cfg->out_codec >= MI_SESSION_PCM16 &&
	             cfg->out_codec <= MI_SESSION_PCMU);
	MI_CHECK_ARG(cfg->out_rate >= 0 && cfg->out_rate % 100 == 0 && cfg->ref_delay_ms >= 0 && cfg->ref_delay_ms <= 1000);
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_session *s = new mi_session();
	s->ctx = ctx;
	s->cfg = *cfg;
	s->n = cfg->nstreams;
	s->nconf = cfg->nstreams / cfg->members_per_conference;
	s->flags.assign((size_t)s->n, (uint8_t)(MI_MIX_LINKED | MI_MIX_ACTIVE | MI_MIX_OUTPUT));
	s->joined.resize((size_t)s->n);
	for (int i = 0; i < s->n; ++i) s->joined[(size_t)i] = ++s->join_seq; // a session is created full: joined in pin order
	s->in_len = cfg->in_rate / 100;
	s->len = cfg->rate / 100;
	const bool down = cfg->out_rate != 0 && cfg->out_rate != cfg->rate;
	s->out_len = (down ? cfg->out_rate : cfg->rate) / 100;
	s->mic_bytes = (size_t)s->in_len * (cfg->mic_codec ? 1 : 2);
	s->ref_bytes = cfg->ref_loopback ? 0 : (size_t)s->len * 2;
	s->out_bytes = (size_t)s->out_len * (cfg->out_codec ? 1 : 2);
	s->frame = mi_aec_framesize(64, cfg->rate); // speexec.c:41,171-180
	int rc = MI_OK;
	auto fail = [&](int code) {
		mi_session_destroy(s);
		return code;
	};
	if (cfg->in_rate != cfg->rate) {
		if ((rc = mi_resampler_create(ctx, s->n, (uint32_t)cfg->in_rate, (uint32_t)cfg->rate, 3, &s->rs)) != MI_OK) return fail(rc);
		s->up_stride = (mi_resampler_out_capacity(s->rs, s->in_len) + 7) & ~7;
	}
	if (down) {
		if ((rc = mi_resampler_create(ctx, s->n, (uint32_t)cfg->rate, (uint32_t)cfg->out_rate, 3, &s->rs_out)) != MI_OK) return fail(rc);
		s->down_stride = (mi_resampler_out_capacity(s->rs_out, s->len) + 7) & ~7;
	}
	if ((rc = mi_aec_create(ctx, s->n, cfg->rate, s->frame, cfg->tail_ms * cfg->rate / 1000, &s->aec)) != MI_OK) return fail(rc);
	if ((rc = mi_volume_create(ctx, s->n, cfg->rate, &s->vol)) != MI_OK) return fail(rc);
	if (cfg->agc) {
		mi_volume_params p;
		mi_volume_default_params(&p);
		p.agc_enabled = 1;
		std::vector<mi_volume_params> all((size_t)s->n, p);
		if ((rc = mi_volume_set_params(s->vol, 0, s->n, all.data())) != MI_OK) return fail(rc);
	}
	if ((rc = mi_mixer_create(ctx, s->nconf, cfg->members_per_conference, s->len, &s->mix)) != MI_OK) return fail(rc);
	// capacities: whole frames (the canceller reads / writes the rings frame-wise, mi_aec_process_fifos), two ticks + two
	// frames, + one frame for the lead of a staggered leg (its output queue stands one frame fuller)
	const int cap = (2 * s->len + (cfg->stagger ? 3 : 2) * s->frame + s->frame - 1) / s->frame * s->frame;
	const int delay = cfg->ref_delay_ms * cfg->rate / 1000;
	const int ref_cap = (cap + delay + s->frame - 1) / s->frame * s->frame;
	if ((rc = mi_fifo_create(ctx, s->n, cap, &s->f_mic)) != MI_OK || (rc = mi_fifo_create(ctx, s->n, ref_cap, &s->f_ref)) != MI_OK ||
	    (rc = mi_fifo_create(ctx, s->n, cap, &s->f_out)) != MI_OK)
		return fail(rc);
	if (hipStreamCreateWithFlags(&s->s_up, hipStreamNonBlocking) != hipSuccess ||
	    hipStreamCreateWithFlags(&s->s_down, hipStreamNonBlocking) != hipSuccess) {
		mi::set_error("hipStreamCreate failed");
		return fail(MI_ENODEV);
	}
	const size_t n = (size_t)s->n;
	for (int i = 0; i < SLOTS; ++i) {
		s->h_mic[i] = (int16_t *)mi_host_alloc(ctx, n * s->mic_bytes);
		s->h_out[i] = (int16_t *)mi_host_alloc(ctx, n * s->out_bytes);
		s->d_mic[i] = (int16_t *)mi_dev_alloc(ctx, n * s->mic_bytes);
		s->d_out[i] = (int16_t *)mi_dev_alloc(ctx, n * s->out_bytes);
		if (!s->h_mic[i] || !s->h_out[i] || !s->d_mic[i] || !s->d_out[i]) return fail(MI_ENOMEM);
		if (!cfg->ref_loopback) {
			s->h_ref[i] = (int16_t *)mi_host_alloc(ctx, n * s->ref_bytes);
			s->d_ref[i] = (int16_t *)mi_dev_alloc(ctx, n * s->ref_bytes);
			if (!s->h_ref[i] || !s->d_ref[i]) return fail(MI_ENOMEM);
		}
		if (cfg->ref_loopback || down || cfg->out_codec) { // the mix at `rate` is kept: next tick's reference and / or the encoder's input
			s->d_mix[i] = (int16_t *)mi_dev_alloc(ctx, n * s->len * 2);
			if (!s->d_mix[i]) return fail(MI_ENOMEM);
			MI_HIP(hipMemsetAsync(s->d_mix[i], 0, n * s->len * 2, ctx->stream));
		}
		if (hipEventCreateWithFlags(&s->ev_up[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_done[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_down[i], hipEventDisableTiming) != hipSuccess) {
			mi::set_error("hipEventCreate failed");
			return fail(MI_ENODEV);
		}
	}
	if (s->rs) s->d_up = (int16_t *)mi_dev_alloc(ctx, n * s->up_stride * 2);
	s->rounds = (s->len + s->frame - 1) / s->frame;
	if (s->rounds > mi_session::ROUNDS_MAX) return fail(MI_ENOTSUP);
	for (int r = 0; r < 1; ++r) {
		s->d_micf[r] = (int16_t *)mi_dev_alloc(ctx, n * s->rounds * s->frame * 2);
		s->d_reff[r] = (int16_t *)mi_dev_alloc(ctx, n * s->rounds * s->frame * 2);
		s->d_clean[r] = (int16_t *)mi_dev_alloc(ctx, n * s->rounds * s->frame * 2);
		s->d_ok[r] = (uint8_t *)mi_dev_alloc(ctx, n);
		if (!s->d_micf[r] || !s->d_reff[r] || !s->d_clean[r] || !s->d_ok[r]) return fail(MI_ENOMEM);
	}
	s->d_tick = (int16_t *)mi_dev_alloc(ctx, n * s->len * 2);
	if ((s->rs && !s->d_up) || !s->d_tick) return fail(MI_ENOMEM);
	if (cfg->mic_codec && !(s->d_pcm = (int16_t *)mi_dev_alloc(ctx, n * s->in_len * 2))) return fail(MI_ENOMEM);
	if (s->rs_out && !(s->d_down = (int16_t *)mi_dev_alloc(ctx, n * s->down_stride * 2))) return fail(MI_ENOMEM);
	if (cfg->plc) {
		if ((rc = mi_plc_create(ctx, s->n, cfg->in_rate, s->in_len, &s->plc)) != MI_OK) return fail(rc);
		std::vector<int32_t> lens(n, s->in_len);
		if (!(s->d_evlen = (int32_t *)mi_dev_alloc(ctx, n * 4))) return fail(MI_ENOMEM);
		MI_HIP(hipMemcpy(s->d_evlen, lens.data(), n * 4, hipMemcpyHostToDevice));
		for (int i = 0; i < SLOTS; ++i) {
			s->h_ev[i] = (uint8_t *)mi_host_alloc(ctx, n);
			s->d_ev[i] = (uint8_t *)mi_dev_alloc(ctx, n);
			if (!s->h_ev[i] || !s->d_ev[i]) return fail(MI_ENOMEM);
		}
	}
	if (delay > 0) { // speexec.c:205-208: delay_ms of silence ahead of the reference
		if (!(s->d_zero = (int16_t *)mi_dev_alloc(ctx, n * (size_t)delay * 2))) return fail(MI_ENOMEM);
		MI_HIP(hipMemsetAsync(s->d_zero, 0, n * (size_t)delay * 2, ctx->stream));
		if ((rc = mi_fifo_push(s->f_ref, s->d_zero, delay, delay, nullptr)) != MI_OK) return fail(rc);
	}
	if (cfg->stagger && (rc = mi_aec_stagger_fifos(s->aec, s->f_mic, s->f_ref, s->len, 0, s->n)) != MI_OK) return fail(rc);
	MI_HIP(hipStreamSynchronize(ctx->stream));
	*out = s;
	return MI_OK;
}

int mi_session_tick_samples(const mi_session *s, int *in_samples, int *out_samples) {
	MI_CHECK_ARG(s != nullptr);
	if (in_samples) *in_samples = s->in_len;
	if (out_samples) *out_samples = s->len;
	return MI_OK;
}

int mi_session_tick_bytes(const mi_session *s, int *mic_bytes, int *ref_bytes, int *out_bytes) {
	MI_CHECK_ARG(s != nullptr);
	if (mic_bytes) *mic_bytes = (int)s->mic_bytes;
	if (ref_bytes) *ref_bytes = (int)s->ref_bytes;
	if (out_bytes) *out_bytes = (int)s->out_bytes;
	return MI_OK;
}

int mi_session_acquire(mi_session *s, int16_t **h_mic, int16_t **h_ref) {
	MI_CHECK_ARG(s && h_mic && h_ref);
	if (s->submitted - s->collected >= SLOTS) {
		mi::set_error("all %d ticks in flight: collect one first", SLOTS);
		return MI_EINVAL;
	}
	const int slot = (int)(s->submitted % SLOTS);
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	// the slot's previous upload must have been consumed by its kernels before the host overwrites the staging
	if (s->used[slot]) MI_HIP(hipEventSynchronize(s->ev_done[slot]));
	*h_mic = s->h_mic[slot];
	*h_ref = s->h_ref[slot];
	if (s->plc) memset(s->h_ev[slot], MI_PLC_RECEIVED, (size_t)s->n);
	s->acquired = true;
	return MI_OK;
}

int mi_session_events(mi_session *s, uint8_t **h_events) {
	MI_CHECK_ARG(s && h_events);
	if (!s->plc || !s->acquired) {
		mi::set_error(s->plc ? "mi_session_events outside acquire .. submit" : "the session was created without plc");
		return MI_EINVAL;
	}
	*h_events = s->h_ev[(int)(s->submitted % SLOTS)];
	return MI_OK;
}

int mi_session_submit(mi_session *s) {
	MI_CHECK_ARG(s != nullptr);
	if (!s->acquired) {
		mi::set_error("mi_session_submit without mi_session_acquire");
		return MI_EINVAL;
	}
	mi_ctx *c = s->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const int slot = (int)(s->submitted % SLOTS);
	const size_t n = (size_t)s->n;
	// upload on its own stream
	MI_HIP(hipMemcpyAsync(s->d_mic[slot], s->h_mic[slot], n * s->mic_bytes, hipMemcpyHostToDevice, s->s_up));
	if (s->ref_bytes) MI_HIP(hipMemcpyAsync(s->d_ref[slot], s->h_ref[slot], n * s->ref_bytes, hipMemcpyHostToDevice, s->s_up));
	if (s->plc) MI_HIP(hipMemcpyAsync(s->d_ev[slot], s->h_ev[slot], n, hipMemcpyHostToDevice, s->s_up));
	MI_HIP(hipEventRecord(s->ev_up[slot], s->s_up));
	// kernels wait for this tick's upload and for the download that last read this slot's output buffer
	MI_HIP(hipStreamWaitEvent(c->stream, s->ev_up[slot], 0));
	if (s->used[slot]) MI_HIP(hipStreamWaitEvent(c->stream, s->ev_down[slot], 0));
	int rc;
	if (s->cfg.use_graphs) {
		if (!s->graph[slot]) {
			// first use of the slot: run eagerly once is not an option (state would advance twice), so capture directly
			if ((rc = mi_ctx_capture_begin(c)) != MI_OK) return rc;
			rc = run_tick_kernels(s, slot);
			mi_graph *g = nullptr;
			const int rc2 = mi_ctx_capture_end(c, &g);
			if (rc != MI_OK) return rc;
			if (rc2 != MI_OK) return rc2;
			s->graph[slot] = g;
		}
		if ((rc = mi_graph_launch(s->graph[slot])) != MI_OK) return rc;
	} else if ((rc = run_tick_kernels(s, slot)) != MI_OK) {
		return rc;
	}
	MI_HIP(hipEventRecord(s->ev_done[slot], c->stream));
	// download on its own stream
	MI_HIP(hipStreamWaitEvent(s->s_down, s->ev_done[slot], 0));
	MI_HIP(hipMemcpyAsync(s->h_out[slot], s->d_out[slot], n * s->out_bytes, hipMemcpyDeviceToHost, s->s_down));
	MI_HIP(hipEventRecord(s->ev_down[slot], s->s_down));
	s->used[slot] = true;
	s->submitted++;
	s->acquired = false;
	return MI_OK;
}

int mi_session_collect(mi_session *s, const int16_t **h_out) {
	MI_CHECK_ARG(s && h_out);
	if (s->collected >= s->submitted) {
		mi::set_error("nothing in flight");
		return MI_EINVAL;
	}
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	const int slot = (int)(s->collected % SLOTS);
	MI_HIP(hipEventSynchronize(s->ev_down[slot]));
	*h_out = s->h_out[slot];
	s->collected++;
	return MI_OK;
}

int mi_session_in_flight(const mi_session *s) { return s ? (int)(s->submitted - s->collected) : 0; }

// ---- conference control plane (what MSAudioConference drives through the mixer's methods, src/voip/audioconference.c:
// mute = MS_AUDIO_MIXER_SET_ACTIVE 0, listen-only = MS_AUDIO_MIXER_ENABLE_OUTPUT ..., per-member input gain) and the level
// meter read-out (MS_VOLUME_GET_LINEAR) an active-speaker detector polls.  Both wait for the ticks already submitted.
int mi_session_set_controls(mi_session *s, const uint8_t *h_flags, const float *h_gain) {
	MI_CHECK_ARG(s && (h_flags || h_gain));
	if (h_flags) s->flags.assign(h_flags, h_flags + s->n);
	return mi_mixer_set_controls(s->mix, h_flags, h_gain); // [nconf][members] == [nstreams]
}

// ms_audio_conference_add_member (src/voip/audioconference.c:322-345): a NEW endpoint joins -- its own filters are fresh
// (resampler history, canceller, meter, FIFOs: mi_session_reset_streams), its mixer pin becomes linked / active / output
// enabled.  The conference graph is detached and re-attached around this (:325-327): the other members' filters keep
// their state (their MSFilter objects survive) and so do their streams here.
int mi_session_add_member(mi_session *s, int stream) {
	MI_CHECK_ARG(s && stream >= 0 && stream < s->n);
	if (s->flags[(size_t)stream] & MI_MIX_LINKED) {
		mi::set_error("mi_session_add_member: stream %d is a member already", stream);
		return MI_EINVAL;
	}
	const int rc = mi_session_reset_streams(s, stream, 1);
	if (rc != MI_OK) return rc;
	s->flags[(size_t)stream] = MI_MIX_LINKED | MI_MIX_ACTIVE | MI_MIX_OUTPUT;
	s->joined[(size_t)stream] = ++s->join_seq; // appended to the member list
	return mi_mixer_set_controls(s->mix, s->flags.data(), nullptr);
}

// ms_audio_conference_remove_member (:366-374): the pin is unplumbed -- it neither contributes nor receives; the row of
// its output is left alone from now on (zeros in a fresh download buffer).  The others carry on.
int mi_session_remove_member(mi_session *s, int stream) {
	MI_CHECK_ARG(s && stream >= 0 && stream < s->n);
	if (!(s->flags[(size_t)stream] & MI_MIX_LINKED)) {
		mi::set_error("mi_session_remove_member: stream %d is no member", stream);
		return MI_EINVAL;
	}
	s->flags[(size_t)stream] = 0;
	const int rc = mi_mixer_set_controls(s->mix, s->flags.data(), nullptr);
	if (rc != MI_OK) return rc;
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	// the mixer leaves an unplumbed pin's row alone: what the departed leg last heard must not linger in the buffers
	MI_HIP(hipStreamSynchronize(s->ctx->stream));
	if (s->s_down) MI_HIP(hipStreamSynchronize(s->s_down));
	for (int i = 0; i < SLOTS; ++i) {
		if (s->d_out[i]) MI_HIP(hipMemsetAsync((uint8_t *)s->d_out[i] + (size_t)stream * s->out_bytes, 0, s->out_bytes, s->ctx->stream));
		if (s->d_mix[i]) MI_HIP(hipMemsetAsync(s->d_mix[i] + (size_t)stream * s->len, 0, (size_t)s->len * 2, s->ctx->stream));
		if (s->h_out[i]) memset((uint8_t *)s->h_out[i] + (size_t)stream * s->out_bytes, 0, s->out_bytes);
	}
	return MI_OK;
}

int mi_session_member_count(const mi_session *s, int conference) {
	if (!s || conference < 0 || conference >= s->nconf) return MI_EINVAL;
	const int mm = s->cfg.members_per_conference;
	int c = 0;
	for (int m = 0; m < mm; ++m) c += (s->flags[(size_t)conference * mm + m] & MI_MIX_LINKED) != 0;
	return c;
}

// ms_audio_conference_process_events' election in mixer mode (:436-452): per conference the unmuted member whose
// MS_VOLUME_GET_MAX -- the maximum of the smoothed energy over a one-second window (msvolume.c:143-148,:402-406), in
// dBm0 -- is the largest and above -30 dB (audioconference.c:31).  now_ms: the caller's clock (the ticker's time).
int mi_session_active_speakers(mi_session *s, uint64_t now_ms, int32_t *h_winner, float *h_max_db) {
	MI_CHECK_ARG(s && h_winner);
	(void)now_ms; // the windows run on the device, one record per tick (msvolume.c:404)
	std::vector<float> mx((size_t)s->n);
	const int rc = mi_volume_get_max(s->vol, 0, s->n, mx.data());
	if (rc != MI_OK) return rc;
	const int mm = s->cfg.members_per_conference;
	for (int c = 0; c < s->nconf; ++c) {
		float best = -120.f; // MS_VOLUME_DB_LOWEST
		int win = -1;
		uint32_t win_joined = 0;
		for (int m = 0; m < mm; ++m) {
			const size_t i = (size_t)c * mm + m;
			const uint8_t f = s->flags[i];
			if (!(f & MI_MIX_LINKED) || !(f & MI_MIX_ACTIVE)) continue; // not plumbed / muted (:445)
			const float lin = mx[i];
			const float db = lin == 0 ? -120.f : 10 * log10f(lin); // ms_volume_linear_to_dbm0 msvolume.c:565-568
			if (db <= -30.0f) continue;
			// the list is walked in joining order and a later member must be strictly louder (:449): of equals, the earliest joiner
			if (db > best || (db == best && win >= 0 && s->joined[i] < win_joined)) best = db, win = (int)i, win_joined = s->joined[i];
		}
		h_winner[c] = win;
		if (h_max_db) h_max_db[c] = best;
	}
	return MI_OK;
}

// A call leg leaves and another takes its place: every per-stream state of the chain goes back to its initial value
// (what destroying and re-creating the leg's filters does in the reference), the other streams are not touched.
int mi_session_reset_streams(mi_session *s, int first, int count) {
	MI_CHECK_ARG(s && first >= 0 && count >= 0 && first + count <= s->n);
	if (count == 0) return MI_OK;
	int rc;
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	if (s->rs && (rc = mi_resampler_reset(s->rs, first, count)) != MI_OK) return rc;
	if ((rc = mi_aec_reset(s->aec, first, count)) != MI_OK) return rc;
	mi_volume_state st;
	memset(&st, 0, sizeof(st));
	st.gain = st.target_gain = 1; // volume_init msvolume.c:92
	st.ng_gain = 1;               // :112
	std::vector<mi_volume_state> all((size_t)count, st);
	if ((rc = mi_volume_set_state(s->vol, first, count, all.data())) != MI_OK) return rc;
	if ((rc = mi_volume_reset_max(s->vol, first, count)) != MI_OK) return rc;
	if ((rc = mi_fifo_reset_range(s->f_mic, first, count)) != MI_OK || (rc = mi_fifo_reset_range(s->f_ref, first, count)) != MI_OK ||
	    (rc = mi_fifo_reset_range(s->f_out, first, count)) != MI_OK)
		return rc;
	if (s->rs_out && (rc = mi_resampler_reset(s->rs_out, first, count)) != MI_OK) return rc;
	if (s->plc && (rc = mi_plc_reset(s->plc, first, count)) != MI_OK) return rc;
	if (s->cfg.ref_loopback) // the new leg was not sent anything yet
		for (int i = 0; i < SLOTS; ++i)
			MI_HIP(hipMemsetAsync(s->d_mix[i] + (size_t)first * s->len, 0, (size_t)count * s->len * 2, s->ctx->stream));
	if (s->d_zero) { // its reference FIFO starts with the configured delay again
		const int delay = s->cfg.ref_delay_ms * s->cfg.rate / 1000;
		std::vector<uint8_t> gate((size_t)s->n, 0);
		for (int i = 0; i < count; ++i) gate[(size_t)(first + i)] = 1;
		MI_HIP(hipMemcpyAsync(s->d_ok[0], gate.data(), (size_t)s->n, hipMemcpyHostToDevice, s->ctx->stream));
		if ((rc = mi_fifo_push_gated(s->f_ref, s->d_zero, delay, delay, s->d_ok[0])) != MI_OK) return rc;
		MI_HIP(hipStreamSynchronize(s->ctx->stream)); // gate is a stack-lifetime buffer
	}
	if (s->cfg.stagger && (rc = mi_aec_stagger_fifos(s->aec, s->f_mic, s->f_ref, s->len, first, count)) != MI_OK) return rc;
	return MI_OK;
}

int mi_session_get_levels(mi_session *s, float *h_linear) {
	MI_CHECK_ARG(s && h_linear);
	std::vector<mi_volume_state> st((size_t)s->n);
	const int rc = mi_volume_get_state(s->vol, 0, s->n, st.data());
	if (rc != MI_OK) return rc;
	for (int i = 0; i < s->n; ++i) h_linear[i] = st[(size_t)i].energy; // volume_get_linear msvolume.c:129-134
	return MI_OK;
}

} // extern "C"
