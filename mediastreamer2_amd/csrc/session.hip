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

namespace {
constexpr int SLOTS = 3; // upload | compute | download can each hold a different tick
}

struct mi_session {
	mi_ctx *ctx = nullptr;
	mi_session_config cfg;
	int n = 0, nconf = 0, in_len = 0, len = 0, frame = 0, up_stride = 0;
	mi_resampler *rs = nullptr;
	mi_aec *aec = nullptr;
	mi_volume *vol = nullptr;
	mi_mixer *mix = nullptr;
	mi_fifo *f_mic = nullptr, *f_ref = nullptr, *f_out = nullptr;
	hipStream_t s_up = nullptr, s_down = nullptr;
	int16_t *h_mic[SLOTS] = {}, *h_ref[SLOTS] = {}, *h_out[SLOTS] = {};
	int16_t *d_mic[SLOTS] = {}, *d_ref[SLOTS] = {}, *d_out[SLOTS] = {};
	int16_t *d_up = nullptr, *d_micf = nullptr, *d_reff = nullptr, *d_clean = nullptr, *d_tick = nullptr;
	uint8_t *d_ok = nullptr;
	hipEvent_t ev_up[SLOTS] = {}, ev_done[SLOTS] = {}, ev_down[SLOTS] = {};
	bool used[SLOTS] = {};
	mi_graph *graph[SLOTS] = {};
	long long submitted = 0, collected = 0;
	bool acquired = false;
};

namespace {

int run_tick_kernels(mi_session *s, int slot) { // everything on the context's stream
	int rc;
	if (s->rs) {
		if ((rc = mi_resampler_process(s->rs, s->d_mic[slot], s->in_len, s->in_len, s->d_up, s->up_stride, nullptr)) != MI_OK) return rc;
		if ((rc = mi_fifo_push(s->f_mic, s->d_up, s->len, s->up_stride, nullptr)) != MI_OK) return rc;
	} else {
		if ((rc = mi_fifo_push(s->f_mic, s->d_mic[slot], s->len, s->len, nullptr)) != MI_OK) return rc;
	}
	if ((rc = mi_fifo_push(s->f_ref, s->d_ref[slot], s->len, s->len, nullptr)) != MI_OK) return rc;
	const int rounds = (s->len + s->frame - 1) / s->frame; // frames a tick can complete (480 / 256 -> 2)
	for (int r = 0; r < rounds; ++r) {
		if ((rc = mi_fifo_pop(s->f_mic, s->frame, s->d_micf, s->frame, s->d_ok, nullptr, 0)) != MI_OK) return rc;
		if ((rc = mi_fifo_pop(s->f_ref, s->frame, s->d_reff, s->frame, nullptr, s->d_ok, 1)) != MI_OK) return rc;
		if ((rc = mi_aec_process(s->aec, s->d_micf, s->d_reff, s->d_clean, s->frame, s->d_ok, MI_AEC_POSTFILTER)) != MI_OK) return rc;
		if ((rc = mi_fifo_push_gated(s->f_out, s->d_clean, s->frame, s->frame, s->d_ok)) != MI_OK) return rc;
	}
	if ((rc = mi_fifo_pop(s->f_out, s->len, s->d_tick, s->len, nullptr, nullptr, 1)) != MI_OK) return rc;
	if ((rc = mi_volume_process(s->vol, s->d_tick, s->len, s->len, nullptr)) != MI_OK) return rc;
	return mi_mixer_process(s->mix, s->d_tick, nullptr, 1, s->d_out[slot]);
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
		if (s->ev_up[i]) (void)hipEventDestroy(s->ev_up[i]);
		if (s->ev_done[i]) (void)hipEventDestroy(s->ev_done[i]);
		if (s->ev_down[i]) (void)hipEventDestroy(s->ev_down[i]);
	}
	void *dv[] = {s->d_up, s->d_micf, s->d_reff, s->d_clean, s->d_tick, s->d_ok};
	for (void *p : dv)
		if (p) mi_dev_free(c, p);
	if (s->rs) mi_resampler_destroy(s->rs);
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
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_session *s = new mi_session();
	s->ctx = ctx;
	s->cfg = *cfg;
	s->n = cfg->nstreams;
	s->nconf = cfg->nstreams / cfg->members_per_conference;
	s->in_len = cfg->in_rate / 100;
	s->len = cfg->rate / 100;
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
	const int cap = ((2 * s->len + 2 * s->frame) + 7) & ~7;
	if ((rc = mi_fifo_create(ctx, s->n, cap, &s->f_mic)) != MI_OK || (rc = mi_fifo_create(ctx, s->n, cap, &s->f_ref)) != MI_OK ||
	    (rc = mi_fifo_create(ctx, s->n, cap, &s->f_out)) != MI_OK)
		return fail(rc);
	if (hipStreamCreateWithFlags(&s->s_up, hipStreamNonBlocking) != hipSuccess ||
	    hipStreamCreateWithFlags(&s->s_down, hipStreamNonBlocking) != hipSuccess) {
		mi::set_error("hipStreamCreate failed");
		return fail(MI_ENODEV);
	}
	const size_t n = (size_t)s->n;
	for (int i = 0; i < SLOTS; ++i) {
		s->h_mic[i] = (int16_t *)mi_host_alloc(ctx, n * s->in_len * 2);
		s->h_ref[i] = (int16_t *)mi_host_alloc(ctx, n * s->len * 2);
		s->h_out[i] = (int16_t *)mi_host_alloc(ctx, n * s->len * 2);
		s->d_mic[i] = (int16_t *)mi_dev_alloc(ctx, n * s->in_len * 2);
		s->d_ref[i] = (int16_t *)mi_dev_alloc(ctx, n * s->len * 2);
		s->d_out[i] = (int16_t *)mi_dev_alloc(ctx, n * s->len * 2);
		if (!s->h_mic[i] || !s->h_ref[i] || !s->h_out[i] || !s->d_mic[i] || !s->d_ref[i] || !s->d_out[i]) return fail(MI_ENOMEM);
		if (hipEventCreateWithFlags(&s->ev_up[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_done[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_down[i], hipEventDisableTiming) != hipSuccess) {
			mi::set_error("hipEventCreate failed");
			return fail(MI_ENODEV);
		}
	}
	if (s->rs) s->d_up = (int16_t *)mi_dev_alloc(ctx, n * s->up_stride * 2);
	s->d_micf = (int16_t *)mi_dev_alloc(ctx, n * s->frame * 2);
	s->d_reff = (int16_t *)mi_dev_alloc(ctx, n * s->frame * 2);
	s->d_clean = (int16_t *)mi_dev_alloc(ctx, n * s->frame * 2);
	s->d_tick = (int16_t *)mi_dev_alloc(ctx, n * s->len * 2);
	s->d_ok = (uint8_t *)mi_dev_alloc(ctx, n);
	if ((s->rs && !s->d_up) || !s->d_micf || !s->d_reff || !s->d_clean || !s->d_tick || !s->d_ok) return fail(MI_ENOMEM);
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
	s->acquired = true;
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
	MI_HIP(hipMemcpyAsync(s->d_mic[slot], s->h_mic[slot], n * s->in_len * 2, hipMemcpyHostToDevice, s->s_up));
	MI_HIP(hipMemcpyAsync(s->d_ref[slot], s->h_ref[slot], n * s->len * 2, hipMemcpyHostToDevice, s->s_up));
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
	MI_HIP(hipMemcpyAsync(s->h_out[slot], s->d_out[slot], n * s->len * 2, hipMemcpyDeviceToHost, s->s_down));
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
	return mi_mixer_set_controls(s->mix, h_flags, h_gain); // [nconf][members] == [nstreams]
}

// A call leg leaves and another takes its place: every per-stream state of the chain goes back to its initial value
// (what destroying and re-creating the leg's filters does in the reference), the other streams are not touched.
int mi_session_reset_streams(mi_session *s, int first, int count) {
	MI_CHECK_ARG(s && first >= 0 && count >= 0 && first + count <= s->n);
	if (count == 0) return MI_OK;
	int rc;
	if (s->rs && (rc = mi_resampler_reset(s->rs, first, count)) != MI_OK) return rc;
	if ((rc = mi_aec_reset(s->aec, first, count)) != MI_OK) return rc;
	mi_volume_state st;
	memset(&st, 0, sizeof(st));
	st.gain = st.target_gain = 1; // volume_init msvolume.c:92
	st.ng_gain = 1;               // :112
	std::vector<mi_volume_state> all((size_t)count, st);
	if ((rc = mi_volume_set_state(s->vol, first, count, all.data())) != MI_OK) return rc;
	if ((rc = mi_fifo_reset_range(s->f_mic, first, count)) != MI_OK || (rc = mi_fifo_reset_range(s->f_ref, first, count)) != MI_OK ||
	    (rc = mi_fifo_reset_range(s->f_out, first, count)) != MI_OK)
		return rc;
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
