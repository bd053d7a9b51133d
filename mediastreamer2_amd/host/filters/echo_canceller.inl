// filters/echo_canceller.inl -- MSSpeexEC facade (src/audiofilters/speexec.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ============================================================== echo canceller
// A slot's frames of one tick are staged back to back, kEcTickFrames to a row, and a row is ONE launch of the canceller's
// per-tick kernel (mi_aec_process_frames: both frames of a 48 kHz tick inside one wavefront, the per-stream state crossing
// HBM once, the foreground filter streamed once) -- the while loop of speexec.c:256-305 unrolled into a count per slot.  A
// burst of more frames (a 20 ms packet, a network hiccup) takes the next row: kMaxRounds launches, one wait.
constexpr int kEcTickFrames = MI_AEC_MAX_TICK_FRAMES;
struct EcPool : Pool {
	int rate, F, flen;
	mi_aec *a = nullptr;
	int16_t *h_mic, *h_ref, *h_out, *d_mic, *d_ref, *d_out;
	uint8_t *h_cnt, *d_cnt;
	std::vector<int> staged, ready; // FRAMES per slot
	EcPool(int cap, int r, int frame, int filter_length) : rate(r), F(frame), flen(filter_length) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_aec_create(hub->ctx, capacity, rate, F, flen, &a));
		const size_t c = (size_t)capacity, row = (size_t)kEcTickFrames * F;
		h_mic = pinned<int16_t>(kMaxRounds * c * row);
		h_ref = pinned<int16_t>(kMaxRounds * c * row);
		h_out = pinned<int16_t>(kMaxRounds * c * row);
		h_cnt = pinned<uint8_t>(kMaxRounds * c);
		d_mic = devmem<int16_t>(c * row);
		d_ref = devmem<int16_t>(c * row);
		d_out = devmem<int16_t>(kMaxRounds * c * row); // a row of results per round: the rounds' downloads need not wait for each other
		d_cnt = devmem<uint8_t>(kMaxRounds * c);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	~EcPool() override {
		if (a) mi_aec_destroy(a);
	}
	static constexpr int max_frames() { return kMaxRounds * kEcTickFrames; }
	// where frame k of a slot is staged (and where its result comes back)
	size_t frame_at(size_t slot, int k) const { return (((size_t)(k / kEcTickFrames) * (size_t)capacity + slot) * kEcTickFrames + (size_t)(k % kEcTickFrames)) * (size_t)F; }
	bool enqueue() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity, u = (size_t)hi, row = (size_t)kEcTickFrames * F; // rows [0, hi) are all that was ever handed out
		int maxf = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxf = std::max(maxf, staged[(size_t)s]);
		const int rounds = (maxf + kEcTickFrames - 1) / kEcTickFrames;
		for (int r = 0; r < rounds; ++r) {
			for (int s = 0; s < capacity; ++s)
				h_cnt[r * c + s] = s < hi && !parked(s) ? (uint8_t)std::clamp(staged[(size_t)s] - r * kEcTickFrames, 0, kEcTickFrames) : 0;
			if (zero_copy_rows()) { // the launch reads the pinned rows and writes the results where they lie: no copy at all (leg_chain.inl says why)
				MI_MUST(mi_aec_process_frames(a, h_mic + r * c * row, h_ref + r * c * row, h_out + r * c * row, (int)row, h_cnt + r * c, kEcTickFrames, MI_AEC_POSTFILTER));
				continue;
			}
			MI_MUST(mi_copy_h2d_pinned(ctx, d_mic, h_mic + r * c * row, u * row * 2));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_ref, h_ref + r * c * row, u * row * 2));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_cnt + r * c, h_cnt + r * c, c));
			MI_MUST(mi_aec_process_frames(a, d_mic, d_ref, d_out + r * c * row, (int)row, d_cnt + r * c, kEcTickFrames, MI_AEC_POSTFILTER));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_out + r * c * row, d_out + r * c * row, u * row * 2));
		}
		return rounds > 0;
	}
	void finish() override {
		if (failed) { // the launch did not happen: the microphone frames leave uncancelled (what bypass mode does, speexec.c:229-237)
			for (int s = 0; s < hi; ++s)
				for (int k = 0; k < staged[(size_t)s] && !parked(s); ++k) memcpy(h_out + frame_at((size_t)s, k), h_mic + frame_at((size_t)s, k), (size_t)F * 2);
		}
		for (int s = 0; s < hi; ++s) {
			if (parked(s)) continue;
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void flushed() override; // (a bypass switch waiting for the last walk's frames goes live: below SpeexECState)
	void emit(MSFilter *f, int slot) override {
		const size_t sl = (size_t)slot;
		for (int k = 0; k < ready[sl]; ++k) { // cleaned frames -> outputs[1] (speexec.c:303)
			mblk_t *oecho = allocb((size_t)F * 2, 0);
			memcpy(oecho->b_wptr, h_out + frame_at(sl, k), (size_t)F * 2);
			oecho->b_wptr += F * 2;
			if (f->outputs[1]) ms_queue_put(f->outputs[1], oecho);
			else freemsg(oecho);
		}
		ready[sl] = 0;
	}
};

// MSFlowControlledBufferizer, src/base/msqueue.c:127-256 (SendEvent drop method)
struct FlowBuf {
	MSBufferizer base;
	MSFilter *filter;
	uint64_t flow_control_time;
	uint32_t interval_ms, max_size_ms, granularity_ms, min_size_ms_during_interval;
	int samplerate, nchannels;
	bool immediate_drop; // MSFlowControlledBufferizerImmediateDrop instead of SendEvent (msqueue.c:213-218)
};
void flowbuf_init(FlowBuf *o, MSFilter *f, int rate) {
	ms_bufferizer_init(&o->base);
	o->filter = f;
	o->interval_ms = 5000;
	o->max_size_ms = 100;
	o->granularity_ms = 0;
	o->flow_control_time = 0;
	o->min_size_ms_during_interval = UINT32_MAX;
	o->samplerate = rate;
	o->nchannels = 1;
	o->immediate_drop = false;
}
void flowbuf_put(FlowBuf *o, mblk_t *m, MSQueue *q = nullptr) { // msqueue.c:193-256 (m, or everything queued on q)
	const uint32_t accumulated_ms = (uint32_t)((o->base.size * 1000) / (size_t)o->samplerate / 2) / (uint32_t)o->nchannels;
	if (accumulated_ms < o->min_size_ms_during_interval) o->min_size_ms_during_interval = accumulated_ms;
	if (q) ms_bufferizer_put_from_queue(&o->base, q);
	else ms_bufferizer_put(&o->base, m);
	const uint64_t now = ticker_now(o->filter->ticker);
	const uint32_t since = (uint32_t)(now - o->flow_control_time);
	if (o->flow_control_time == 0) o->flow_control_time = now;
	if (since >= o->interval_ms) {
		uint32_t diff_ms = 0;
		bool trig = false;
		if (o->min_size_ms_during_interval != UINT32_MAX && o->min_size_ms_during_interval > o->max_size_ms) {
			diff_ms = o->min_size_ms_during_interval - o->max_size_ms;
			trig = true;
		} else if (accumulated_ms > o->max_size_ms * 4) {
			diff_ms = (accumulated_ms - o->max_size_ms) / 2;
			trig = true;
		}
		if (trig && diff_ms > o->granularity_ms / 2) {
			MSAudioFlowControlDropEvent ev;
			ev.flow_control_interval_ms = o->interval_ms;
			ev.drop_ms = diff_ms - o->granularity_ms / 2;
			if (ev.drop_ms > 0) {
				if (o->immediate_drop) ms_bufferizer_skip_bytes(&o->base, (int)((ev.drop_ms * 2 * (uint32_t)o->nchannels * (uint32_t)o->samplerate) / 1000));
				else ms_filter_notify(o->filter, MS_AUDIO_FLOW_CONTROL_DROP_EVENT, &ev);
			}
		}
		o->flow_control_time = now;
		o->min_size_ms_during_interval = UINT32_MAX;
	}
}

struct SpeexECState { // speexec.c:49-72
	MSBufferizer delayed_ref;
	FlowBuf ref;
	MSBufferizer echo;
	int framesize, framesize_at_8000, filterlength, samplerate, delay_ms, tail_length_ms, nominal_ref_samples;
	char *state_str;
	bool_t echostarted, bypass_mode, using_zeroes;
	// bypass_mode is what MS_ECHO_CANCELLER_SET_BYPASS_MODE set (and GET returns); process() goes by bypass_live, which follows it when
	// the blocks of the walk BEFORE the call are through (Pool::work_waiting / flushed): the reference's process() of that
	// walk ran before the call
	bool_t bypass_live;
	bool_t unsupported; // the attached rate needs a frame size the kernels do not have: both pins pass (internal, not the user's flag)
	EcPool *pool;
	int slot;
	FusedLeg *leg; // the filter is part of a fused call leg (filters/leg_chain.inl): its canceller and queues live in that bank
	// preprocess sizes the canceller (speexec.c:188-216) but opens no bank slot: a filter that joins a fused leg at the attach never
	// needs one of its own (a bank opened for it alone would be destroyed again with its pinned and device memory, per leg) -- process()
	// takes the slot at the first block of a filter that did not fuse (ec_acquire)
	bool configured, acquire_failed;
	bool fuse_checked; // as the HEAD of a leg (no MSResample of ours in front): looked for a chain to fuse with since the last attach
};

void EcPool::flushed() {
	for (int s = 0; s < hi; ++s) {
		if (parked(s) || !owner[(size_t)s]) continue;
		SpeexECState *st = (SpeexECState *)owner[(size_t)s]->data;
		if (st->bypass_live != st->bypass_mode) __atomic_store_n(&st->bypass_live, st->bypass_mode, __ATOMIC_RELAXED);
	}
}

void ec_init(MSFilter *f) { // speexec.c:74-109
	SpeexECState *s = (SpeexECState *)ms_malloc0(sizeof(*s));
	s->samplerate = 8000;
	ms_bufferizer_init(&s->delayed_ref);
	ms_bufferizer_init(&s->echo);
	flowbuf_init(&s->ref, f, s->samplerate);
	s->tail_length_ms = 250;
	s->framesize_at_8000 = 64;
	s->slot = -1;
	f->data = s;
}
void ec_uninit(MSFilter *f) {
	SpeexECState *s = (SpeexECState *)f->data;
	if (s->leg) leg_release(s->leg, false);
	if (s->state_str) ms_free(s->state_str);
	ms_bufferizer_uninit(&s->delayed_ref);
	ms_free(s);
}
void ec_configure_flow(SpeexECState *s) { // speexec.c:182-186
	s->ref.samplerate = s->samplerate;
	s->ref.max_size_ms = (uint32_t)s->delay_ms;
	s->ref.granularity_ms = (uint32_t)((s->framesize * 1000) / s->samplerate);
}
// ---- the canceller's state as a string: fetch_config / apply_config, speexec.c:119-167 (there a SpeexEchoStateBlob of the
// speex fork through bctbx_base64_*; here the blob of mi_aec_export_state through a local RFC 4648 codec)
const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
char *b64_encode(const uint8_t *p, size_t n) {
	char *out = (char *)ms_malloc0(4 * ((n + 2) / 3) + 1), *o = out;
	for (size_t i = 0; i < n; i += 3) {
		const uint32_t v = ((uint32_t)p[i] << 16) | ((i + 1 < n ? (uint32_t)p[i + 1] : 0u) << 8) | (i + 2 < n ? (uint32_t)p[i + 2] : 0u);
		*o++ = kB64[(v >> 18) & 63];
		*o++ = kB64[(v >> 12) & 63];
		*o++ = i + 1 < n ? kB64[(v >> 6) & 63] : '=';
		*o++ = i + 2 < n ? kB64[v & 63] : '=';
	}
	*o = 0;
	return out;
}
bool b64_decode(const char *txt, std::vector<uint8_t> &out) {
	int8_t rev[256];
	memset(rev, -1, sizeof(rev));
	for (int i = 0; i < 64; ++i) rev[(uint8_t)kB64[i]] = (int8_t)i;
	out.clear();
	uint32_t acc = 0;
	int bits = 0;
	for (const char *c = txt; *c && *c != '='; ++c) {
		if (*c == '\n' || *c == '\r' || *c == ' ') continue;
		const int v = rev[(uint8_t)*c];
		if (v < 0) return false;
		acc = (acc << 6) | (uint32_t)v;
		bits += 6;
		if (bits >= 8) {
			bits -= 8;
			out.push_back((uint8_t)(acc >> bits));
		}
	}
	return true;
}
struct SpeexECState;
void ec_apply_config(SpeexECState *s);
void ec_fetch_config(SpeexECState *s);

void ec_prepare(MSFilter *f);
void ec_preprocess(MSFilter *f) { // speexec.c:188-216
	ec_prepare(f); // (the filter's own fields: it is being attached by this thread, nobody runs it yet)
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f);
}
void ec_prepare(MSFilter *f) { // (the hub locked by the caller where the filter is running: a conference that leaves its batch)
	SpeexECState *s = (SpeexECState *)f->data;
	s->echostarted = FALSE;
	s->fuse_checked = false;
	__atomic_store_n(&s->bypass_live, s->bypass_mode, __ATOMIC_RELAXED);
	s->filterlength = (s->tail_length_ms * s->samplerate) / 1000;
	s->framesize = mi_aec_framesize(s->framesize_at_8000, s->samplerate);
	if (s->framesize != 64 && s->framesize != 128 && s->framesize != 256) {
		// e.g. 96 kHz would need 512-sample frames: audio keeps flowing uncancelled rather than the process dying
		ms_error("mi355x echo canceller: frame size %d (rate %d) is not built; the filter forwards both pins untouched",
		         s->framesize, s->samplerate);
		s->unsupported = TRUE; // internal: the user's MS_ECHO_CANCELLER_SET_BYPASS_MODE value stays what it was
		return;
	}
	s->unsupported = FALSE;
	if (s->filterlength > 64 * s->framesize) { // the kernels hold at most 64 filter blocks (341 ms at 48 kHz, 512 ms at 8/16 kHz)
		ms_warning("mi355x echo canceller: tail of %d ms shortened to %d ms (64 blocks of %d samples)", s->tail_length_ms,
		           64 * s->framesize * 1000 / s->samplerate, s->framesize);
		s->filterlength = 64 * s->framesize;
	}
	const int delay_samples = s->delay_ms * s->samplerate / 1000;
	ms_message("Initializing mi355x echo canceler with framesize=%i, filterlength=%i, delay_samples=%i", s->framesize,
	           s->filterlength, delay_samples);
	ms_bufferizer_flush(&s->delayed_ref);
	mblk_t *m = allocb((size_t)delay_samples * 2, 0); // zeroes for the time of the delay
	memset(m->b_wptr, 0, (size_t)delay_samples * 2);
	m->b_wptr += delay_samples * 2;
	ms_bufferizer_put(&s->delayed_ref, m);
	s->nominal_ref_samples = delay_samples;
	s->configured = true;
	s->acquire_failed = false;
}
// a bank slot of its own for a canceller that did not join a fused leg (hub locked by the caller)
void ec_acquire(MSFilter *f) {
	SpeexECState *s = (SpeexECState *)f->data;
	if (s->pool || !s->configured || s->unsupported) return;
	{
		const int rate = s->samplerate, F = s->framesize, flen = s->filterlength;
		s->pool = bank<EcPool>("ec:" + std::to_string(rate) + ":" + std::to_string(F) + ":" + std::to_string(flen), 1,
		                       [&](int cap) { return new EcPool(cap, rate, F, flen); });
	}
	s->slot = s->pool ? s->pool->acquire(f) : -1;
	if (s->slot < 0) {
		s->pool = nullptr; // no canceller to be had: process() forwards both pins, like bypass mode
		s->acquire_failed = true;
		return;
	}
	note_slot(f);
	ec_apply_config(s); // :209-211
}
mi_aec *leg_canceller(FusedLeg *leg, int *slot); // leg_chain.inl
void ec_apply_config(SpeexECState *s) { // :121-143
	if (s->state_str == NULL || s->pool == nullptr) return;
	std::vector<uint8_t> blob;
	if (!b64_decode(s->state_str, blob)) {
		ms_error("Could not decode base64 %.32s...", s->state_str);
		return;
	}
	if (mi_aec_import_state(s->pool->a, s->slot, blob.data(), blob.size()) != MI_OK) {
		ms_error("Could not apply mi355x echo blob: %s", mi_last_error()); // e.g. saved at another rate or tail length
		return;
	}
	ms_message("mi355x echo state restored.");
}
void ec_fetch_config(SpeexECState *s) { // :145-167
	int slot = s->slot;
	mi_aec *a = s->leg ? leg_canceller(s->leg, &slot) : (s->pool ? s->pool->a : nullptr);
	if (a == nullptr) return;
	std::vector<uint8_t> blob(mi_aec_blob_bytes(a));
	if (mi_aec_export_state(a, slot, blob.data(), blob.size()) != MI_OK) {
		ms_error("Could not retrieve mi355x echo blob: %s", mi_last_error());
		return;
	}
	if (s->state_str) ms_free(s->state_str);
	s->state_str = b64_encode(blob.data(), blob.size());
}
void ec_postprocess(MSFilter *f) { // speexec.c:307-321: state destroyed at detach
	SpeexECState *s = (SpeexECState *)f->data;
	facade_detached(f);
	if (s->leg) leg_release(s->leg, false);
	s->fuse_checked = false;
	HubLock lk(f);
	ms_bufferizer_flush(&s->delayed_ref);
	ms_bufferizer_flush(&s->echo);
	ms_bufferizer_flush(&s->ref.base);
	s->configured = false;
	if (s->pool) {
		EcPool *p = s->pool;
		p->staged[(size_t)s->slot] = p->ready[(size_t)s->slot] = 0;
		if (p->in_use > 1 && !p->failed && mi_aec_reset(p->a, s->slot, 1) != MI_OK) p->failed = mi_failed("mi_aec_reset");
		p->release(s->slot); // the last release of a bank destroys it
	}
	s->pool = nullptr;
	s->slot = -1;
}

// inputs[0] far-end reference, inputs[1] mic; outputs[0] reference copy, outputs[1] cleaned mic (speexec.c:218-222)
// ---- the framing of speexec.c:223-305, in three steps ---------------------------------------------------------------
// (1) far-end blocks: kept twice once the microphone has started -- in `delayed_ref` (what the canceller will be fed,
//     behind the configured delay) and in the flow-controlled `ref` (what goes on to the speaker, frame by frame).
void ec_take_far_end(MSFilter *f, SpeexECState *s) {
	if (!f->inputs[0]) return;
	if (!s->echostarted) {
		ms_warning("Getting reference signal but no echo to synchronize on.");
		ms_queue_flush(f->inputs[0]);
		return;
	}
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) {
		ms_bufferizer_put(&s->delayed_ref, dupmsg(m));
		flowbuf_put(&s->ref, m);
	}
}

mblk_t *ec_block(size_t nbytes) {
	mblk_t *m = allocb(nbytes, 0);
	memset(m->b_wptr, 0, nbytes);
	m->b_wptr += nbytes;
	return m;
}

// (2) one speaker frame per microphone frame: from `ref` when the delay line holds more than the nominal delay plus a
//     frame, otherwise a frame of silence that is ALSO appended to the delay line (the canceller then sees zeros too).
void ec_emit_speaker_frame(MSFilter *f, SpeexECState *s, size_t nbytes) {
	const size_t needed = (size_t)s->nominal_ref_samples * 2 + nbytes;
	if (ms_bufferizer_get_avail(&s->delayed_ref) < needed) {
		mblk_t *silence = ec_block(nbytes);
		ms_bufferizer_put(&s->delayed_ref, silence);
		ms_queue_put(f->outputs[0], dupmsg(silence));
		if (!s->using_zeroes) ms_warning("Not enough ref samples, using zeroes");
		s->using_zeroes = TRUE;
		return;
	}
	if (s->using_zeroes) ms_message("Samples are back.");
	s->using_zeroes = FALSE;
	mblk_t *m = ec_block(nbytes);
	if (ms_bufferizer_read(&s->ref.base, m->b_rptr, nbytes) == 0) { // the reference treats this as fatal (speexec.c:281-283); a frame of
		ms_error("mi355x echo canceller: the far-end bufferizer ran dry; silence sent to the speaker"); // silence is kinder to a server
		memset(m->b_rptr, 0, nbytes);
	}
	ms_queue_put(f->outputs[0], m);
}

// (3) every complete microphone frame is staged with its reference frame; the batch cancels them at the next flush
void ec_process(MSFilter *f) {
	SpeexECState *s = (SpeexECState *)f->data;
	if (!s->leg && !s->fuse_checked && s->configured && f->ticker && !__atomic_load_n(&s->bypass_live, __ATOMIC_RELAXED) && !s->unsupported && f->inputs[1] && !ms_queue_empty(f->inputs[1])) {
		// the first microphone block since the attach, and no MSResample of ours in front (behind one, the resampler is the leg's
		// head and has looked already): is this the head of  MSSpeexEC -> MSVolume (AGC) -> [conference mixer | anything else] ?
		s->fuse_checked = true;
		MSFilter *prev = f->inputs[1]->prev.filter;
		if (prev && prev->desc != &ms_mi355x_resample_desc) {
			HubLock lk(f, s->pool);
			if (MSFilter *mx = leg_find_mixer_ec(f)) conf_try_fuse(mx);
			else leg_try_fuse_plain_ec(f);
		}
	}
	if (!s->leg && !s->pool && s->configured && !s->unsupported && !s->acquire_failed && !__atomic_load_n(&s->bypass_live, __ATOMIC_RELAXED)) {
		HubLock lk(f);
		if (!s->leg) ec_acquire(f); // (not part of a fused leg: a bank slot of its own from here on)
	}
	if (s->leg && leg_wants_out(s->leg)) leg_release(s->leg, true); // (whichever facade of the leg is walked first -- the MSResample in front, when it has a block in this walk)
	if (s->leg) { // fused leg: the microphone is staged (by the leg's MSResample, or here) for the device, the far end for the leg's delay line
		HubLock lk(f, leg_pool(s->leg));
		leg_take_far_end(f, s);
		if (!leg_has_resampler(s->leg)) {
			leg_stage_mic_ec(f, s);
			leg_head_done(s->leg);
		}
		return;
	}
	if (__atomic_load_n(&s->bypass_live, __ATOMIC_RELAXED) || s->unsupported || !s->pool) { // both pins straight through (no canceller to be had: the same)
		for (int pin = 0; pin < 2; ++pin)
			for (mblk_t *m; (m = ms_queue_get(f->inputs[pin])) != NULL;) ms_queue_put(f->outputs[pin], m);
		return;
	}
	HubLock lk(f, s->pool);
	EcPool *p = s->pool;
	const size_t nbytes = (size_t)s->framesize * 2, slot = (size_t)s->slot;
	ec_take_far_end(f, s);
	ms_bufferizer_put_from_queue(&s->echo, f->inputs[1]);
	while (ms_bufferizer_get_avail(&s->echo) >= nbytes) {
		if (p->staged[slot] >= EcPool::max_frames()) { // a burst of more frames than the launch rounds hold: what is staged goes out now
			p->flush();
			p->emit_all();
		}
		const size_t at = p->frame_at(slot, p->staged[slot]);
		ms_bufferizer_read(&s->echo, (uint8_t *)(p->h_mic + at), nbytes);
		s->echostarted = TRUE;
		ec_emit_speaker_frame(f, s, nbytes);
		if (ms_bufferizer_read(&s->delayed_ref, (uint8_t *)(p->h_ref + at), nbytes) == 0) {
			ms_error("mi355x echo canceller: the delayed reference ran dry (speexec.c:291-294 calls this impossible); silence used");
			memset(p->h_ref + at, 0, nbytes);
		}
		p->staged[slot]++;
	}
	if (p->staged[slot]) request_flush(f);
}

int ec_set_sr(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->samplerate = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_framesize(MSFilter *f, void *arg) {
	((SpeexECState *)f->data)->framesize_at_8000 = *(int *)arg;
	return 0;
}
int ec_set_delay(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->delay_ms = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_tail_length(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->tail_length_ms = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_bypass_mode(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	HubLock lk(f);
	__atomic_store_n(&s->bypass_mode, *(bool_t *)arg, __ATOMIC_RELAXED); // (read by process() on the ticker thread, as speexec.c:229 does; s->leg only under the hub's lock)
	if (s->bypass_mode) leg_disqualify(s->leg); // a fused conference goes back to its facades (which then forward both pins)
	// (a fused leg's batch has the last walk's blocks behind it when its head takes it out: ec_prepare makes the flag live there)
	if (!(s->pool && f->ticker && s->pool->work_waiting())) __atomic_store_n(&s->bypass_live, s->bypass_mode, __ATOMIC_RELAXED);
	return 0;
}
int ec_get_bypass_mode(MSFilter *f, void *arg) {
	*(bool_t *)arg = ((SpeexECState *)f->data)->bypass_mode;
	return 0;
}
int ec_set_state(MSFilter *f, void *arg) { // :361-365 (the previous string leaks there; freed here)
	SpeexECState *s = (SpeexECState *)f->data;
	const size_t n = strlen((const char *)arg) + 1;
	if (s->state_str) ms_free(s->state_str);
	s->state_str = (char *)ms_malloc0(n);
	memcpy(s->state_str, arg, n);
	// base64("MIEC..") starts with "TUlFQ": anything else was saved by another canceller (the reference's MSSpeexEC stores its
	// speex fork's SPEEX_ECHO_GET_BLOB, a format of its own) and will be refused when the filter is next prepared
	if (n > 1 && strncmp(s->state_str, "TUlFQ", 5) != 0)
		ms_warning("MSSpeexEC (mi355x): the state string was not saved by this filter (format 'MIEC' v%u expected): it will be ignored "
		           "and the canceller will converge from scratch",
		           (unsigned)MI_AEC_BLOB_VERSION);
	return 0;
}
int ec_get_state(MSFilter *f, void *arg) { // :367-374: the CURRENT state while attached, the stored string otherwise
	SpeexECState *s = (SpeexECState *)f->data;
	{
		HubLock lk(f);
		ec_fetch_config(s);
	}
	*(char **)arg = s->state_str;
	return 0;
}
int ec_get_delay(MSFilter *f, void *arg) {
	*(int *)arg = ((SpeexECState *)f->data)->delay_ms;
	return 0;
}
MSFilterMethod ec_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, ec_set_sr},
                               {MS_ECHO_CANCELLER_SET_TAIL_LENGTH, ec_set_tail_length},
                               {MS_ECHO_CANCELLER_SET_DELAY, ec_set_delay},
                               {MS_ECHO_CANCELLER_SET_FRAMESIZE, ec_set_framesize},
                               {MS_ECHO_CANCELLER_SET_BYPASS_MODE, ec_set_bypass_mode},
                               {MS_ECHO_CANCELLER_GET_BYPASS_MODE, ec_get_bypass_mode},
                               {MS_ECHO_CANCELLER_GET_STATE_STRING, ec_get_state},
                               {MS_ECHO_CANCELLER_SET_STATE_STRING, ec_set_state},
                               {MS_ECHO_CANCELLER_GET_DELAY, ec_get_delay},
                               {0, NULL}};
