// filters/mixer.inl -- MSAudioMixer facade (src/audiofilters/audiomixer.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ======================================================================= mixer
constexpr int MIXER_MAX_CHANNELS = MI_MIXER_MAX_CHANNELS; // audiomixer.c:29
Pool *leg_pool_of(LegBank *b);                            // leg_chain.inl
void leg_conf_walked(LegBank *b, int c);
double leg_trace_ms(LegBank *b);
uint64_t leg_trace_now();
constexpr uint64_t BYPASS_MODE_TIMEOUT = 1000;            // audiomixer.c:31

struct MixerPool : Pool {
	int ns; // samples per tick (all channels interleaved)
	mi_mixer *m = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	uint8_t *h_has, *d_has, *h_run, *d_run, *h_mode, *d_mode;
	std::vector<uint8_t> flags;
	std::vector<float> gain;
	bool ctl_dirty = true;
	// A method between two ticks (MS_AUDIO_MIXER_SET_ACTIVE, SET_INPUT_GAIN, ENABLE_OUTPUT) meets the NEXT walk's blocks in the
	// reference.  Here the last walk's blocks are still on their way to this bank (they are mixed by the coming flush, in one of
	// its rounds -- and a tick the mixer staged in that walk itself in another), so what a method sets waits in next_* and goes
	// live when that flush is through (flushed()): the walk the call preceded is the first it is heard in
	std::vector<uint8_t> next_flags, next_set;
	std::vector<float> next_gain;
	bool next_any = false;
	std::vector<uint8_t> staged, ready;
	MixerPool(int cap, int nsamples) : ns(nsamples) {
		Building b(this, std::max(1, cap / 8)); // a mixer slot carries 50 channel rows
		if (!failed) MI_MUST(mi_mixer_create(hub->ctx, capacity, MIXER_MAX_CHANNELS, ns, &m));
		const size_t c = (size_t)capacity, n = c * MIXER_MAX_CHANNELS;
		h_in = pinned<int16_t>(n * ns);
		h_out = pinned<int16_t>(n * ns);
		d_in = devmem<int16_t>(n * ns);
		d_out = devmem<int16_t>(n * ns);
		h_has = pinned<uint8_t>(n);
		d_has = devmem<uint8_t>(n);
		h_run = pinned<uint8_t>(c);
		d_run = devmem<uint8_t>(c);
		h_mode = pinned<uint8_t>(c);
		d_mode = devmem<uint8_t>(c);
		flags.assign(n, 0);
		gain.assign(n, 1.0f);
		next_flags.assign(n, 0);
		next_gain.assign(n, 1.0f);
		next_set.assign(c, 0);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	~MixerPool() override {
		if (m) mi_mixer_destroy(m);
	}
	bool enqueue() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity;
		bool any = false;
		for (size_t s = 0; s < c; ++s) {
			h_run[s] = (int)s < hi && !parked((int)s) ? staged[s] : 0;
			any |= h_run[s] != 0;
		}
		if (ctl_dirty) {
			MI_MUST(mi_mixer_set_controls(m, flags.data(), gain.data()));
			ctl_dirty = false;
		}
		if (any) {
			const size_t un = (size_t)hi * MIXER_MAX_CHANNELS; // channel rows of the slots ever handed out
			MI_MUST(mi_copy_h2d_pinned(ctx, d_in, h_in, un * ns * 2));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_has, h_has, un));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_run, h_run, c));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_mode, h_mode, c));
			MI_MUST(mi_mixer_process_masked(m, d_in, d_has, 1, d_mode, d_out, d_run));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_out, d_out, un * ns * 2));
		}
		return any;
	}
	void finish() override {
		for (size_t s = 0; s < (size_t)capacity; ++s) {
			if ((int)s < hi && parked((int)s)) continue;
			ready[s] = staged[s];
			staged[s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void emit(MSFilter *f, int slot) override;
	void flushed() override { // the whole flush is through (every round of its chain: the last walk's blocks have all been mixed)
		if (!next_any) return;
		bool left = false;
		for (int s = 0; s < hi; ++s) {
			if (!next_set[(size_t)s]) continue;
			if (parked(s)) { // (a detaching graph's flush is not this conference's)
				left = true;
				continue;
			}
			const size_t at = (size_t)s * MIXER_MAX_CHANNELS;
			std::copy(next_flags.begin() + at, next_flags.begin() + at + MIXER_MAX_CHANNELS, flags.begin() + at);
			std::copy(next_gain.begin() + at, next_gain.begin() + at + MIXER_MAX_CHANNELS, gain.begin() + at);
			next_set[(size_t)s] = 0;
			ctl_dirty = true;
		}
		next_any = left;
	}
};

struct Channel { // audiomixer.c:53-63
	MSBufferizer bufferizer;
	float gain;
	int min_fullness;
	uint64_t last_flow_control, last_activity;
	bool_t active, output_enabled;
};
struct MixerState { // audiomixer.c:132-143
	int nchannels, rate, bytespertick;
	Channel channels[MIXER_MAX_CHANNELS];
	int conf_mode, skip_threshold, master_channel;
	bool_t bypass_mode, single_output;
	MixerPool *pool;
	int slot;
	// Blocks forwarded in bypass mode leave one tick later, like mixed ones (which come out of the next tick's flush): the
	// filter's latency does not jump by 10 ms when a second contributor appears or the last but one falls silent -- a
	// canceller behind two mixers would otherwise see its two inputs slip against each other (aec3_tester.c graph).
	// A mixer with ONE linked input never leaves bypass mode (audiomixer.c:244-286: one contributor or none) -- an AudioStream's
	// outbound_mixer without a remote player, its local_mixer without a local one (audiostream.c:1585-1588,1770-1772,1807,1815):
	// nothing to jump between, so its blocks go on IN the walk, as the reference's do (one_input).
	std::vector<std::pair<int, mblk_t *>> *held;
	bool one_input;
	// that, outside conference mode with every linked output enabled: process() moves the pin's blocks on without the hub's (or the filter's)
	// lock -- nothing of the hub is touched and the filter's own state is not read.  Set at the attach, taken back for good (until the next
	// attach) by a method that changes an output or the mode; an acquire load in process()
	std::atomic<bool> forwards_unlocked;
	int fwd_pin, fwd_nout;
	MSQueue *fwd_out; // (the one linked output when fwd_nout == 1: the links stand while the filter is attached)
	// the conference and every leg that feeds it as one device-resident batch (filters/leg_chain.inl)
	LegBank *fbank;     // non-null: fused; the conference is slot `fconf` of that bank
	int fconf;
	ServerBank *sbank;  // non-null: fused as a conference of remote members (filters/server_leg.inl), slot `sconf`
	int sconf;
	int fuse_state;     // 0 not looked at since the attach, 1 fused, 2 refused
	std::atomic<bool> unfuse_wanted; // a member stopped qualifying (set by a method on any thread, honoured by the next process())
	bool first_walk;    // the walk right after an attach is still to come (see mixer_process)
	bool prepared, acquire_failed; // preprocess has sized the tick; a bank slot could not be had
};
void leg_push_mixer_controls(MSFilter *f, MixerState *s, bool from_method); // leg_chain.inl
void server_push_mixer_controls(MSFilter *f, MixerState *s, bool from_method); // server_leg.inl

void mixer_release_held(MSFilter *f, MixerState *s, bool deliver) {
	for (auto &pm : *s->held) {
		MSQueue *q = deliver ? f->outputs[pm.first] : NULL;
		if (q) ms_queue_put(q, pm.second);
		else freemsg(pm.second);
	}
	s->held->clear();
}

void mixer_init(MSFilter *f) { // audiomixer.c:145-156
	MixerState *s = (MixerState *)ms_malloc0(sizeof(*s));
	s->conf_mode = FALSE;
	s->nchannels = 1;
	s->rate = 44100;
	s->master_channel = -1;
	s->slot = -1;
	s->held = new std::vector<std::pair<int, mblk_t *>>();
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
		ms_bufferizer_init(&s->channels[i].bufferizer);
		s->channels[i].gain = 1.0;
		s->channels[i].active = TRUE;
		s->channels[i].output_enabled = TRUE;
	}
	f->data = s;
}
void mixer_uninit(MSFilter *f) {
	MixerState *s = (MixerState *)f->data;
	conf_unfuse(f, false);
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) ms_bufferizer_uninit(&s->channels[i].bufferizer);
	mixer_release_held(f, s, false);
	delete s->held;
	ms_free(s);
}
bool_t has_single_output(MSFilter *f, MixerState *s) { // audiomixer.c:167-176
	int count = 0;
	for (int i = 0; i < f->desc->noutputs; ++i)
		if (f->outputs[i] && s->channels[i].output_enabled) count++;
	return count == 1;
}
// from_method: set by a method on a running filter -- live behind the coming flush (MixerPool::next_*); otherwise (attach) at once
void mixer_push_controls(MSFilter *f, MixerState *s, bool from_method = false) {
	if (s->fbank) {
		leg_push_mixer_controls(f, s, from_method);
		return;
	}
	if (s->sbank) {
		server_push_mixer_controls(f, s, from_method);
		return;
	}
	if (!s->pool) return;
	MixerPool *p = s->pool;
	const bool later = from_method && f->ticker != NULL && p->work_waiting();
	std::vector<uint8_t> &fl_row = later ? p->next_flags : p->flags;
	std::vector<float> &g_row = later ? p->next_gain : p->gain;
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
		uint8_t fl = 0;
		if (f->inputs[i]) fl |= MI_MIX_LINKED;
		if (s->channels[i].active) fl |= MI_MIX_ACTIVE;
		if (f->outputs[i] && s->channels[i].output_enabled) fl |= MI_MIX_OUTPUT;
		fl_row[(size_t)s->slot * MIXER_MAX_CHANNELS + i] = fl;
		g_row[(size_t)s->slot * MIXER_MAX_CHANNELS + i] = s->channels[i].gain;
	}
	if (later) {
		p->next_set[(size_t)s->slot] = 1;
		p->next_any = true;
	} else {
		p->next_set[(size_t)s->slot] = 0; // (an attach supersedes what a method left waiting)
		p->ctl_dirty = true;
	}
}
void mixer_prepare(MSFilter *f, bool running = false);
void mixer_acquire(MSFilter *f);
void mixer_preprocess(MSFilter *f) { // audiomixer.c:178-200
	((MixerState *)f->data)->fuse_state = 0;
	mixer_prepare(f); // (the filter's own fields: it is being attached by this thread, nobody runs it yet)
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f);
}
// running: the conference left a fused batch while attached -- no preprocess in the reference's terms: the channels' clocks (census,
// flow control) and the bypass state run on, they are the very fields the batch kept (LegBank / ServerBank::conf_tick)
void mixer_prepare(MSFilter *f, bool running) { // (running: the hub locked by the caller; at an attach the filter is the attaching thread's alone)
	MixerState *s = (MixerState *)f->data;
	s->bytespertick = (2 * s->nchannels * s->rate * f->ticker->interval) / 1000;
	for (int i = 0; i < MIXER_MAX_CHANNELS && !running; ++i) {
		s->channels[i].last_flow_control = (uint64_t)-1;
		s->channels[i].last_activity = (uint64_t)-1;
	}
	s->skip_threshold = s->bytespertick * 2;
	s->first_walk = true; // (running: this walk's tick was the batch's)
	if (!running) s->bypass_mode = FALSE;
	s->single_output = has_single_output(f, s);
	int linked = 0;
	for (int i = 0; i < f->desc->ninputs; ++i) linked += f->inputs[i] != NULL;
	s->one_input = linked == 1;
	s->fwd_pin = -1;
	bool every_output = true;
	for (int i = 0; i < f->desc->ninputs; ++i)
		if (f->inputs[i]) s->fwd_pin = i;
	s->fwd_nout = 0, s->fwd_out = nullptr;
	for (int i = 0; i < f->desc->noutputs; ++i) {
		every_output = every_output && (!f->outputs[i] || s->channels[i].output_enabled);
		if (f->outputs[i]) s->fwd_out = f->outputs[i], s->fwd_nout++;
	}
	s->forwards_unlocked.store(!running && s->one_input && s->conf_mode == 0 && every_output, std::memory_order_release);
	s->prepared = true;
	s->acquire_failed = false;
	// (no bank slot yet: a conference that fuses at the attach never needs one of its own, a mixer that can only forward never mixes --
	// mixer_acquire at the first tick that has something to mix; `running`: the conference just left its batch and mixes from this walk on)
	if (running) mixer_acquire(f);
}
void mixer_acquire(MSFilter *f) { // (hub locked by the caller)
	MixerState *s = (MixerState *)f->data;
	if (s->pool || !s->prepared || s->acquire_failed) return;
	const int ns = s->bytespertick / 2;
	s->pool = bank<MixerPool>("mixer:" + std::to_string(ns), 1, [&](int cap) { return new MixerPool(cap, ns); });
	s->slot = s->pool ? s->pool->acquire(f) : -1;
	if (s->slot < 0) s->pool = nullptr, s->acquire_failed = true;
	else note_slot(f);
	mixer_push_controls(f, s);
}
void mixer_postprocess(MSFilter *f) { // audiomixer.c:202-208 (slot released at every detach)
	MixerState *s = (MixerState *)f->data;
	facade_detached(f);
	conf_unfuse(f, false);
	HubLock lk(f);
	mixer_release_held(f, s, false);
	s->prepared = false;
	s->forwards_unlocked.store(false, std::memory_order_release);
	if (s->pool) {
		s->pool->staged[(size_t)s->slot] = s->pool->ready[(size_t)s->slot] = 0;
		s->pool->release(s->slot); // the last release of a bank destroys it
	}
	s->pool = nullptr;
	s->slot = -1;
}

// ---- bypass: one contributor, nothing to sum (behaviour of audiomixer.c:219-286) -----------------------------------
// A pin "contributes" while it has data queued or had some less than BYPASS_MODE_TIMEOUT ms ago.
struct Contributors {
	int count = 0;
	int pin = -1; // the highest-numbered contributing pin (the one that forwards when count == 1)
};

Contributors mixer_census(MSFilter *f, MixerState *s) {
	Contributors c;
	const uint64_t now = ticker_now(f->ticker);
	for (int pin = 0; pin < f->desc->ninputs; ++pin) {
		if (!f->inputs[pin]) continue;
		uint64_t &seen = s->channels[pin].last_activity;
		bool contributes;
		if (!ms_queue_empty(f->inputs[pin])) {
			seen = now;
			contributes = true;
		} else if (seen == (uint64_t)-1) {
			seen = now; // first look at a silent pin only starts its clock
			contributes = false;
		} else {
			contributes = now - seen < BYPASS_MODE_TIMEOUT;
		}
		if (contributes) {
			c.count++;
			c.pin = pin;
		}
	}
	return c;
}

// The single contributor's blocks go to every enabled output except (in conference mode) its own pin: moved when
// only one output is wired, referenced (dupmsg) otherwise.
void mixer_forward(MSFilter *f, MixerState *s, int from_pin) {
	MSQueue *src = f->inputs[from_pin];
	for (int pin = 0; pin < f->desc->noutputs; ++pin) {
		MSQueue *dst = f->outputs[pin];
		if (!dst || !s->channels[pin].output_enabled) continue;
		if (s->conf_mode != 0 && pin == from_pin) continue;
		if (s->single_output) {
			for (mblk_t *m; (m = ms_queue_get(src)) != NULL;) {
				if (s->one_input) ms_queue_put(dst, m);
				else s->held->push_back({pin, m});
			}
			break;
		}
		for (mblk_t *m = peekq(&src->q); m != NULL && m != &src->q._q_stopper; m = m->b_next) {
			if (s->one_input) ms_queue_put(dst, dupmsg(m));
			else s->held->push_back({pin, dupmsg(m)});
		}
	}
	ms_queue_flush(src);
}

// true = this tick is already dealt with (forwarded, or nobody contributes)
bool_t mixer_check_bypass(MSFilter *f, MixerState *s) {
	const Contributors c = mixer_census(f, s);
	if (c.count > 1) {
		if (s->bypass_mode) ms_message("mi355x mixer %p: two or more contributors, mixing on the device again", (void *)f);
		s->bypass_mode = FALSE;
		return FALSE;
	}
	if (c.count == 1) {
		if (!s->bypass_mode) ms_message("mi355x mixer %p: a single contributor, forwarding its blocks", (void *)f);
		s->bypass_mode = TRUE;
		mixer_forward(f, s, c.pin);
	}
	return TRUE;
}

// ---- per-channel flow control (behaviour of audiomixer.c:92-111): every 5 s, if the bufferizer never dropped below
// `threshold` bytes in that window, discard the standing excess down to half the threshold.  Returns the bytes dropped.
// the decision alone, from the bufferizer's level in bytes (the fused chain keeps that level as a count)
int channel_flow_control_level(Channel *chan, int level, int threshold, uint64_t now) {
	const bool first_call = chan->last_flow_control == (uint64_t)-1;
	int dropped = 0;
	if (!first_call) {
		if (chan->min_fullness == -1 || level < chan->min_fullness) chan->min_fullness = level;
		if (now - chan->last_flow_control < 5000) return 0;
		if (chan->min_fullness >= threshold) dropped = chan->min_fullness - threshold / 2;
	}
	chan->last_flow_control = now; // a new observation window starts
	chan->min_fullness = -1;
	return dropped;
}
int channel_flow_control(Channel *chan, int threshold, uint64_t now) {
	const int dropped = channel_flow_control_level(chan, (int)ms_bufferizer_get_avail(&chan->bufferizer), threshold, now);
	if (dropped > 0) ms_bufferizer_skip_bytes(&chan->bufferizer, dropped);
	return dropped;
}

void MixerPool::emit(MSFilter *f, int slot) {
	MixerState *s = (MixerState *)f->data;
	if (!ready[(size_t)slot]) return;
	ready[(size_t)slot] = 0;
	const int16_t *base = h_out + (size_t)slot * MIXER_MAX_CHANNELS * ns;

	if (s->conf_mode == 0) { // one block shared by every enabled output (:321-334)
		mblk_t *om = NULL;
		for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
			MSQueue *q = f->outputs[i];
			if (q && s->channels[i].output_enabled) {
				if (om == NULL) {
					om = allocb((size_t)ns * 2, 0);
					memcpy(om->b_wptr, base, (size_t)ns * 2);
					om->b_wptr += ns * 2;
				} else {
					om = dupb(om);
				}
				ms_queue_put(q, om);
			}
		}
	} else { // :336-343
		for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
			MSQueue *q = f->outputs[i];
			if (q && s->channels[i].output_enabled) {
				mblk_t *om = allocb((size_t)ns * 2, 0);
				memcpy(om->b_wptr, base + (size_t)i * ns, (size_t)ns * 2);
				om->b_wptr += ns * 2;
				ms_queue_put(q, om);
			}
		}
	}
}

void mixer_process(MSFilter *f) { // audiomixer.c:288-346
	MixerState *s = (MixerState *)f->data;
	if (s->forwards_unlocked.load(std::memory_order_acquire)) { // one linked input, a plain mixer: bypass mode for life (audiomixer.c:244-286)
		MSQueue *src = f->inputs[s->fwd_pin];
		for (mblk_t *m; (m = ms_queue_get(src)) != NULL;) {
			if (s->fwd_nout == 1) {
				ms_queue_put(s->fwd_out, m);
				continue;
			}
			for (int pin = 0; pin < f->desc->noutputs; ++pin)
				if (f->outputs[pin]) ms_queue_put(f->outputs[pin], dupmsg(m));
			freemsg(m);
		}
		return;
	}
	const double trace_ms = s->fbank ? leg_trace_ms(s->fbank) : 0.0; // MSMI355X_TRACE_SLOW_MS
	const uint64_t tr0 = trace_ms > 0 ? leg_trace_now() : 0;
	// lock order everywhere: the hub first, the filter's own lock inside it (the flush task pumps this filter with the hub held)
	HubLock lk(f, s->fbank ? leg_pool_of(s->fbank) : (s->sbank ? server_pool_of(s->sbank) : static_cast<Pool *>(s->pool)));
	ms_filter_lock(f);
	const uint64_t tr1 = trace_ms > 0 ? leg_trace_now() : 0;
	if (s->unfuse_wanted && (s->fbank || s->sbank)) conf_unfuse(f, true); // a member stopped qualifying: back to the facades' own banks
	if (s->fuse_state == 0 && !s->fbank && !s->sbank) conf_try_fuse(f); // (normally a leg's head got here first)
	if (s->sbank) { // a conference of remote members: as below, on its own kind of bank
		mixer_release_held(f, s, true);
		request_flush(f);
		server_conf_walked(s->sbank, s->sconf);
		ms_filter_unlock(f);
		return;
	}
	if (s->fbank) { // fused: the conference ticks inside the hub's flush; a pump keeps that flush coming every tick
		// (no census here: what the members staged in THIS walk meets the mixer in LegBank::conf_tick, whose three cases are
		// mixer_check_bypass's, audiomixer.c:244-286 -- a pin's clock starts at its first look there, without counting yet)
		mixer_release_held(f, s, true);
		request_flush(f);
		const uint64_t tr2 = trace_ms > 0 ? leg_trace_now() : 0;
		leg_conf_walked(s->fbank, s->fconf); // the last conference of the bank to be walked sends the bank's work to the device right away
		ms_filter_unlock(f);
		if (trace_ms > 0) {
			const uint64_t tr3 = leg_trace_now();
			if ((double)(tr3 - tr0) * 1e-6 > trace_ms)
				fprintf(stderr, "mi355x mixer %p tick %u: process() took %.2f ms: locks %.2f, census + flush request %.2f, bank walked / enqueue %.2f\n", (void *)f,
				        (unsigned)f->ticker->ticks, (double)(tr3 - tr0) * 1e-6, (double)(tr1 - tr0) * 1e-6, (double)(tr2 - tr1) * 1e-6, (double)(tr3 - tr2) * 1e-6);
		}
		return;
	}
	if (already_ran_this_tick(f)) { // the flush task pumped this mixer right behind the facades that feed it
		ms_filter_unlock(f);
		return;
	}
	// Fed by facades of this plugin only, the mixer meets a walk's blocks one tick later: pumped by the flush that delivers them,
	// or -- when that flush brought it nothing -- in the next walk.  The walk right after an attach has no such predecessor: a
	// census here would start every pin's clock a tick before the walk it belongs to and make the conference "contribute" (and
	// stream out 10 ms of silence) where the reference's first walk finds only first looks (audiomixer.c:258-260)
	if (s->first_walk && !g_hub.in_flush) {
		s->first_walk = false;
		if (all_inputs_ours(f) && !inputs_waiting(f)) {
			ms_filter_unlock(f);
			return;
		}
	}
	s->first_walk = false;
	mixer_release_held(f, s, true); // what bypass mode forwarded on the previous tick
	if (!s->prepared || s->acquire_failed) {
		ms_filter_unlock(f);
		return;
	}
	if (mixer_check_bypass(f, s)) {
		ms_filter_unlock(f);
		return;
	}
	mixer_acquire(f); // two or more contributors: a bank slot of its own from here on
	if (!s->pool) {
		ms_filter_unlock(f);
		return;
	}
	MixerPool *p = s->pool;
	const int nwords = s->bytespertick / 2;
	int16_t *in = p->h_in + (size_t)s->slot * MIXER_MAX_CHANNELS * nwords;
	uint8_t *has = p->h_has + (size_t)s->slot * MIXER_MAX_CHANNELS;
	for (int i = 0; i < f->desc->ninputs; ++i) {
		MSQueue *q = f->inputs[i];
		has[i] = 0;
		if (!q) continue;
		Channel *chan = &s->channels[i];
		ms_bufferizer_put_from_queue(&chan->bufferizer, q); // channel_process_in :78-90
		has[i] = ms_bufferizer_read(&chan->bufferizer, (uint8_t *)(in + (size_t)i * nwords), (size_t)nwords * 2) != 0;
		const int skip = channel_flow_control(chan, s->skip_threshold, ticker_now(f->ticker));
		if (skip > 0)
			ms_warning("mi355x mixer: pin %i kept more than two ticks queued for 5 s; %i ms discarded", i, (skip * 1000) / (2 * s->nchannels * s->rate));
	}
	p->h_mode[(size_t)s->slot] = (uint8_t)(s->conf_mode != 0);
	p->staged[(size_t)s->slot] = 1; // ALWAYS_STREAMOUT :315-317
	request_flush(f);
	ms_filter_unlock(f);
}

int mixer_set_rate(MSFilter *f, void *data) {
	MixerState *s = (MixerState *)f->data;
	if ((s->fbank || s->sbank) && s->rate != *(int *)data) s->unfuse_wanted = true;
	s->rate = *(int *)data;
	return 0;
}
int mixer_get_rate(MSFilter *f, void *data) {
	*(int *)data = ((MixerState *)f->data)->rate;
	return 0;
}
int mixer_set_nchannels(MSFilter *f, void *data) {
	MixerState *s = (MixerState *)f->data;
	if ((s->fbank || s->sbank) && s->nchannels != *(int *)data) s->unfuse_wanted = true;
	s->nchannels = *(int *)data;
	return 0;
}
int mixer_get_nchannels(MSFilter *f, void *data) {
	*(int *)data = ((MixerState *)f->data)->nchannels;
	return 0;
}
bool mixer_pin_ok(const char *who, int pin) {
	if (pin < 0 || pin >= MIXER_MAX_CHANNELS) {
		ms_warning("mi355x mixer, %s: there is no pin %i", who, pin);
		return false;
	}
	return true;
}
int mixer_set_input_gain(MSFilter *f, void *data) { // audiomixer.c:372-382
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_set_input_gain", ctl->pin)) return -1;
	HubLock lk(f);
	s->channels[ctl->pin].gain = ctl->param.gain;
	mixer_push_controls(f, s, true);
	return 0;
}
int mixer_set_active(MSFilter *f, void *data) { // :384-393
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_set_active_gain", ctl->pin)) return -1;
	HubLock lk(f);
	s->channels[ctl->pin].active = (bool_t)ctl->param.active;
	mixer_push_controls(f, s, true);
	return 0;
}
int mixer_enable_output(MSFilter *f, void *data) { // :395-408
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_enable_output", ctl->pin)) return -1;
	HubLock lk(f);
	ms_filter_lock(f);
	s->channels[ctl->pin].output_enabled = (bool_t)ctl->param.enabled;
	s->forwards_unlocked.store(false, std::memory_order_release); // (from the next walk on under the locks, whatever the new value)
	s->single_output = has_single_output(f, s);
	mixer_push_controls(f, s, true);
	ms_filter_unlock(f);
	return 0;
}
int mixer_set_conference_mode(MSFilter *f, void *data) {
	MixerState *s = (MixerState *)f->data;
	if ((s->fbank || s->sbank) && *(int *)data == 0) s->unfuse_wanted = true; // the fused batch mixes in conference mode only
	s->conf_mode = *(int *)data;
	s->forwards_unlocked.store(false, std::memory_order_release);
	return 0;
}
int mixer_set_master_channel(MSFilter *f, void *data) {
	((MixerState *)f->data)->master_channel = *(int *)data;
	return 0;
}
MSFilterMethod mixer_methods[] = {{MS_FILTER_SET_NCHANNELS, mixer_set_nchannels},
                                  {MS_FILTER_GET_NCHANNELS, mixer_get_nchannels},
                                  {MS_FILTER_SET_SAMPLE_RATE, mixer_set_rate},
                                  {MS_FILTER_GET_SAMPLE_RATE, mixer_get_rate},
                                  {MS_AUDIO_MIXER_SET_INPUT_GAIN, mixer_set_input_gain},
                                  {MS_AUDIO_MIXER_SET_ACTIVE, mixer_set_active},
                                  {MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE, mixer_set_conference_mode},
                                  {MS_AUDIO_MIXER_SET_MASTER_CHANNEL, mixer_set_master_channel},
                                  {MS_AUDIO_MIXER_ENABLE_OUTPUT, mixer_enable_output},
                                  {0, NULL}};
