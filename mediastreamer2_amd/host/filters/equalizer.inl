// filters/equalizer.inl -- MSEqualizer facade (src/audiofilters/equalizer.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// =================================================================== equalizer
struct EqualizerPool : Pool {
	int rate, cap_samples;
	mi_equalizer *e = nullptr;
	int16_t *h_buf, *d_buf;
	int32_t *h_n, *d_n, *h_nsc; // (h_nsc: as VolumePool's)
	std::vector<int> staged, ready;
	// MS_EQUALIZER_SET_GAIN / SET_ACTIVE while the last walk's blocks are still waiting for the coming flush (Pool::work_waiting): they go
	// live behind that flush (flushed()), as the reference's process() of that walk ran before the call
	struct Op {
		int slot, kind; // kind 0: gain, 1: active, 2: flatten (MS_FILTER_SET_SAMPLE_RATE: equalizer_rate_update re-allocates a flat spectrum at ANY rate, equalizer.c:305-309)
		MSEqualizerGain g;
		int active;
	};
	std::vector<Op> later;
	std::vector<uint8_t> used; // the slot's FIR memory may hold an earlier owner's samples (the batch is created cleared)
	// equalizer_rate_update (equalizer.c:57-79): a flat spectrum AND a cleared FIR memory, whatever the rate was
	static int rate_update(mi_equalizer *eq, int slot) {
		const int rc = mi_equalizer_flatten(eq, slot);
		return rc != MI_OK ? rc : mi_equalizer_set_history(eq, slot, nullptr, mi_equalizer_fir_len(eq));
	}
	static void apply(mi_equalizer *eq, const Op &o) {
		const int rc = o.kind == 0 ? mi_equalizer_set_gain(eq, o.slot, o.g.frequency, o.g.gain, o.g.width) : (o.kind == 1 ? mi_equalizer_set_active(eq, o.slot, o.active) : rate_update(eq, o.slot));
		if (rc != MI_OK) ms_error("mi355x equalizer: a deferred method failed: %s", mi_last_error());
	}
	void flushed() override {
		if (later.empty()) return;
		std::vector<Op> keep;
		for (const Op &o : later) {
			if (parked(o.slot)) keep.push_back(o);
			else if (!failed) apply(e, o);
		}
		later.swap(keep);
	}
	EqualizerPool(int cap, int r) : rate(r) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_equalizer_create(hub->ctx, capacity, rate, &e));
		cap_samples = (std::max(960, rate / 100 * 2) + 7) & ~7;
		const size_t c = (size_t)capacity;
		h_buf = pinned<int16_t>(kMaxRounds * c * cap_samples);
		h_n = pinned<int32_t>(kMaxRounds * c);
		h_nsc = pinned<int32_t>(kMaxRounds * c);
		d_buf = devmem<int16_t>(c * cap_samples);
		d_n = devmem<int32_t>(c);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	~EqualizerPool() override {
		if (e) mi_equalizer_destroy(e);
	}
	bool enqueue() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity, u = (size_t)hi; // rows [0, hi) are all that was ever handed out
		int maxr = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			const int32_t *nrow = h_n + r * c;
			if (hub->scope) { // a detaching graph's slots alone (see VolumePool::enqueue)
				for (int s = 0; s < capacity; ++s) h_nsc[r * c + s] = (s < hi && staged[(size_t)s] > r && !parked(s)) ? h_n[r * c + s] : 0;
				nrow = h_nsc + r * c;
			} else {
				for (int s = 0; s < capacity; ++s)
					if (s >= hi || staged[(size_t)s] <= r) h_n[r * c + s] = 0;
			}
			if (zero_copy_rows()) { // (in place in pinned memory, as VolumePool)
				MI_MUST(mi_equalizer_process_masked(e, h_buf + r * c * cap_samples, cap_samples, cap_samples, nrow));
			} else {
				MI_MUST(mi_copy_h2d_pinned(ctx, d_buf, h_buf + r * c * cap_samples, u * cap_samples * 2));
				MI_MUST(mi_copy_h2d_pinned(ctx, d_n, nrow, c * 4));
				MI_MUST(mi_equalizer_process_masked(e, d_buf, cap_samples, cap_samples, d_n));
				MI_MUST(mi_copy_d2h_pinned(ctx, h_buf + r * c * cap_samples, d_buf, u * cap_samples * 2));
			}
		}
		return maxr > 0;
	}
	void finish() override {
		for (int s = 0; s < hi; ++s) { // after a failed launch the staged blocks leave as they came (flat response)
			if (parked(s)) continue;
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void emit(MSFilter *f, int slot) override {
		const size_t c = (size_t)capacity, s = (size_t)slot;
		for (int r = 0; r < ready[s]; ++r) {
			const int n = h_n[r * c + s];
			mblk_t *om = allocb((size_t)n * 2, 0);
			memcpy(om->b_wptr, h_buf + (r * c + s) * cap_samples, (size_t)n * 2);
			om->b_wptr += n * 2;
			if (f->outputs[0]) ms_queue_put(f->outputs[0], om);
			else freemsg(om);
		}
		ready[s] = 0;
	}
};

struct EqualizerData {
	int rate;
	bool active;
	EqualizerPool *pool;
	int slot;
	std::vector<MSEqualizerGain> *pending; // gains since the last rate change, in call order
	MSBufferizer *spill;                   // the part of an over-long block that did not fit this tick's rounds
	// mic_equalizer between MSResample and MSSpeexEC of a fused call leg (audiostream.c:1801; filters/leg_chain.inl): it runs in that
	// leg's bank.  The FIR's memory goes with the FILTER from slot to slot (equalizer.c keeps it as long as the filter lives)
	FusedLeg *leg;
	std::vector<int16_t> *hist;
	bool has_hist;
	// it has been active at some point since the attach: from then on its blocks are staged (a tick of latency) whether it is active or not -- a
	// filter that went back to forwarding in the walk when it is switched off would let two blocks meet in one tick (equalizer_passes)
	bool was_active;
	// equalizer_passes() held at the attach and nothing has switched the filter on since: process() hands the blocks on without the hub's
	// lock (nothing of the hub is touched).  Taken back for good -- until the next attach -- by MS_EQUALIZER_SET_ACTIVE(true); an acquire load
	std::atomic<bool> passes_unlocked;
};
// An MSEqualizer that is NOT active hands every block on as it came, in the walk, and leaves its FIR's memory alone (equalizer.c:279-288:
// `if (s->active) equalizer_state_run(..)` around the same ms_queue_put).  The reference's AudioStream creates BOTH equalizers whenever
// AUDIO_STREAM_FEATURE_EQUALIZER is set -- part of AUDIO_STREAM_FEATURE_ALL -- and leaves them inactive unless the application or the
// sound device's description asks (audiostream.c:1623-1640): mic_equalizer in front of the canceller's microphone pin, spk_equalizer in
// front of its far end (:1801,:1828).  Such a filter is transparent: no bank slot, no tick of latency, and a fused leg is recognised
// THROUGH it (equalizer_passes; leg_chain.inl).  Activated in mid-call it stages its blocks from the next walk on (a tick later, like
// any facade) and the leg it stands in goes back to its facades.
bool equalizer_passes(MSFilter *g, MSTicker *ticker) { // (hub locked)
	if (!g || g->desc != &ms_mi355x_equalizer_desc || g->ticker != ticker) return false;
	const EqualizerData *d = (const EqualizerData *)g->data;
	if (d->active || d->was_active || d->leg || ms_bufferizer_get_avail(d->spill)) return false;
	return !d->pool || (d->pool->staged[(size_t)d->slot] == 0 && d->pool->ready[(size_t)d->slot] == 0);
}
// (for a neighbour's preprocess, which may run before this filter's own: `was_active` is then still the previous attach's)
bool equalizer_idle(MSFilter *g) { return g && g->desc == &ms_mi355x_equalizer_desc && !((const EqualizerData *)g->data)->active && !((const EqualizerData *)g->data)->leg; }
FusedLeg *leg_fed_far_end_by(MSFilter *f); // leg_chain.inl: the fused leg whose far end / microphone passes through this filter
FusedLeg *leg_fed_mic_by(MSFilter *f);
void leg_eq_op(FusedLeg *leg, const EqualizerPool::Op &op); // leg_chain.inl
mi_equalizer *leg_eq(FusedLeg *leg, int *slot);
// the batch and slot the filter's equalizer lives in right now (hub locked): its own bank's, or its fused leg's
mi_equalizer *equalizer_where(EqualizerData *d, int *slot) {
	if (d->leg) return leg_eq(d->leg, slot);
	*slot = d->slot;
	return d->pool ? d->pool->e : nullptr;
}

// Gains set before the filter is attached to a ticker are kept in `pending` and replayed, in
// order, when the slot is acquired (the reference keeps them in its own fft_cpx array).
void equalizer_attach(MSFilter *f) {
	EqualizerData *d = (EqualizerData *)f->data;
	if (d->pool) { // the slot goes back under ITS hub's lock when the rate or the ticker changed, or its bank failed
		HubLock old(f);
		if (!d->pool->failed && d->pool->rate == d->rate && d->pool->hub->ticker == f->ticker) return;
		d->pool->release(d->slot);
		d->pool = nullptr;
		d->slot = -1;
	}
	if (!f->ticker) return;
	HubLock lk(f);
	const int rate = d->rate;
	d->pool = bank<EqualizerPool>("equalizer:" + std::to_string(rate), 1, [&](int cap) { return new EqualizerPool(cap, rate); });
	d->slot = d->pool ? d->pool->acquire(f) : -1;
	if (d->slot < 0) {
		d->pool = nullptr;
		return;
	}
	note_slot(f);
	MI_MUST(mi_equalizer_flatten(d->pool->e, d->slot)); // equalizer_rate_update flattens
	MI_MUST(mi_equalizer_set_active(d->pool->e, d->slot, d->active));
	for (const MSEqualizerGain &g : *d->pending)
		MI_MUST(mi_equalizer_set_gain(d->pool->e, d->slot, g.frequency, g.gain, g.width));
	// a slot is a new filter's (cleared memory) unless this filter comes back from a fused leg with its own
	if (d->pool->used.size() != (size_t)d->pool->capacity) d->pool->used.assign((size_t)d->pool->capacity, 0);
	if (d->has_hist || d->pool->used[(size_t)d->slot])
		MI_MUST(mi_equalizer_set_history(d->pool->e, d->slot, d->has_hist ? d->hist->data() : nullptr, mi_equalizer_fir_len(d->pool->e)));
	d->pool->used[(size_t)d->slot] = 1;
	d->has_hist = false;
}

void equalizer_init(MSFilter *f) { // equalizer.c:271-273: default rate 8000
	EqualizerData *d = (EqualizerData *)ms_malloc0(sizeof(*d));
	d->rate = 8000;
	d->active = true;
	d->slot = -1;
	d->pending = new std::vector<MSEqualizerGain>();
	d->spill = ms_bufferizer_new();
	d->hist = new std::vector<int16_t>();
	f->data = d;
}
void equalizer_preprocess(MSFilter *f) {
	((EqualizerData *)f->data)->was_active = ((EqualizerData *)f->data)->active;
	if (graph_ready(f) || ((EqualizerData *)f->data)->pool) {
		HubLock lk(f);
		graph_preprocessed(f);
		((EqualizerData *)f->data)->passes_unlocked.store(equalizer_passes(f, f->ticker), std::memory_order_release);
	} else ((EqualizerData *)f->data)->passes_unlocked.store(equalizer_passes(f, f->ticker), std::memory_order_release); // (no slot: nothing of the hub's is read)
	// (its own slot, where it needs one, is taken by the graph's last facade to be preprocessed -- attach.inl -- after the legs have been recognised;
	// a graph some of whose facades were configured after the attach: process() takes it)
}
void equalizer_postprocess(MSFilter *f) {
	EqualizerData *d = (EqualizerData *)f->data;
	d->passes_unlocked.store(false, std::memory_order_release);
	facade_detached(f);
	if (d->leg) leg_release(d->leg, false);
}
void equalizer_uninit(MSFilter *f) {
	EqualizerData *d = (EqualizerData *)f->data;
	if (d->leg) leg_release(d->leg, false);
	delete d->hist;
	if (d->pool) {
		HubLock lk(f);
		d->pool->release(d->slot);
	}
	delete d->pending;
	ms_bufferizer_destroy(d->spill);
	ms_free(d);
}
void equalizer_process(MSFilter *f) { // equalizer.c:279-288
	EqualizerData *d = (EqualizerData *)f->data;
	mblk_t *m;
	if (d->passes_unlocked.load(std::memory_order_acquire)) { // inactive since the attach (equalizer.c:281-286): the blocks as they came
		while ((m = ms_queue_get(f->inputs[0])) != NULL) {
			if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
			else freemsg(m);
		}
		return;
	}
	if (d->leg) { // part of a fused leg: the leg's MSResample emits nothing, the equalizer runs in the leg's bank
		ms_queue_flush(f->inputs[0]);
		return;
	}
	HubLock lk(f, d->pool);
	if (equalizer_passes(f, f->ticker)) { // not active: the blocks as they came, now (equalizer.c:281-286)
		while ((m = ms_queue_get(f->inputs[0])) != NULL) {
			if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
			else freemsg(m);
		}
		return;
	}
	if (!d->pool) equalizer_attach(f);
	if (!d->pool) {
		ms_queue_flush(f->inputs[0]);
		return;
	}
	EqualizerPool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	// one FIR block per mblk; a block longer than a batch row is cut into row-sized pieces (a streaming filter: the
	// sample sequence does not depend on the blocking), nothing is dropped
	for (;;) {
		if (p->staged[s] >= kMaxRounds) {
			if (ms_bufferizer_get_avail(d->spill) == 0 && ms_queue_empty(f->inputs[0])) break;
			p->flush(); // more pieces than launch rounds in one tick: what is staged goes out now
			p->emit_all();
		}
		int16_t *row = p->h_buf + (p->staged[s] * c + s) * p->cap_samples;
		int n = 0;
		const size_t spilled = ms_bufferizer_get_avail(d->spill);
		if (spilled) {
			n = (int)std::min(spilled / 2, (size_t)p->cap_samples);
			ms_bufferizer_read(d->spill, (uint8_t *)row, (size_t)n * 2);
		} else if ((m = ms_queue_get(f->inputs[0])) != NULL) {
			n = (int)(msgdsize(m) / 2);
			if (n > p->cap_samples) {
				ms_bufferizer_put(d->spill, m);
				continue;
			}
			memcpy(row, m->b_rptr, (size_t)n * 2);
			freemsg(m);
		} else {
			break;
		}
		p->h_n[p->staged[s] * c + s] = n;
		p->staged[s]++;
	}
	if (p->staged[s]) request_flush(f);
}
int equalizer_set_gain(MSFilter *f, void *arg) { // equalizer.c:290-295
	EqualizerData *d = (EqualizerData *)f->data;
	MSEqualizerGain *g = (MSEqualizerGain *)arg;
	HubLock lk(f);
	d->pending->push_back(*g);
	const EqualizerPool::Op op{d->slot, 0, *g, 0};
	if (d->leg) {
		leg_eq_op(d->leg, op);
		return 0;
	}
	if (!d->pool) return 0;
	if (f->ticker && d->pool->work_waiting()) {
		d->pool->later.push_back(op);
		return 0;
	}
	return mi_equalizer_set_gain(d->pool->e, d->slot, g->frequency, g->gain, g->width) == MI_OK ? 0 : -1;
}
int equalizer_get_gain(MSFilter *f, void *arg) { // equalizer.c:297-303 incl. its slot-indexing quirk
	EqualizerData *d = (EqualizerData *)f->data;
	MSEqualizerGain *g = (MSEqualizerGain *)arg;
	g->width = 0;
	g->gain = 0;
	HubLock lk(f);
	int slot = -1;
	mi_equalizer *e = equalizer_where(d, &slot);
	if (!e) return -1;
	const int nfft = mi_equalizer_fir_len(e);
	std::vector<float> dump((size_t)nfft / 2);
	if (mi_equalizer_dump(e, slot, dump.data(), nfft / 2) != MI_OK) return -1;
	int hz = (int)g->frequency;
	if (hz >= 0) {
		if (hz > d->rate / 2) hz = d->rate / 2;
		int idx = ((hz * nfft) + (d->rate / 2)) / d->rate;
		if (idx == nfft / 2) idx = nfft / 2 - 1;
		// the reference reads fft_cpx[idx*2]: an imaginary slot, 0 for idx >= 1; DC slot for idx == 0
		g->gain = idx == 0 ? dump[0] * nfft : 0.f;
	}
	return 0;
}
int equalizer_set_rate(MSFilter *f, void *arg) { // equalizer.c:305-309
	EqualizerData *d = (EqualizerData *)f->data;
	HubLock lk(f);
	if (d->leg && d->rate != *(int *)arg) leg_disqualify(d->leg);
	d->rate = *(int *)arg;
	d->pending->clear(); // equalizer_rate_update re-allocates a flat spectrum
	if (d->leg) { // (another rate: the leg leaves its bank at its next walk, equalizer_attach then starts from the flat spectrum; the same rate: the slot in the leg's bank is flattened, behind the walk's blocks like every method)
		leg_eq_op(d->leg, EqualizerPool::Op{d->slot, 2, MSEqualizerGain{0, 0, 0}, 0});
		return 0;
	}
	if (d->pool && d->pool->rate == d->rate) { // (behind the blocks of the walk that preceded the call, like the other methods: DESIGN 6.5)
		if (f->ticker && d->pool->work_waiting()) d->pool->later.push_back(EqualizerPool::Op{d->slot, 2, MSEqualizerGain{0, 0, 0}, 0});
		else MI_MUST(EqualizerPool::rate_update(d->pool->e, d->slot));
		d->has_hist = false;
	} else equalizer_attach(f);
	return 0;
}
int equalizer_set_active(MSFilter *f, void *arg) { // equalizer.c:311-315: arg read as bool_t
	EqualizerData *d = (EqualizerData *)f->data;
	HubLock lk(f);
	d->active = *(bool_t *)arg != 0;
	if (d->active) d->was_active = true, d->passes_unlocked.store(false, std::memory_order_release);
	if (d->active && !d->leg) { // no longer transparent: a fused leg recognised through it goes back to its facades
		leg_disqualify(leg_fed_far_end_by(f));
		leg_disqualify(leg_fed_mic_by(f));
	}
	const EqualizerPool::Op op{d->slot, 1, MSEqualizerGain{0, 0, 0}, d->active ? 1 : 0};
	if (d->leg) leg_eq_op(d->leg, op);
	else if (d->pool && f->ticker && d->pool->work_waiting()) d->pool->later.push_back(op);
	else if (d->pool) MI_MUST(mi_equalizer_set_active(d->pool->e, d->slot, d->active));
	return 0;
}
int equalizer_dump(MSFilter *f, void *arg) {
	EqualizerData *d = (EqualizerData *)f->data;
	HubLock lk(f);
	int slot = -1;
	mi_equalizer *e = equalizer_where(d, &slot);
	if (!e) return -1;
	return mi_equalizer_dump(e, slot, (float *)arg, mi_equalizer_fir_len(e) / 2) == MI_OK ? 0 : -1;
}
int equalizer_get_nfreqs(MSFilter *f, void *arg) {
	EqualizerData *d = (EqualizerData *)f->data;
	*(int *)arg = (d->rate < 16000 ? 128 : (d->rate < 32000 ? 256 : 512)) / 2;
	return 0;
}
MSFilterMethod equalizer_methods[] = {{MS_EQUALIZER_SET_GAIN, equalizer_set_gain},
                                      {MS_EQUALIZER_GET_GAIN, equalizer_get_gain},
                                      {MS_EQUALIZER_SET_ACTIVE, equalizer_set_active},
                                      {MS_FILTER_SET_SAMPLE_RATE, equalizer_set_rate},
                                      {MS_EQUALIZER_DUMP_STATE, equalizer_dump},
                                      {MS_EQUALIZER_GET_NUM_FREQUENCIES, equalizer_get_nfreqs},
                                      {0, NULL}};
