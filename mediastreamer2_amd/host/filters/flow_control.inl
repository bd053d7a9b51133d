// filters/flow_control.inl -- MSAudioFlowControl facade (src/audiofilters/flowcontrol.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ---- MSAudioFlowControl flowcontrol.c:154-279
constexpr int kFlowBlock = 2048; // samples per staged block (mi_flowctl's limit); longer blocks are split
struct FlowPool : Pool {
	mi_flowctl *fc = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	int32_t *h_len, *h_olen, *d_len, *d_olen;
	int32_t *h_lensc; // the rounds' length rows while a detaching graph's slots alone are flushed (see VolumePool::enqueue)
	std::vector<uint32_t> req_drop, req_total; // pending MS_AUDIO_FLOW_CONTROL_DROP requests ...
	std::vector<int> req_round;                // ... and how many staged blocks of the stream precede each
	std::vector<uint32_t> arm_drop, arm_total;
	bool have_req = false;
	// A method between two walks meets the NEXT walk's blocks in the reference (the filter's next process()).  Here the last walk's blocks
	// may still be on their way to this bank (they arrive with the coming flush, Pool::work_waiting): a request made then waits in
	// later_* and is armed when that flush is through (flushed()) -- in front of the blocks of the walk the call preceded
	std::vector<uint32_t> later_drop, later_total;
	bool have_later = false;
	std::vector<int> staged, ready;
	std::vector<uint8_t> used; // the slot has had a filter since the bank was created (a fresh one is a controller at rest with the default configuration: mi_flowctl_create)
	std::vector<std::vector<mblk_t *>> held, done; // the blocks themselves: the dropper edits them in place
	explicit FlowPool(int cap) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_flowctl_create(hub->ctx, capacity, kFlowBlock, &fc));
		const size_t c = (size_t)capacity;
		h_in = pinned<int16_t>(kMaxRounds * c * kFlowBlock);
		h_out = pinned<int16_t>(kMaxRounds * c * kFlowBlock);
		h_len = pinned<int32_t>(kMaxRounds * c);
		h_lensc = pinned<int32_t>(kMaxRounds * c);
		h_olen = pinned<int32_t>(kMaxRounds * c);
		d_in = devmem<int16_t>(c * kFlowBlock);
		d_out = devmem<int16_t>(c * kFlowBlock);
		d_len = devmem<int32_t>(c);
		d_olen = devmem<int32_t>(c);
		req_drop.assign(c, 0);
		req_total.assign(c, 0);
		req_round.assign(c, 0);
		arm_drop.assign(c, 0);
		arm_total.assign(c, 0);
		later_drop.assign(c, 0);
		later_total.assign(c, 0);
		staged.assign(c, 0);
		ready.assign(c, 0);
		used.assign(c, 0);
		held.resize(c);
		done.resize(c);
	}
	~FlowPool() override {
		if (fc) mi_flowctl_destroy(fc);
	}
	void flush() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity;
		// MS_AUDIO_FLOW_CONTROL_DROP calls since the last launch (:199-211) take effect exactly where they fell in the
		// stream's block sequence: before round r for a request that r staged blocks preceded (last = everything left).
		// A stream that is still dropping ignores its request on the device, like :204 does.
		auto arm = [&](int r, bool last) {
			if (!have_req) return;
			bool any = false, left = false;
			for (int s = 0; s < capacity; ++s) {
				arm_drop[(size_t)s] = arm_total[(size_t)s] = 0;
				if (req_drop[(size_t)s] == 0 && req_total[(size_t)s] == 0) continue;
				if (s < hi && parked(s)) { // (not this flush's business: the request waits for the slot's own)
					left = true;
					continue;
				}
				if (last || req_round[(size_t)s] <= r) {
					arm_drop[(size_t)s] = req_drop[(size_t)s], arm_total[(size_t)s] = req_total[(size_t)s];
					req_drop[(size_t)s] = req_total[(size_t)s] = 0;
					any = true;
				} else left = true;
			}
			if (any) MI_MUST(mi_flowctl_request_drop(fc, arm_drop.data(), arm_total.data()));
			have_req = left;
		};
		int maxr = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxr = std::max(maxr, staged[(size_t)s]);
		if (failed) { // a broken context is not given more work: the staged blocks leave as they came, nothing dropped
			for (int r = 0; r < maxr; ++r) {
				memcpy(h_out + r * c * kFlowBlock, h_in + r * c * kFlowBlock, c * kFlowBlock * 2);
				for (int s = 0; s < hi; ++s)
					if (!parked(s)) h_olen[r * c + s] = staged[(size_t)s] > r ? h_len[r * c + s] : 0;
			}
			have_req = false;
			std::fill(req_drop.begin(), req_drop.end(), 0u);
			std::fill(req_total.begin(), req_total.end(), 0u);
			maxr = 0;
		}
		for (int r = 0; r < maxr; ++r) {
			arm(r, false);
			const int32_t *lrow = h_len + r * c;
			if (hub->scope) {
				for (int s = 0; s < capacity; ++s) h_lensc[r * c + s] = (s < hi && staged[(size_t)s] > r && !parked(s)) ? h_len[r * c + s] : 0;
				lrow = h_lensc + r * c;
			} else {
				for (int s = 0; s < capacity; ++s)
					if (staged[(size_t)s] <= r) h_len[r * c + s] = 0;
			}
			MI_MUST(mi_copy_h2d_pinned(ctx, d_in, h_in + r * c * kFlowBlock, c * kFlowBlock * 2));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_len, lrow, c * 4));
			MI_MUST(mi_flowctl_process(fc, d_in, kFlowBlock, d_len, kFlowBlock, d_out, kFlowBlock, d_olen));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_out + r * c * kFlowBlock, d_out, c * kFlowBlock * 2));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_olen + r * c, d_olen, c * 4));
		}
		if (!failed) arm(maxr, true);
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			if (s < hi && parked(s)) continue;
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
			done[(size_t)s].swap(held[(size_t)s]);
			held[(size_t)s].clear();
		}
	}
	bool scoped() const override { return true; }
	void flushed() override {
		if (!have_later) return;
		bool left = false;
		for (int s = 0; s < capacity; ++s) {
			if (later_drop[(size_t)s] == 0 && later_total[(size_t)s] == 0) continue;
			if (s < hi && parked(s)) {
				left = true;
				continue;
			}
			if (req_drop[(size_t)s] == 0 && req_total[(size_t)s] == 0) { // (ignored while one is pending, like :204)
				req_drop[(size_t)s] = later_drop[(size_t)s], req_total[(size_t)s] = later_total[(size_t)s];
				req_round[(size_t)s] = staged[(size_t)s];
				have_req = true;
			}
			later_drop[(size_t)s] = later_total[(size_t)s] = 0;
		}
		have_later = left;
	}
	void emit(MSFilter *f, int slot) override {
		const size_t c = (size_t)capacity, s = (size_t)slot;
		for (int r = 0; r < ready[s]; ++r) {
			mblk_t *m = done[s][(size_t)r];
			const int left = h_olen[r * c + s];
			if (left > 0 && f->outputs[0]) {
				memcpy(m->b_rptr, h_out + (r * c + s) * kFlowBlock, (size_t)left * 2);
				m->b_wptr = m->b_rptr + (size_t)left * 2; // m->b_wptr -= 2 per deleted sample (:84)
				ms_queue_put(f->outputs[0], m);
			} else freemsg(m); // dropped entirely (:118,:131,:139)
		}
		ready[s] = 0;
		done[s].clear();
	}
};

struct FlowFilter { // MSAudioFlowControlState :154-158
	FlowPool *pool;
	int slot;
	int samplerate, nchannels;
	MSAudioFlowControlConfig config;
	RecvLeg *rleg; // part of a stream's fused receiving side (filters/recv_leg.inl): the controller lives in that bank
};

void flowctl_init(MSFilter *f) { // :160-164
	FlowFilter *d = (FlowFilter *)ms_malloc0(sizeof(FlowFilter));
	d->slot = -1;
	d->config.strategy = MSAudioFlowControlSoft;
	d->config.silent_threshold = 0.02f;
	f->data = d;
}
void flowctl_release(FlowFilter *d) {
	if (!d->pool) return;
	HubLock lk(d->pool->hub);
	const size_t s = (size_t)d->slot;
	for (auto *v : {&d->pool->held[s], &d->pool->done[s]}) {
		for (mblk_t *m : *v) freemsg(m);
		v->clear();
	}
	d->pool->staged[s] = d->pool->ready[s] = 0;
	d->pool->req_drop[s] = d->pool->req_total[s] = 0;
	d->pool->later_drop[s] = d->pool->later_total[s] = 0;
	d->pool->release(d->slot);
	d->pool = nullptr;
	d->slot = -1;
}
bool flowctl_attach(MSFilter *f, FlowFilter *d) {
	if (d->pool && !d->pool->failed && d->pool->hub->ticker == f->ticker) return true;
	flowctl_release(d);
	HubLock lk(f);
	FlowPool *p = bank<FlowPool>("flowctl", 1, [&](int cap) { return new FlowPool(cap); });
	const int sl = p ? p->acquire(f) : -1;
	if (sl < 0) return false;
	note_slot(f);
	d->pool = p;
	d->slot = sl;
	const bool was_used = p->used[(size_t)sl] != 0, dflt = d->config.strategy != MSAudioFlowControlBasic && d->config.silent_threshold == 0.02f; // flowcontrol.c:37-41
	p->used[(size_t)sl] = 1;
	if (was_used) MI_MUST(mi_flowctl_reset(d->pool->fc, sl, 1));
	if (was_used || !dflt)
		MI_MUST(mi_flowctl_set_config(d->pool->fc, sl, 1, d->config.strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT,
		                              d->config.silent_threshold));
	return true;
}
void flowctl_preprocess(MSFilter *f) { // :166-169 ms_audio_flow_controller_reset
	FlowFilter *d = (FlowFilter *)f->data;
	(void)d;
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f); // (fused: reset with its slot there; else a slot of its own bank, reset, there or at its first block)
}
void flowctl_process(MSFilter *f) { // :171-183
	FlowFilter *d = (FlowFilter *)f->data;
	HubLock lk(f); // lock order everywhere: the hub first, the filter's own lock inside it
	if (d->rleg) { // (nothing arrives while the chain is fused: the blocks leave this filter's output with the flush)
		ms_queue_flush(f->inputs[0]);
		return;
	}
	ms_filter_lock(f);
	if (!flowctl_attach(f, d)) {
		ms_queue_flush(f->inputs[0]);
		ms_filter_unlock(f);
		return;
	}
	FlowPool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	mblk_t *m;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		const size_t n = msgdsize(m) / 2;
		if (n == 0 || n > (size_t)kFlowBlock || m->b_cont || p->staged[s] >= kMaxRounds) {
			// nothing to edit, or a shape the batch does not take (longer than 2048 samples, chained, a fifth block
			// within one tick): it passes unedited -- never lost
			if (p->staged[s] == 0 && p->ready[s] == 0) ms_queue_put(f->outputs[0], m);
			else { // keep the order: let what is staged go first
				p->flush();
				p->emit_all();
				ms_queue_put(f->outputs[0], m);
			}
			continue;
		}
		const size_t r = (size_t)p->staged[s];
		memcpy(p->h_in + (r * c + s) * kFlowBlock, m->b_rptr, n * 2);
		p->h_len[r * c + s] = (int32_t)n;
		p->held[s].push_back(m);
		p->staged[s]++;
	}
	if (p->staged[s]) request_flush(f);
	ms_filter_unlock(f);
}
void flowctl_postprocess(MSFilter *f) {
	FlowFilter *d = (FlowFilter *)f->data;
	facade_detached(f);
	if (d->rleg) recv_release(d->rleg, false);
	flowctl_release(d);
}
void flowctl_uninit(MSFilter *f) { // :188-191
	if (((FlowFilter *)f->data)->rleg) recv_release(((FlowFilter *)f->data)->rleg, false);
	flowctl_release((FlowFilter *)f->data);
	ms_free(f->data);
}
int flowctl_set_config(MSFilter *f, void *arg) { // :193-197
	FlowFilter *d = (FlowFilter *)f->data;
	HubLock lk(f);
	d->config = *(MSAudioFlowControlConfig *)arg;
	if (d->rleg) recv_flow_config(d->rleg, &d->config);
	if (d->pool)
		MI_MUST(mi_flowctl_set_config(d->pool->fc, d->slot, 1, d->config.strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT,
		                              d->config.silent_threshold));
	return 0;
}
int flowctl_drop(MSFilter *f, void *arg) { // :199-211; applied by the next launch at this point of the block sequence
	FlowFilter *d = (FlowFilter *)f->data;
	const MSAudioFlowControlDropEvent *ev = (const MSAudioFlowControlDropEvent *)arg;
	HubLock lk(f);
	ms_filter_lock(f);
	if (d->rleg)
		recv_flow_drop(d->rleg, (ev->drop_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000,
		               (ev->flow_control_interval_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000);
	if (d->pool && f->ticker && d->pool->work_waiting()) { // the last walk's blocks are still on their way here: behind them (FlowPool::flushed)
		FlowPool *p = d->pool;
		const size_t s = (size_t)d->slot;
		if (p->later_drop[s] == 0 && p->later_total[s] == 0 && p->req_drop[s] == 0 && p->req_total[s] == 0) {
			p->later_drop[s] = (ev->drop_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000;
			p->later_total[s] = (ev->flow_control_interval_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000;
			p->have_later = true;
		}
	} else {
		if (d->pool && d->pool->req_drop[(size_t)d->slot] == 0 && d->pool->req_total[(size_t)d->slot] == 0) {
			d->pool->req_drop[(size_t)d->slot] = (ev->drop_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000;
			d->pool->req_total[(size_t)d->slot] = (ev->flow_control_interval_ms * (uint32_t)d->samplerate * (uint32_t)d->nchannels) / 1000;
			d->pool->req_round[(size_t)d->slot] = d->pool->staged[(size_t)d->slot];
			d->pool->have_req = true;
		}
	}
	ms_filter_unlock(f);
	return 0;
}
int flowctl_set_sr(MSFilter *f, void *arg) {
	((FlowFilter *)f->data)->samplerate = *(int *)arg;
	return 0;
}
int flowctl_get_sr(MSFilter *f, void *arg) {
	*(int *)arg = ((FlowFilter *)f->data)->samplerate;
	return 0;
}
int flowctl_set_nch(MSFilter *f, void *arg) {
	((FlowFilter *)f->data)->nchannels = *(int *)arg;
	return 0;
}
int flowctl_get_nch(MSFilter *f, void *arg) {
	*(int *)arg = ((FlowFilter *)f->data)->nchannels;
	return 0;
}
MSFilterMethod flowctl_methods[] = {{MS_AUDIO_FLOW_CONTROL_SET_CONFIG, flowctl_set_config},
                                    {MS_AUDIO_FLOW_CONTROL_DROP, flowctl_drop},
                                    {MS_FILTER_SET_SAMPLE_RATE, flowctl_set_sr},
                                    {MS_FILTER_GET_SAMPLE_RATE, flowctl_get_sr},
                                    {MS_FILTER_SET_NCHANNELS, flowctl_set_nch},
                                    {MS_FILTER_GET_NCHANNELS, flowctl_get_nch},
                                    {0, NULL}};
