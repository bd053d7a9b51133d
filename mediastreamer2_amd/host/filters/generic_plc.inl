// filters/generic_plc.inl -- MSGenericPLC facade (src/audiofilters/msgenericplc.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ---- MSGenericPLC msgenericplc.c (build without bcg729: comfort noise is silence)
struct Concealer { // MSConcealerContext, src/base/mscommon.c:315-366
	int64_t sample_time = -1, plc_start_time = -1;
	unsigned long total_number_for_plc = 0;
	uint32_t max_plc_time = UINT32_MAX; // MAX_PLC_COUNT msgenericplc.c:43
	void inc_sample_time(uint64_t now, uint32_t increment, bool got_packet) { // :328-343
		if (sample_time == -1) sample_time = (int64_t)now;
		sample_time += increment;
		if (plc_start_time != -1 && got_packet) plc_start_time = -1;
	}
	bool required(uint64_t now) { // :345-366
		if (sample_time == -1) return false;
		if ((uint64_t)sample_time <= now) {
			if (plc_start_time == -1) plc_start_time = sample_time;
			const uint32_t plc_duration = (uint32_t)(now - (uint64_t)plc_start_time);
			if (plc_duration < max_plc_time) {
				total_number_for_plc++;
				return true;
			}
			sample_time = -1;
		}
		return false;
	}
};

constexpr int kPlcBlock = 1920; // samples per staged piece (mi_plc's LDS budget at 48 kHz); longer blocks are cut
struct PlcEntry {
	int kind;   // MI_PLC_RECEIVED (mblk edited in place), MI_PLC_CONCEAL (new block, plc flag), 0 = host-made comfort-noise block
	int round;  // launch round of a GPU entry
	int n;      // samples
	size_t off; // RECEIVED: sample offset of this piece inside its block
	bool last;  // RECEIVED: the block's last piece: forward it
	mblk_t *m;
};
struct PlcPool : Pool {
	int rate;
	mi_plc *plc = nullptr;
	int16_t *h_rows, *d_rows;
	int32_t *h_len, *d_len, *h_lensc;   // (h_lensc / h_modesc: the rounds' rows while a detaching graph's slots alone are flushed)
	uint8_t *h_mode, *d_mode, *h_modesc;
	std::vector<int> staged;
	std::vector<uint8_t> used; // the slot has had a filter since the bank was created (a fresh one IS a fresh context: mi_plc_create)
	std::vector<std::vector<PlcEntry>> pending, done;
	PlcPool(int cap, int r) : rate(r) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_plc_create(hub->ctx, capacity, rate, kPlcBlock, &plc));
		const size_t c = (size_t)capacity;
		h_rows = pinned<int16_t>(kMaxRounds * c * kPlcBlock);
		h_len = pinned<int32_t>(kMaxRounds * c);
		h_mode = pinned<uint8_t>(kMaxRounds * c);
		h_lensc = pinned<int32_t>(kMaxRounds * c);
		h_modesc = pinned<uint8_t>(kMaxRounds * c);
		d_rows = devmem<int16_t>(c * kPlcBlock);
		d_len = devmem<int32_t>(c);
		d_mode = devmem<uint8_t>(c);
		staged.assign(c, 0);
		used.assign(c, 0);
		pending.resize(c);
		done.resize(c);
	}
	int16_t *stage(int slot, int mode, int n) { // a launch round for `slot`; returns its host row
		const size_t c = (size_t)capacity, s = (size_t)slot;
		if (staged[s] >= kMaxRounds) { // a fifth piece within one tick: what is staged goes out now
			flush();
			emit_all();
		}
		const size_t r = (size_t)staged[s]++;
		h_len[r * c + s] = n;
		h_mode[r * c + s] = (uint8_t)mode;
		return h_rows + (r * c + s) * kPlcBlock;
	}
	~PlcPool() override {
		if (plc) mi_plc_destroy(plc);
	}
	void flush() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity;
		int maxr = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxr = std::max(maxr, staged[(size_t)s]);
		if (failed) { // a broken context is not given more work: received blocks pass as they came, a concealment is silence
			for (int r = 0; r < maxr; ++r)
				for (int s = 0; s < hi; ++s)
					if (!parked(s) && staged[(size_t)s] > r && h_mode[r * c + s] == MI_PLC_CONCEAL) memset(h_rows + (r * c + s) * kPlcBlock, 0, (size_t)kPlcBlock * 2);
			maxr = 0;
		}
		for (int r = 0; r < maxr; ++r) {
			const int32_t *lrow = h_len + r * c;
			const uint8_t *mrow = h_mode + r * c;
			if (hub->scope) {
				for (int s = 0; s < capacity; ++s) {
					const bool in = s < hi && staged[(size_t)s] > r && !parked(s);
					h_lensc[r * c + s] = in ? h_len[r * c + s] : 0;
					h_modesc[r * c + s] = in ? h_mode[r * c + s] : (uint8_t)MI_PLC_NONE;
				}
				lrow = h_lensc + r * c, mrow = h_modesc + r * c;
			} else {
				for (int s = 0; s < capacity; ++s)
					if (staged[(size_t)s] <= r) h_mode[r * c + s] = MI_PLC_NONE, h_len[r * c + s] = 0;
			}
			if (zero_copy_rows()) { // the launch works on the pieces where they lie in pinned memory: what crosses PCIe is the audio, not 1920 samples of row per stream
				MI_MUST(mi_plc_process(plc, h_rows + r * c * kPlcBlock, kPlcBlock, lrow, mrow));
			} else {
				MI_MUST(mi_copy_h2d_pinned(ctx, d_rows, h_rows + r * c * kPlcBlock, c * kPlcBlock * 2));
				MI_MUST(mi_copy_h2d_pinned(ctx, d_len, lrow, c * 4));
				MI_MUST(mi_copy_h2d_pinned(ctx, d_mode, mrow, c));
				MI_MUST(mi_plc_process(plc, d_rows, kPlcBlock, d_len, d_mode));
				MI_MUST(mi_copy_d2h_pinned(ctx, h_rows + r * c * kPlcBlock, d_rows, c * kPlcBlock * 2));
			}
		}
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			if (s < hi && parked(s)) continue;
			auto &p = pending[(size_t)s], &d = done[(size_t)s];
			d.insert(d.end(), p.begin(), p.end());
			p.clear();
			staged[(size_t)s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void emit(MSFilter *f, int slot) override {
		const size_t c = (size_t)capacity, s = (size_t)slot;
		for (const PlcEntry &e : done[s]) {
			const int16_t *row = h_rows + ((size_t)e.round * c + s) * kPlcBlock;
			mblk_t *m = e.m;
			if (e.kind == MI_PLC_RECEIVED) {
				memcpy(m->b_rptr + e.off * 2, row, (size_t)e.n * 2);
				if (!e.last) continue;
			} else if (e.kind == MI_PLC_CONCEAL) {
				memcpy(m->b_wptr, row, (size_t)e.n * 2);
				m->b_wptr += (size_t)e.n * 2;
			}
			if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
			else freemsg(m);
		}
		done[s].clear();
	}
};

struct PlcFilter { // generic_plc_struct msgenericplc.c:30-41
	PlcPool *pool;
	int slot;
	Concealer *concealer;
	int rate, nchannels;
	bool cng_set, cng_running;
	RecvLeg *rleg; // part of a stream's fused receiving side (filters/recv_leg.inl): the concealer's decisions stay here, the context lives in that bank
};

void plc_init(MSFilter *f) { // :45-53
	PlcFilter *d = (PlcFilter *)ms_malloc0(sizeof(PlcFilter));
	d->slot = -1;
	d->concealer = new Concealer();
	d->nchannels = 1;
	f->data = d;
}
void plc_release(PlcFilter *d) {
	if (!d->pool) return;
	HubLock lk(d->pool->hub);
	const size_t s = (size_t)d->slot;
	for (auto *v : {&d->pool->pending[s], &d->pool->done[s]}) {
		for (PlcEntry &e : *v)
			if (e.m && (e.kind != MI_PLC_RECEIVED || e.last)) freemsg(e.m);
		v->clear();
	}
	d->pool->staged[s] = 0;
	d->pool->release(d->slot);
	d->pool = nullptr;
	d->slot = -1;
}
bool plc_attach(MSFilter *f, PlcFilter *d) { // generic_plc_preprocess :55-58: a context for the configured rate
	if (d->pool && !d->pool->failed && d->pool->hub->ticker == f->ticker && d->pool->rate == d->rate) return true;
	plc_release(d);
	HubLock lk(f);
	const int rate = d->rate;
	// a rate the kernel does not take fails the bank's constructor: no abort, the audio is forwarded without concealment
	PlcPool *p = bank<PlcPool>("plc:" + std::to_string(rate), 1, [&](int cap) { return new PlcPool(cap, rate); });
	const int sl = p ? p->acquire(f) : -1;
	if (sl < 0) return false;
	note_slot(f);
	d->pool = p;
	d->slot = sl;
	if (p->used[(size_t)sl]) MI_MUST(mi_plc_reset(d->pool->plc, sl, 1));
	p->used[(size_t)sl] = 1;
	return true;
}
void plc_preprocess(MSFilter *f) {
	PlcFilter *d = (PlcFilter *)f->data;
	(void)d;
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f); // (a filter that does not join a fused chain takes its own bank's slot there -- or, configured later, at its first block)
}
void plc_process(MSFilter *f) { // generic_plc_process :59-167
	PlcFilter *d = (PlcFilter *)f->data;
	// a member stopped qualifying: the chain leaves its batch before anything of this walk is staged (behind a decoder of ours that did
	// stage in this walk -- it runs first -- the chain leaves with the decoder's next packet instead)
	if (d->rleg && recv_wants_out(d->rleg) && recv_idle(d->rleg)) recv_release(d->rleg, true);
	HubLock lk(f, d->rleg ? recv_pool(d->rleg) : static_cast<Pool *>(d->pool)); // (a filter that holds a slot knows its hub through the bank: no registry look-up)
	if (already_ran_this_tick(f)) return; // pumped by the flush task right behind the decoder that feeds it
	if (d->rleg) { // the stream's receiving side is one batch: the decisions of this walk, on counts
		recv_plc_walk(f, d);
		return;
	}
	if (d->rate <= 0 || !plc_attach(f, d)) { // no usable context: the stream passes as it is
		mblk_t *m;
		while ((m = ms_queue_get(f->inputs[0])) != NULL) ms_queue_put(f->outputs[0], m);
		return;
	}
	PlcPool *p = d->pool;
	const size_t s = (size_t)d->slot;
	const int nch = d->nchannels < 1 ? 1 : d->nchannels;
	bool any = false;
	mblk_t *m;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		const size_t msg_size = msgdsize(m);
		const unsigned int time = (unsigned int)((1000 * msg_size) / ((size_t)d->rate * sizeof(int16_t) * (size_t)nch));
		d->concealer->inc_sample_time(f->ticker->time, time, true);
		const size_t total = msg_size / 2;
		if (total == 0 || m->b_cont) { // nothing to edit / a chained block: forwarded as it is, in order
			if (p->staged[s] || !p->pending[s].empty()) {
				p->flush();
				p->emit_all();
			}
			ms_queue_put(f->outputs[0], m);
			continue;
		}
		for (size_t off = 0; off < total; off += kPlcBlock) {
			const int n = (int)std::min<size_t>(kPlcBlock, total - off);
			const bool cng = off == 0 && d->cng_running; // resuming after comfort noise :76-89
			int16_t *row = p->stage(d->slot, MI_PLC_RECEIVED | (cng ? MI_PLC_CNG_RESUME : 0), n);
			memcpy(row, m->b_rptr + off * 2, (size_t)n * 2);
			p->pending[s].push_back(PlcEntry{MI_PLC_RECEIVED, p->staged[s] - 1, n, off, off + (size_t)n >= total, m});
		}
		if (d->cng_running) d->cng_running = d->cng_set = false;
		any = true;
	}
	if (d->concealer->required(f->ticker->time)) { // :117-166
		const int buff = d->rate * nch * f->ticker->interval / 1000; // samples
		if (d->cng_set || d->cng_running) { // comfort noise: a silent block flagged as such, no concealer involved
			mblk_t *o = allocb((size_t)buff * 2, 0);
			memset(o->b_wptr, 0, (size_t)buff * 2);
			o->b_wptr += (size_t)buff * 2;
			o->reserved2 |= 1u << 3; // mblk_set_cng_flag msqueue.h:116
			if (d->cng_set) {
				d->cng_set = false;
				d->cng_running = true;
			}
			p->pending[s].push_back(PlcEntry{0, 0, buff, 0, true, o});
			any = true;
		} else {
			for (int off = 0; off < buff; off += kPlcBlock) { // one block per piece when a tick is longer than a row
				const int n = std::min(kPlcBlock, buff - off);
				mblk_t *o = allocb((size_t)n * 2, 0);
				o->reserved2 |= 1u << 2; // mblk_set_plc_flag msqueue.h:113
				p->stage(d->slot, MI_PLC_CONCEAL, n);
				p->pending[s].push_back(PlcEntry{MI_PLC_CONCEAL, p->staged[s] - 1, n, 0, true, o});
			}
			any = true;
		}
		d->concealer->inc_sample_time(f->ticker->time, (uint32_t)f->ticker->interval, false);
	}
	if (any) request_flush(f);
}
void plc_postprocess(MSFilter *f) {
	PlcFilter *d = (PlcFilter *)f->data;
	facade_detached(f);
	if (d->rleg) recv_release(d->rleg, false);
	plc_release(d);
}
void plc_uninit(MSFilter *f) { // :169-178
	PlcFilter *d = (PlcFilter *)f->data;
	if (d->rleg) recv_release(d->rleg, false);
	plc_release(d);
	delete d->concealer;
	ms_free(d);
}
int plc_get_sr(MSFilter *f, void *arg) {
	*(int *)arg = ((PlcFilter *)f->data)->rate;
	return 0;
}
int plc_set_sr(MSFilter *f, void *arg) {
	PlcFilter *d = (PlcFilter *)f->data;
	HubLock lk(f);
	if (d->rleg && d->rate != *(int *)arg) recv_disqualify(d->rleg); // (the facade's context starts over at another rate too: plc_attach)
	d->rate = *(int *)arg;
	return 0;
}
int plc_set_nch(MSFilter *f, void *arg) {
	((PlcFilter *)f->data)->nchannels = *(int *)arg;
	return 0;
}
int plc_set_cn(MSFilter *f, void *arg) { // :196-201 (the noise description itself is only used with bcg729)
	((PlcFilter *)f->data)->cng_set = true;
	return 0;
}
MSFilterMethod plc_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, plc_set_sr},
                                {MS_FILTER_GET_SAMPLE_RATE, plc_get_sr},
                                {MS_FILTER_SET_NCHANNELS, plc_set_nch},
                                {MS_GENERIC_PLC_SET_CN, plc_set_cn},
                                {0, NULL}};
