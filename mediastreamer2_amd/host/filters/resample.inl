// filters/resample.inl -- MSResample facade (src/audiofilters/msresample.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// =================================================================== resampler
struct ResamplePool : Pool {
	uint32_t in_rate, out_rate;
	int in_len, ostride;
	mi_resampler *r = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	int32_t *h_olen, *d_olen;
	uint8_t *h_run, *d_run;
	std::vector<int> staged, ready;
	ResamplePool(int cap, uint32_t ir, uint32_t orate) : in_rate(ir), out_rate(orate) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_resampler_create(hub->ctx, capacity, ir, orate, 3 /* SPEEX_RESAMPLER_QUALITY_VOIP */, &r));
		in_len = (int)(ir / 100);
		ostride = r ? (mi_resampler_out_capacity(r, in_len) + 7) & ~7 : 8;
		const size_t c = (size_t)capacity;
		h_in = pinned<int16_t>(kMaxRounds * c * in_len);
		h_out = pinned<int16_t>(kMaxRounds * c * ostride);
		h_olen = pinned<int32_t>(kMaxRounds * c);
		h_run = pinned<uint8_t>(kMaxRounds * c);
		d_in = devmem<int16_t>(c * in_len);
		d_out = devmem<int16_t>(c * ostride);
		d_olen = devmem<int32_t>(c);
		d_run = devmem<uint8_t>(c);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	~ResamplePool() override {
		if (r) mi_resampler_destroy(r);
	}
	bool enqueue() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity, u = (size_t)hi; // rows [0, hi) are all that was ever handed out
		int maxr = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r_ = 0; r_ < maxr; ++r_) {
			for (int s = 0; s < capacity; ++s) h_run[r_ * c + s] = s < hi && staged[(size_t)s] > r_ && !parked(s);
			MI_MUST(mi_copy_h2d_pinned(ctx, d_in, h_in + r_ * c * in_len, u * in_len * 2));
			MI_MUST(mi_copy_h2d_pinned(ctx, d_run, h_run + r_ * c, c));
			MI_MUST(mi_resampler_process_masked(r, d_in, in_len, in_len, d_out, ostride, d_olen, d_run));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_out + r_ * c * ostride, d_out, u * ostride * 2));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_olen + r_ * c, d_olen, u * 4));
		}
		return maxr > 0;
	}
	void finish() override {
		for (int s = 0; s < hi; ++s) {
			if (parked(s)) continue;
			ready[(size_t)s] = failed ? 0 : staged[(size_t)s]; // a failed launch delivers nothing (late event counted)
			staged[(size_t)s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void emit(MSFilter *f, int slot) override;
};

struct ResampleData { // ResampleData msresample.c:33-42
	MSBufferizer *bz;
	uint32_t ts;
	uint32_t input_rate, output_rate;
	int in_nchannels, out_nchannels;
	ResamplePool *pool;
	int slot;                 // first channel's slot (the one that emits)
	std::vector<int> *slots;  // one batch slot per input channel (speex keeps one state per channel too)
	FusedLeg *leg;            // the filter is the head of a fused call leg (filters/leg_chain.inl): its blocks go to that bank
	bool fuse_checked;        // looked for a conference to fuse with since the last attach
	// the speex handle lives as long as the filter (msresample.c:117-120: a detach keeps it): the running state of a mono filter
	// travels from a bank slot that is given up (a conference fused or un-fused) to the next one of the same rates
	std::vector<uint8_t> *kept;
	uint32_t kept_in, kept_out;
};

// the state of the mono filter's slot in resampler `r` (hub locked), before the slot is given up
void resample_keep_from(ResampleData *d, mi_resampler *r, int slot, uint32_t in_rate, uint32_t out_rate) {
	const int n = r ? mi_resampler_state_bytes(r) : 0;
	d->kept->clear();
	if (n <= 0 || d->in_nchannels != 1) return;
	d->kept->resize((size_t)n);
	if (mi_resampler_get_state(r, slot, d->kept->data(), d->kept->size()) != MI_OK) d->kept->clear();
	d->kept_in = in_rate, d->kept_out = out_rate;
}
// ... and into the slot the filter gets next.  whole_periods_only: the fused leg's launch up-samples whole output periods (its
// resampler never leaves phase (0, 0)): a state that stands elsewhere is not taken over
bool resample_restore_to(ResampleData *d, mi_resampler *r, int slot, uint32_t in_rate, uint32_t out_rate, bool whole_periods_only) {
	if (!r || d->kept->empty() || d->kept_in != in_rate || d->kept_out != out_rate || (int)d->kept->size() != mi_resampler_state_bytes(r)) return false;
	if (whole_periods_only) {
		int32_t pos[2];
		memcpy(pos, d->kept->data(), sizeof(pos));
		if (pos[0] != 0 || pos[1] != 0) return false;
	}
	const bool ok = mi_resampler_set_state(r, slot, d->kept->data(), d->kept->size()) == MI_OK;
	d->kept->clear();
	return ok;
}

// the same out of / into a buffer of mi_resampler_get_states / set_states (a whole conference's members in one round trip).
// restore: dst keeps what it holds -- a fresh stream's state, zeros -- when there is nothing to restore
void resample_keep_bytes(ResampleData *d, const uint8_t *src, size_t each, uint32_t in_rate, uint32_t out_rate) {
	d->kept->clear();
	if (each == 0 || d->in_nchannels != 1) return;
	d->kept->assign(src, src + each);
	d->kept_in = in_rate, d->kept_out = out_rate;
}
bool resample_restore_bytes(ResampleData *d, uint8_t *dst, size_t each, uint32_t in_rate, uint32_t out_rate, bool whole_periods_only) {
	if (d->kept->empty() || d->kept_in != in_rate || d->kept_out != out_rate || d->kept->size() != each) return false;
	if (whole_periods_only) {
		int32_t pos[2];
		memcpy(pos, d->kept->data(), sizeof(pos));
		if (pos[0] != 0 || pos[1] != 0) return false;
	}
	memcpy(dst, d->kept->data(), each);
	d->kept->clear();
	return true;
}

void resample_init(MSFilter *f) { // msresample.c:44-54,:62-80
	ResampleData *d = (ResampleData *)ms_malloc0(sizeof(*d));
	d->bz = ms_bufferizer_new();
	d->input_rate = 8000;
	d->output_rate = 16000;
	d->in_nchannels = d->out_nchannels = 1;
	d->slot = -1;
	d->slots = new std::vector<int>();
	d->kept = new std::vector<uint8_t>();
	f->data = d;
}

void resample_release(ResampleData *d) { // hub locked by the caller
	if (d->pool) {
		ResamplePool *p = d->pool;
		std::vector<int> sl = *d->slots;
		for (size_t i = 0; i < sl.size(); ++i) {
			p->staged[(size_t)sl[i]] = p->ready[(size_t)sl[i]] = 0;
			if (i + 1 < sl.size() || p->in_use > 1) { // the bank lives on: the slot's next owner starts from a fresh handle
				if (!p->failed && mi_resampler_reset(p->r, sl[i], 1) != MI_OK) p->failed = mi_failed("mi_resampler_reset");
			}
			p->release(sl[i]); // the last release of a bank destroys it
		}
	}
	d->slots->clear();
	d->pool = nullptr;
	d->slot = -1;
}

void resample_postprocess(MSFilter *f) { // detach: a fused conference goes back to its facades' own banks
	ResampleData *d = (ResampleData *)f->data;
	facade_detached(f);
	if (d->leg) leg_release(d->leg, false);
	d->fuse_checked = false;
}

void resample_uninit(MSFilter *f) {
	ResampleData *d = (ResampleData *)f->data;
	if (d->leg) leg_release(d->leg, false);
	{
		HubLock lk(f);
		resample_release(d);
	}
	ms_bufferizer_destroy(d->bz);
	delete d->slots;
	delete d->kept;
	ms_free(d);
}

// msresample.c:87-100: first input channel copied to every output channel
mblk_t *channel_adapt(int in_nch, int out_nch, mblk_t *im) {
	if (out_nch == in_nch) return im;
	const size_t n = msgdsize(im) / (2 * (size_t)in_nch);
	mblk_t *om = allocb(n * 2 * (size_t)out_nch, 0);
	const int16_t *s = (const int16_t *)im->b_rptr;
	int16_t *o = (int16_t *)om->b_wptr;
	for (size_t i = 0; i < n; ++i)
		for (int c = 0; c < out_nch; ++c) o[i * out_nch + c] = s[i * in_nch];
	om->b_wptr += n * 2 * (size_t)out_nch;
	mblk_meta_copy(im, om);
	freemsg(im);
	return om;
}

void resample_process(MSFilter *f) { // resample_process_ms2 msresample.c:122-179
	ResampleData *d = (ResampleData *)f->data;
	mblk_t *im;
	if (d->output_rate == d->input_rate) { // :126-135 pass-through
		while ((im = ms_queue_get(f->inputs[0])) != NULL)
			ms_queue_put(f->outputs[0], channel_adapt(d->in_nchannels, d->out_nchannels, im));
		return;
	}
	// lock order everywhere: the hub first, the filter's own lock inside it (the flush task calls process() with the hub held)
	if (d->pool && d->pool->hub->ticker != f->ticker) { // the filter moved to another ticker: its slots go back under the OLD hub's lock
		HubLock old(d->pool->hub);
		resample_release(d);
	}
	HubLock lk(f, d->leg ? leg_pool(d->leg) : static_cast<Pool *>(d->pool));
	ms_filter_lock(f);
	if (!d->leg && !d->fuse_checked && !d->pool && f->ticker) { // first block since the attach: is this the head of a leg of a conference?
		d->fuse_checked = true;
		if (MSFilter *mx = leg_find_mixer(f)) conf_try_fuse(mx);
		else leg_try_fuse_plain(f); // MSResample -> MSSpeexEC -> MSVolume -> anything else: an AudioStream's sending side (audiostream.c:1798-1810)
	}
	if (d->leg && leg_wants_out(d->leg)) leg_release(d->leg, true); // a member stopped qualifying (a method call): back to the facades' own banks
	if (d->leg) { // fused: the block goes into the leg's row of the conference's bank, nothing is emitted here
		leg_stage_mic(f, d);
		ms_filter_unlock(f);
		return;
	}
	const int nch = d->in_nchannels < 1 ? 1 : d->in_nchannels;
	if (d->pool && (d->pool->failed || d->pool->in_rate != d->input_rate || d->pool->out_rate != d->output_rate || (int)d->slots->size() != nch))
		resample_release(d); // rates / channels changed: the handle is re-created, history lost (:138-148)
	if (!d->pool) {
		const uint32_t ir = d->input_rate, orate = d->output_rate;
		d->pool = bank<ResamplePool>("resample:" + std::to_string(ir) + ":" + std::to_string(orate), nch,
		                             [&](int cap) { return new ResamplePool(cap, ir, orate); });
		for (int ch = 0; d->pool && ch < nch; ++ch) { // interleaved input: one state per channel, like speex_resampler_init(nb_channels)
			const int sl = d->pool->acquire(f);
			if (sl < 0) break;
			d->slots->push_back(sl);
			note_slot(f);
		}
		if (!d->pool || (int)d->slots->size() != nch) { // the device refused: this tick's audio is lost, counted, and retried next tick
			resample_release(d);
			g_late_events.fetch_add(1, std::memory_order_relaxed);
			ms_queue_flush(f->inputs[0]);
			ms_filter_unlock(f);
			return;
		}
		d->slot = (*d->slots)[0];
		if (nch == 1) resample_restore_to(d, d->pool->r, d->slot, d->input_rate, d->output_rate, false); // (a leg that left its fused batch carries on where it was)
	}
	ResamplePool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	// this tick's input, re-framed to 10 ms blocks (a streaming filter: the sample sequence is
	// independent of the blocking); the results are emitted by the flush task (ResamplePool::emit)
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	const size_t nbytes = (size_t)p->in_len * 2 * (size_t)nch;
	std::vector<int16_t> frame;
	while (ms_bufferizer_get_avail(d->bz) >= nbytes) {
		if (p->staged[s] >= kMaxRounds) { // a burst of more blocks than launch rounds: what is staged goes out now
			p->flush();
			p->emit_all();
		}
		const size_t round = (size_t)p->staged[s];
		if (nch == 1) {
			ms_bufferizer_read(d->bz, (uint8_t *)(p->h_in + (round * c + s) * p->in_len), nbytes);
		} else { // de-interleave into the channels' rows
			frame.resize((size_t)p->in_len * nch);
			ms_bufferizer_read(d->bz, (uint8_t *)frame.data(), nbytes);
			for (int ch = 0; ch < nch; ++ch) {
				int16_t *row = p->h_in + (round * c + (size_t)(*d->slots)[(size_t)ch]) * p->in_len;
				for (int i = 0; i < p->in_len; ++i) row[i] = frame[(size_t)i * nch + ch];
			}
		}
		for (int sl : *d->slots) p->staged[(size_t)sl]++;
	}
	if (p->staged[s]) request_flush(f);
	ms_filter_unlock(f);
}

void ResamplePool::emit(MSFilter *f, int slot) {
	ResampleData *d = (ResampleData *)f->data;
	if (slot != d->slot) return; // the other channels' slots are emitted together with the first
	const size_t c = (size_t)capacity, s = (size_t)slot;
	const int nch = (int)d->slots->size();
	for (int r = 0; r < ready[s]; ++r) {
		const int outlen = h_olen[r * c + s];
		mblk_t *om = allocb((size_t)outlen * 2 * (size_t)nch, 0);
		if (nch == 1) {
			memcpy(om->b_wptr, h_out + (r * c + s) * ostride, (size_t)outlen * 2);
		} else { // re-interleave (speex_resampler_process_interleaved_int's output layout)
			int16_t *o = (int16_t *)om->b_wptr;
			for (int ch = 0; ch < nch; ++ch) {
				const int16_t *row = h_out + (r * c + (size_t)(*d->slots)[(size_t)ch]) * ostride;
				for (int i = 0; i < outlen; ++i) o[(size_t)i * nch + ch] = row[i];
			}
		}
		om->b_wptr += (size_t)outlen * 2 * (size_t)nch;
		mblk_set_timestamp_info(om, d->ts); // msresample.c:168-169
		d->ts += (uint32_t)outlen;
		if (f->outputs[0]) ms_queue_put(f->outputs[0], channel_adapt(nch, d->out_nchannels, om));
		else freemsg(om);
	}
	for (int sl : *d->slots) ready[(size_t)sl] = 0;
}

int resample_set_sr(MSFilter *f, void *arg) { // :181-192
	ResampleData *d = (ResampleData *)f->data;
	HubLock lk(f); // hub first, the filter's lock inside it; d->leg is only read under the hub's lock (the ticker thread deletes legs under it)
	ms_filter_lock(f);
	if (d->input_rate != *(unsigned int *)arg) leg_disqualify(d->leg), leg_forwarder_changed(f);
	d->input_rate = *(unsigned int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_output_sr(MSFilter *f, void *arg) { // :194-205
	ResampleData *d = (ResampleData *)f->data;
	HubLock lk(f); // hub first, the filter's lock inside it; d->leg is only read under the hub's lock (the ticker thread deletes legs under it)
	ms_filter_lock(f);
	if (d->output_rate != *(unsigned int *)arg) leg_disqualify(d->leg), leg_forwarder_changed(f);
	d->output_rate = *(unsigned int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_in_nch(MSFilter *f, void *arg) {
	ResampleData *d = (ResampleData *)f->data;
	ms_filter_lock(f);
	if (d->in_nchannels != *(int *)arg) leg_forwarder_changed(f);
	d->in_nchannels = *(int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_out_nch(MSFilter *f, void *arg) {
	ResampleData *d = (ResampleData *)f->data;
	ms_filter_lock(f);
	if (d->out_nchannels != *(int *)arg) leg_forwarder_changed(f);
	d->out_nchannels = *(int *)arg;
	ms_filter_unlock(f);
	return 0;
}
MSFilterMethod resample_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, resample_set_sr},
                                     {MS_FILTER_SET_OUTPUT_SAMPLE_RATE, resample_set_output_sr},
                                     {MS_FILTER_SET_NCHANNELS, resample_set_in_nch},
                                     {MS_FILTER_SET_OUTPUT_NCHANNELS, resample_set_out_nch},
                                     {0, NULL}};
