// filters/codec.inl -- G.711 / L16 / MSChannelAdapter facades (src/audiofilters/alaw.c, ulaw.c, l16.c, chanadapt.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ============================================ codecs, channel adapter, flow control
// Element-wise stages.  Every block any instance stages during a tick is packed, 16-element aligned, into ONE flat
// buffer per (ticker, operation): a tick of any number of decoders is one copy in, one launch, one copy out.
enum MapOp { OP_ALAW_DEC, OP_ULAW_DEC, OP_ALAW_ENC, OP_ULAW_ENC, OP_L16_SWAP, OP_MONO_TO_STEREO, OP_STEREO_TO_MONO, OP_TWO_MONO_TO_STEREO, OP_COUNT };
struct MapOpInfo {
	size_t in_bpe, out_bpe; // bytes per element (code word / sample / frame) on either side
};
const MapOpInfo kMapOps[OP_COUNT] = {{1, 2}, {1, 2}, {2, 1}, {2, 1}, {2, 2}, {2, 4}, {4, 2}, {2, 4}};

struct MapBlock {
	size_t off, n; // element offset into the flat buffers, element count
	mblk_t *meta;  // the input block when its meta data travel with the samples (decoders, L16 decoder), else NULL
	uint32_t ts;   // timestamp the encoders stamp (alaw.c:84-85, l16.c:89-91)
	bool set_ts;
	mblk_t *prefix = nullptr; // G.711 encoders: codes of this packet that were already made (by a fused conference the encoder has left since)
};

struct MapPool : Pool {
	MapOp op;
	size_t cap, used = 0; // elements
	uint8_t *h_in, *h_in2 = nullptr, *h_out, *d_in, *d_in2 = nullptr, *d_out;
	std::vector<std::vector<MapBlock>> staged, ready;
	MapPool(int cap_slots, MapOp o) : op(o) {
		Building b(this, cap_slots);
		cap = (size_t)capacity * kMaxRounds * 1024; // elements; a pool that fills up flushes early (reserve())
		const MapOpInfo &k = kMapOps[op];
		h_in = pinned<uint8_t>(cap * k.in_bpe);
		h_out = pinned<uint8_t>(cap * k.out_bpe);
		d_in = devmem<uint8_t>(cap * k.in_bpe);
		d_out = devmem<uint8_t>(cap * k.out_bpe);
		if (op == OP_TWO_MONO_TO_STEREO) {
			h_in2 = pinned<uint8_t>(cap * k.in_bpe);
			d_in2 = devmem<uint8_t>(cap * k.in_bpe);
		}
		staged.resize((size_t)capacity);
		ready.resize((size_t)capacity);
	}
	// room for n elements of `slot`; the caller fills n * in_bpe bytes at the returned address (and at *second)
	uint8_t *reserve(int slot, size_t n, mblk_t *meta, bool set_ts, uint32_t ts, uint8_t **second = nullptr) {
		const size_t need = (n + 15) & ~(size_t)15;
		if (need > cap) return nullptr;
		if (used + need > cap) { // full: what is staged goes out now, one tick early
			flush();
			emit_all();
		}
		staged[(size_t)slot].push_back(MapBlock{used, n, meta, ts, set_ts});
		uint8_t *p = h_in + used * kMapOps[op].in_bpe;
		if (second) *second = h_in2 + used * kMapOps[op].in_bpe;
		used += need;
		return p;
	}
	// (in two halves, so that a hub's flush waits ONCE for all of its banks: Pool::enqueue / finish)
	bool enqueue() override {
		if (used && !failed) {
			mi_ctx *ctx = hub->ctx;
			const MapOpInfo &k = kMapOps[op];
			MI_MUST(mi_copy_h2d_pinned(ctx, d_in, h_in, used * k.in_bpe));
			if (d_in2) MI_MUST(mi_copy_h2d_pinned(ctx, d_in2, h_in2, used * k.in_bpe));
			switch (op) {
			case OP_ALAW_DEC:
			case OP_ULAW_DEC:
				MI_MUST(mi_g711_decode(ctx, op == OP_ALAW_DEC ? MI_LAW_PCMA : MI_LAW_PCMU, d_in, used, (int16_t *)d_out, used, nullptr, (int)used, 1));
				break;
			case OP_ALAW_ENC:
			case OP_ULAW_ENC:
				MI_MUST(mi_g711_encode(ctx, op == OP_ALAW_ENC ? MI_LAW_PCMA : MI_LAW_PCMU, (const int16_t *)d_in, used, d_out, used, nullptr, (int)used, 1));
				break;
			case OP_L16_SWAP:
				MI_MUST(mi_l16_swap(ctx, (const int16_t *)d_in, (int16_t *)d_out, used));
				break;
			case OP_MONO_TO_STEREO:
				MI_MUST(mi_chan_adapt(ctx, MI_CHAN_MONO_TO_STEREO, (const int16_t *)d_in, nullptr, (int16_t *)d_out, used));
				break;
			case OP_STEREO_TO_MONO:
				MI_MUST(mi_chan_adapt(ctx, MI_CHAN_STEREO_TO_MONO, (const int16_t *)d_in, nullptr, (int16_t *)d_out, used));
				break;
			case OP_TWO_MONO_TO_STEREO:
				MI_MUST(mi_chan_adapt(ctx, MI_CHAN_TWO_MONO_TO_STEREO, (const int16_t *)d_in, (const int16_t *)d_in2, (int16_t *)d_out, used));
				break;
			default:
				break;
			}
			MI_MUST(mi_copy_d2h_pinned(ctx, h_out, d_out, used * k.out_bpe));
			return true;
		}
		return false;
	}
	void finish() override {
		bool kept = false; // a detaching graph's slots alone leave: the others' blocks stay where they lie, converted again by their own flush (a pure map)
		for (int s = 0; s < hi; ++s) {
			auto &st = staged[(size_t)s], &rd = ready[(size_t)s];
			if (parked(s)) {
				kept |= !st.empty();
				continue;
			}
			if (failed) { // nothing was converted: the blocks are lost (counted), their meta blocks freed
				for (MapBlock &b : st)
					if (b.meta) freemsg(b.meta);
			} else {
				rd.insert(rd.end(), st.begin(), st.end());
			}
			st.clear();
		}
		if (!kept) used = 0;
	}
	bool scoped() const override { return true; }
	void emit(MSFilter *f, int slot) override {
		const size_t bpe = kMapOps[op].out_bpe;
		for (const MapBlock &b : ready[(size_t)slot]) {
			const size_t pre = b.prefix ? (size_t)(b.prefix->b_wptr - b.prefix->b_rptr) : 0;
			mblk_t *o = allocb(pre + b.n * bpe, 0);
			if (pre) {
				memcpy(o->b_wptr, b.prefix->b_rptr, pre);
				o->b_wptr += pre;
				freemsg(b.prefix);
			}
			memcpy(o->b_wptr, h_out + b.off * bpe, b.n * bpe);
			o->b_wptr += b.n * bpe;
			if (b.meta) {
				mblk_meta_copy(b.meta, o);
				freemsg(b.meta);
			}
			if (b.set_ts) mblk_set_timestamp_info(o, b.ts);
			if (f->outputs[0]) ms_queue_put(f->outputs[0], o);
			else freemsg(o);
		}
		ready[(size_t)slot].clear();
	}
	void drop_slot(int slot) {
		for (auto *v : {&staged[(size_t)slot], &ready[(size_t)slot]}) {
			for (MapBlock &b : *v) {
				if (b.meta) freemsg(b.meta);
				if (b.prefix) freemsg(b.prefix);
			}
			v->clear();
		}
	}
};

struct MapFilter { // AlawEncData alaw.c:25-30 / EncState l16.c:22-29 / AdapterState chanadapt.c:29-38, one shape for all
	MapPool *pool;
	int slot;
	MSBufferizer *bz; // encoders re-frame to ptime
	int law;          // 0 A-law, 1 mu-law
	int ptime, maxptime;
	uint32_t ts;
	int rate, nchannels, out_nchannels;
	size_t nbytes;      // L16 encoder packet size
	size_t buffer_size; // channel adapter, two-input mode: bytes per tick and side
	FlowBuf *side[2];
	// G.711 encoders behind a conference whose members live in a ServerBank (filters/server_leg.inl): the pin's mix is encoded in
	// that batch, the facade only packs the codes to its ptime -- `pending` is the packet being filled (it survives the conference
	// leaving its batch: the facade's own next packet starts with it)
	bool sleg;
	void *sleg_bank;
	void *sleg_leg; // a DECODER that heads a fused member: its ServerLeg
	mblk_t *pending;
	bool fuse_checked; // a decoder: looked for a conference to fuse with since the last attach
	// a G.711 decoder that heads the receiving side of an AudioStream (decoder -> MSGenericPLC -> MSAudioFlowControl,
	// filters/recv_leg.inl): its packets' code bytes go into that batch as they are
	RecvLeg *rleg;
	FusedLeg *fleg; // a G.711 encoder behind a fused sending leg's MSVolume (filters/leg_chain.inl): the leg's chunks are encoded in that batch
};
void server_encoder_gone(MSFilter *e); // server_leg.inl
void server_stage_codes(MSFilter *f, MapFilter *d);
Pool *server_pool_of_map(MapFilter *d);
MSFilter *leg_volume_sink(MSFilter *vol);

MapFilter *map_new(MSFilter *f) {
	MapFilter *d = (MapFilter *)ms_malloc0(sizeof(MapFilter));
	d->slot = -1;
	d->rate = 8000;
	d->nchannels = d->out_nchannels = 1;
	f->data = d;
	return d;
}

void map_release(MapFilter *d) {
	if (!d->pool) return;
	HubLock lk(d->pool->hub);
	d->pool->drop_slot(d->slot);
	d->pool->release(d->slot); // the last release of a bank destroys it
	d->pool = nullptr;
	d->slot = -1;
}

// a filter that was moved to another ticker gives its slot back under the OLD hub's lock before anything else
void map_rehome(MSFilter *f, MapFilter *d) {
	if (d->pool && d->pool->hub->ticker != f->ticker) map_release(d);
}

// the pool of (this ticker, op) and a slot in it; false when the pool is exhausted
bool map_attach(MSFilter *f, MapFilter *d, MapOp op) {
	if (d->pool && !d->pool->failed && d->pool->op == op && d->pool->hub->ticker == f->ticker) return true;
	map_release(d);
	HubLock lk(f); // the hub of the ticker the filter runs on now (callers hold it already: recursive)
	MapPool *p = bank<MapPool>("map:" + std::to_string((int)op), 1, [&](int cap) { return new MapPool(cap, op); });
	const int sl = p ? p->acquire(f) : -1;
	if (sl < 0) {
		g_late_events.fetch_add(1, std::memory_order_relaxed);
		return false;
	}
	note_slot(f);
	d->pool = p;
	d->slot = sl;
	return true;
}

void map_uninit(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	if (d->sleg) server_encoder_gone(f);
	if (d->rleg) recv_release(d->rleg, false);
	if (d->fleg) leg_release(d->fleg, false);
	if (d->pending) freemsg(d->pending);
	map_release(d);
	if (d->bz) ms_bufferizer_destroy(d->bz);
	ms_free(d);
}

// copies a (possibly chained) block's payload: what msgpullup(m, -1) would make contiguous (alaw.c:211)
void copy_payload(const mblk_t *m, uint8_t *dst) {
	for (; m; m = m->b_cont) {
		const size_t n = (size_t)(m->b_wptr - m->b_rptr);
		memcpy(dst, m->b_rptr, n);
		dst += n;
	}
}

// ---- G.711 decoders: alaw_dec_process alaw.c:208-221 (ulaw.c the same with Snack_Mulaw2Lin)
void g711_dec_init_a(MSFilter *f) { map_new(f)->law = 0; }
void g711_dec_init_u(MSFilter *f) { map_new(f)->law = 1; }
void g711_dec_postprocess(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	facade_detached(f);
	if (d->sleg) server_encoder_gone(f);
	if (d->rleg) recv_release(d->rleg, false);
	d->fuse_checked = false;
}
// (alaw.c has none) the attaching thread: is this the head of a stream's receiving chain?  (the last of the chain's filters to be
// preprocessed finds every ticker set and fuses it: msticker.c:163-166 runs the graph's preprocess calls one after the other)
void g711_dec_preprocess(MSFilter *f) {
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f);
}
void g711_dec_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	if (d->rleg && recv_wants_out(d->rleg)) recv_release(d->rleg, true); // a member stopped qualifying: the facades' own banks from this walk on
	if (d->rleg) { // the head of a stream's receiving side: the packet goes into the stream's row of that batch as it is
		HubLock lk(f, recv_pool(d->rleg));
		recv_stage_codes(f, d);
		return;
	}
	if (!d->sleg && !d->fuse_checked && f->ticker && f->inputs[0] && !ms_queue_empty(f->inputs[0])) {
		// the first packet since the attach: decoder -> MSVolume -> [in_resampler ->] a conference mixer whose members are all remote endpoints?
		d->fuse_checked = true;
		MSFilter *vol = f->outputs[0] ? f->outputs[0]->next.filter : NULL;
		MSFilter *mx = (vol && vol->desc == &ms_mi355x_volume_desc) ? leg_volume_sink(vol) : NULL;
		if (mx && mx->desc == &ms_mi355x_audio_mixer_desc) {
			HubLock lk(f);
			conf_try_fuse(mx);
		}
	}
	if (d->sleg) { // the head of a conference server's member: the packet goes into the conference's bank as it is
		HubLock lk(f, server_pool_of_map(d));
		server_stage_codes(f, d);
		return;
	}
	map_rehome(f, d);
	HubLock lk(f, d->pool);
	if (!map_attach(f, d, d->law ? OP_ULAW_DEC : OP_ALAW_DEC)) {
		ms_queue_flush(f->inputs[0]);
		return;
	}
	mblk_t *m;
	bool any = false;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		const size_t n = msgdsize(m);
		uint8_t *dst = n ? d->pool->reserve(d->slot, n, m, false, 0) : nullptr;
		if (!dst) { // empty (or absurdly large) packet: the reference emits an empty block for the former
			if (n == 0) {
				mblk_t *o = allocb(0, 0);
				mblk_meta_copy(m, o);
				ms_queue_put(f->outputs[0], o);
			} else ms_error("msmi355x plugin: %s: packet of %zu bytes refused", f->desc->name, n);
			freemsg(m);
			continue;
		}
		copy_payload(m, dst);
		any = true;
	}
	if (any) request_flush(f);
}

// ---- G.711 encoders: alaw_enc_process alaw.c:56-90
void g711_enc_new(MSFilter *f, int law) { // alaw_enc_data_new alaw.c:32-39
	MapFilter *d = map_new(f);
	d->law = law;
	d->bz = ms_bufferizer_new();
	d->ptime = 0;
	d->maxptime = std::min(MS_DEFAULT_MAX_PTIME, 140);
}
void g711_enc_init_a(MSFilter *f) { g711_enc_new(f, 0); }
void g711_enc_init_u(MSFilter *f) { g711_enc_new(f, 1); }
void g711_enc_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	int frame_per_packet = 2;
	if (d->ptime >= 10) frame_per_packet = d->ptime / 10;
	if (frame_per_packet <= 0) frame_per_packet = 1;
	if (frame_per_packet > 14) frame_per_packet = 14; // 140 ms max (:68-69)
	const size_t size_of_pcm = (size_t)160 * (size_t)frame_per_packet;
	map_rehome(f, d);
	HubLock lk(f, d->pool); // (a filter that holds a slot knows its hub through the bank: no registry look-up)
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	if (ms_bufferizer_get_avail(d->bz) < size_of_pcm) return;
	if (!map_attach(f, d, d->law ? OP_ULAW_ENC : OP_ALAW_ENC)) {
		ms_bufferizer_flush(d->bz);
		return;
	}
	while (ms_bufferizer_get_avail(d->bz) >= size_of_pcm) {
		// (a packet a fused conference had begun: its codes stand, the PCM of the rest completes it)
		const size_t have = d->pending ? std::min((size_t)(d->pending->b_wptr - d->pending->b_rptr), size_of_pcm / 2) : 0;
		const size_t need = size_of_pcm - 2 * have;
		uint8_t *dst = d->pool->reserve(d->slot, need / 2, nullptr, true, d->ts);
		if (!dst) break;
		ms_bufferizer_read(d->bz, dst, need);
		d->pool->staged[(size_t)d->slot].back().prefix = d->pending;
		d->pending = nullptr;
		d->ts += (uint32_t)(size_of_pcm / 2);
	}
	request_flush(f);
}
void g711_enc_postprocess(MSFilter *f) { // (alaw.c has none: the bufferizer -- here also the packet being filled -- outlives a detach)
	facade_detached(f);
	if (((MapFilter *)f->data)->sleg) server_encoder_gone(f);
	if (((MapFilter *)f->data)->fleg) leg_release(((MapFilter *)f->data)->fleg, false);
}

// "key=value" out of an fmtp line "a=1;key=value; b=2" (what oRTP's fmtp_get_value does for the callers in alaw.c:92-105)
bool fmtp_value(const char *fmtp, const char *key, char *out, size_t cap) {
	const size_t klen = strlen(key);
	for (const char *p = fmtp; p && *p;) {
		while (*p == ' ' || *p == ';' || *p == '\t') ++p;
		const char *end = strchr(p, ';');
		const size_t len = end ? (size_t)(end - p) : strlen(p);
		if (len > klen && strncmp(p, key, klen) == 0 && p[klen] == '=') {
			const size_t vlen = std::min(len - klen - 1, cap - 1);
			memcpy(out, p + klen + 1, vlen);
			out[vlen] = 0;
			return true;
		}
		p = end;
	}
	return false;
}
int g711_enc_add_fmtp(MSFilter *f, void *arg) { // alaw.c:92-105
	MapFilter *d = (MapFilter *)f->data;
	char tmp[30];
	if (fmtp_value((const char *)arg, "maxptime", tmp, sizeof(tmp))) d->maxptime = std::min(atoi(tmp), MS_DEFAULT_MAX_PTIME);
	if (fmtp_value((const char *)arg, "ptime", tmp, sizeof(tmp))) d->ptime = std::min(atoi(tmp), d->maxptime);
	return 0;
}
int g711_enc_add_attr(MSFilter *f, void *arg) { // alaw.c:107-140: the first of "ptime:10", "ptime:20", .. "ptime:140" found anywhere in the line
	MapFilter *d = (MapFilter *)f->data;
	for (int v = 10; v <= 140; v += 10) { // in this order, so "ptime:100" already matches "ptime:10", exactly as there
		char key[16];
		snprintf(key, sizeof(key), "ptime:%d", v);
		if (strstr((const char *)arg, key) != NULL) {
			d->ptime = v;
			break;
		}
	}
	return 0;
}
int g711_get_sr(MSFilter *, void *arg) {
	*(int *)arg = 8000;
	return 0;
}
int g711_get_nch(MSFilter *, void *arg) {
	*(int *)arg = 1;
	return 0;
}
int g711_have_plc(MSFilter *, void *arg) {
	*(int *)arg = 0;
	return 0;
}
int g711_get_ptime(MSFilter *f, void *arg) {
	*(int *)arg = ((MapFilter *)f->data)->ptime;
	return 0;
}
MSFilterMethod g711_enc_methods[] = {{MS_FILTER_ADD_ATTR, g711_enc_add_attr}, {MS_FILTER_ADD_FMTP, g711_enc_add_fmtp},
                                     {MS_FILTER_GET_NCHANNELS, g711_get_nch}, {MS_FILTER_GET_SAMPLE_RATE, g711_get_sr},
                                     {MS_AUDIO_ENCODER_GET_PTIME, g711_get_ptime}, {0, NULL}};
MSFilterMethod g711_dec_methods[] = {{MS_FILTER_GET_NCHANNELS, g711_get_nch}, {MS_FILTER_GET_SAMPLE_RATE, g711_get_sr},
                                     {MS_DECODER_HAVE_PLC, g711_have_plc}, {0, NULL}};

// ---- L16: enc_process l16.c:76-93, dec_process :192-199
void l16_enc_init(MSFilter *f) { // :31-39
	MapFilter *d = map_new(f);
	d->bz = ms_bufferizer_new();
	d->ptime = 10;
}
void l16_enc_update(MapFilter *d) { d->nbytes = (size_t)((2 * d->nchannels * d->rate * d->ptime) / 1000); } // :48-50
void l16_enc_preprocess(MSFilter *f) {
	l16_enc_update((MapFilter *)f->data);
	generic_preprocess(f);
}
void l16_enc_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	map_rehome(f, d);
	HubLock lk(f); // lock order everywhere: the hub first, the filter's own lock inside it
	ms_filter_lock(f);
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	if (d->nbytes >= 2 && ms_bufferizer_get_avail(d->bz) >= d->nbytes) {
		if (!map_attach(f, d, OP_L16_SWAP)) ms_bufferizer_flush(d->bz);
		while (d->pool && ms_bufferizer_get_avail(d->bz) >= d->nbytes) {
			uint8_t *dst = d->pool->reserve(d->slot, d->nbytes / 2, nullptr, true, d->ts);
			if (!dst) break;
			ms_bufferizer_read(d->bz, dst, d->nbytes);
			d->ts += (uint32_t)(d->nbytes / (2 * (size_t)d->nchannels));
		}
		request_flush(f);
	}
	ms_filter_unlock(f);
}
void l16_set_ptime(MapFilter *d, int value) { // :95-101
	if (value > 0 && value <= 100) {
		d->ptime = value;
		l16_enc_update(d);
	}
}
int l16_enc_add_attr(MSFilter *f, void *arg) { // :103-112 (reads the number right after the first six characters, as there)
	const char *fmtp = (const char *)arg;
	if (strstr(fmtp, "ptime:")) {
		ms_filter_lock(f);
		l16_set_ptime((MapFilter *)f->data, atoi(fmtp + 6));
		ms_filter_unlock(f);
	}
	return 0;
}
int l16_enc_add_fmtp(MSFilter *f, void *arg) { // :114-124
	char tmp[16] = {0};
	if (fmtp_value((const char *)arg, "ptime", tmp, sizeof(tmp))) {
		ms_filter_lock(f);
		l16_set_ptime((MapFilter *)f->data, atoi(tmp));
		ms_filter_unlock(f);
	}
	return 0;
}
void l16_dec_init(MSFilter *f) { map_new(f); }
void l16_dec_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	map_rehome(f, d);
	HubLock lk(f);
	if (!map_attach(f, d, OP_L16_SWAP)) {
		ms_queue_flush(f->inputs[0]);
		return;
	}
	mblk_t *m;
	bool any = false;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		const size_t n = msgdsize(m) / 2;
		uint8_t *dst = n ? d->pool->reserve(d->slot, n, m, false, 0) : nullptr;
		if (!dst) {
			if (n == 0) ms_queue_put(f->outputs[0], m);
			else freemsg(m);
			continue;
		}
		std::vector<uint8_t> flat(msgdsize(m));
		copy_payload(m, flat.data());
		memcpy(dst, flat.data(), n * 2);
		any = true;
	}
	if (any) request_flush(f);
}
int map_set_sr(MSFilter *f, void *arg) {
	((MapFilter *)f->data)->rate = *(int *)arg;
	return 0;
}
int map_get_sr(MSFilter *f, void *arg) {
	*(int *)arg = ((MapFilter *)f->data)->rate;
	return 0;
}
int map_set_nch(MSFilter *f, void *arg) {
	((MapFilter *)f->data)->nchannels = *(int *)arg;
	return 0;
}
int map_get_nch(MSFilter *f, void *arg) {
	*(int *)arg = ((MapFilter *)f->data)->nchannels;
	return 0;
}
MSFilterMethod l16_enc_methods[] = {{MS_FILTER_ADD_ATTR, l16_enc_add_attr},   {MS_FILTER_ADD_FMTP, l16_enc_add_fmtp},
                                    {MS_FILTER_SET_SAMPLE_RATE, map_set_sr},  {MS_FILTER_SET_NCHANNELS, map_set_nch},
                                    {MS_FILTER_GET_SAMPLE_RATE, map_get_sr},  {MS_FILTER_GET_NCHANNELS, map_get_nch},
                                    {0, NULL}};
MSFilterMethod l16_dec_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, map_set_sr}, {MS_FILTER_GET_SAMPLE_RATE, map_get_sr},
                                    {MS_FILTER_GET_NCHANNELS, map_get_nch},  {MS_FILTER_SET_NCHANNELS, map_set_nch},
                                    {0, NULL}};

// ---- MSChannelAdapter chanadapt.c
void adapter_init(MSFilter *f) { map_new(f); } // :40-46
void adapter_free_sides(MapFilter *d) {
	for (FlowBuf *&b : d->side)
		if (b) {
			ms_bufferizer_uninit(&b->base);
			ms_free(b);
			b = nullptr;
		}
}
void adapter_preprocess(MSFilter *f) { // :53-66; the two-input buffers are needed whenever both pins are linked
	MapFilter *d = (MapFilter *)f->data;
	if ((f->inputs[0] && f->inputs[1]) || (d->nchannels == 2 && d->out_nchannels == 1)) {
		d->buffer_size = (size_t)((f->ticker->interval * d->rate) / 1000) * 2;
		for (FlowBuf *&b : d->side) {
			b = (FlowBuf *)ms_malloc0(sizeof(FlowBuf));
			flowbuf_init(b, f, d->rate);
			b->immediate_drop = true;
			b->max_size_ms = (uint32_t)f->ticker->interval * 2;
		}
	}
	generic_preprocess(f);
}
void adapter_postprocess(MSFilter *f) { // :125-135
	MapFilter *d = (MapFilter *)f->data;
	facade_detached(f);
	adapter_free_sides(d);
	map_release(d);
}
void adapter_uninit(MSFilter *f) {
	adapter_free_sides((MapFilter *)f->data);
	map_uninit(f);
}
void adapter_two_inputs(MSFilter *f, MapFilter *d) { // adapter_process_2_inputs_to_single_stereo_output :68-93
	flowbuf_put(d->side[0], nullptr, f->inputs[0]);
	flowbuf_put(d->side[1], nullptr, f->inputs[1]);
	const size_t a = ms_bufferizer_get_avail(&d->side[0]->base), b = ms_bufferizer_get_avail(&d->side[1]->base);
	if (d->buffer_size == 0 || (a < d->buffer_size && b < d->buffer_size)) return;
	if (!map_attach(f, d, OP_TWO_MONO_TO_STEREO)) return;
	uint8_t *second = nullptr;
	uint8_t *first = d->pool->reserve(d->slot, d->buffer_size / 2, nullptr, false, 0, &second);
	if (!first) return;
	if (a < d->buffer_size) memset(first, 0, d->buffer_size); // a short side is silent for the tick (:81-82)
	else ms_bufferizer_read(&d->side[0]->base, first, d->buffer_size);
	if (b < d->buffer_size) memset(second, 0, d->buffer_size);
	else ms_bufferizer_read(&d->side[1]->base, second, d->buffer_size);
	request_flush(f);
}
void adapter_process(MSFilter *f) { // :95-123
	MapFilter *d = (MapFilter *)f->data;
	map_rehome(f, d);
	HubLock lk(f);
	if (already_ran_this_tick(f)) return; // pumped by the flush task right behind the facades that feed it
	if (f->inputs[0] != NULL && f->inputs[1] != NULL && d->side[0]) {
		adapter_two_inputs(f, d);
		return;
	}
	mblk_t *im;
	bool any = false;
	while ((im = ms_queue_get(f->inputs[0])) != NULL) {
		if (d->nchannels == d->out_nchannels) {
			ms_queue_put(f->outputs[0], im);
			continue;
		}
		const bool widen = d->out_nchannels == 2;
		if (!widen && d->nchannels != 2) { // neither branch of the reference takes it: the block is leaked there, freed here
			freemsg(im);
			continue;
		}
		const size_t frames = msgdsize(im) / (widen ? 2 : 4);
		uint8_t *dst = (frames && map_attach(f, d, widen ? OP_MONO_TO_STEREO : OP_STEREO_TO_MONO))
		                   ? d->pool->reserve(d->slot, frames, nullptr, false, 0)
		                   : nullptr;
		if (dst) {
			std::vector<uint8_t> flat(msgdsize(im));
			copy_payload(im, flat.data());
			memcpy(dst, flat.data(), frames * (widen ? 2 : 4));
			any = true;
		}
		freemsg(im); // no meta data cross this filter (:108-121 allocate a bare block)
	}
	if (any) request_flush(f);
}
int adapter_set_out_nch(MSFilter *f, void *arg) {
	((MapFilter *)f->data)->out_nchannels = *(int *)arg;
	return 0;
}
int adapter_get_out_nch(MSFilter *f, void *arg) {
	*(int *)arg = ((MapFilter *)f->data)->out_nchannels;
	return 0;
}
MSFilterMethod adapter_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, map_set_sr},
                                    {MS_FILTER_GET_SAMPLE_RATE, map_get_sr},
                                    {MS_FILTER_SET_NCHANNELS, map_set_nch},
                                    {MS_FILTER_GET_NCHANNELS, map_get_nch},
                                    {MS_CHANNEL_ADAPTER_SET_OUTPUT_NCHANNELS, adapter_set_out_nch},
                                    {MS_CHANNEL_ADAPTER_GET_OUTPUT_NCHANNELS, adapter_get_out_nch},
                                    {0, NULL}};
