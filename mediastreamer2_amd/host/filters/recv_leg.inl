// filters/recv_leg.inl -- the RECEIVING side of an AudioStream as one device-resident batch.
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the facades it joins); not
// compiled on its own.
//
// The reference plumbs what a call receives as   rtprecv -> decoder -> [local_mixer] -> MSGenericPLC -> MSAudioFlowControl ->
// dtmfgen -> volrecv -> [recv_tee] -> [spk_equalizer] -> ec pin 0 -> soundwrite   (src/voip/audiostream.c:1812-1832).  The first
// stretch -- MSAlawDec / MSUlawDec, a local_mixer that can only forward (one linked input: audiomixer.c:244-286), MSGenericPLC,
// MSAudioFlowControl -- is a chain of THIS plugin's facades; facade by facade that is three banks, three uploads and downloads and,
// because each hands its blocks to the next with the hub's flush, three flush rounds (waits) per ticker and tick.  dtmfgen behind
// them is the reference's own CPU filter: the audio has to be in host memory there.  So, when the chain's filters are all ours, on one
// ticker, freshly attached:
//
//   * one slot per STREAM in a RecvBank; the decoder (the chain's head) stages every packet's code bytes as they are (80 B per 10 ms)
//     in a pinned row, or -- no decoder of ours in front -- MSGenericPLC stages the PCM blocks it is handed;
//   * the PLC facade, a pump (msgenericplc.c:203), keeps the reference's host-side decisions and makes them IN the walk, on COUNTS:
//     ms_concealer_inc_sample_time per block staged in this walk, ms_concealer_context_is_concealement_required for the tick
//     (src/base/mscommon.c:328-366, msgenericplc.c:59-167) -- a lost packet is concealed in the tick it is missing in, as in the
//     reference (the facades one by one conceal a tick later: their PLC sees a walk's blocks with the next flush);
//   * at the END of the graph walk (every stream's PLC has run) the bank's launches leave: decode (g711_decode_kernel) -> conceal /
//     cross-fade (mi_plc_process on the decoded rows, device-resident) -> drop (mi_flowctl_process) -> 160 B of PCM per stream and 10 ms
//     into pinned rows; the next tick's flush hands them on, one block per packet / concealment with the packet's meta data, to
//     whatever follows the chain (dtmfgen): ONE wait for the whole receiving side, one tick of added latency in total.
//
// MS_AUDIO_FLOW_CONTROL_DROP (flowcontrol.c:199-211: MSSpeexEC's and the sound card's events) and SET_CONFIG reach the bank's slot;
// a drop request takes effect where it fell in the stream's block sequence, as in FlowPool.  The filters leave the batch at detach
// (the tick in flight is delivered first, filters.cpp: facade_detached) or when MSGenericPLC is given another rate in mid-call (its
// context then starts over, as the facade's does).  A packet longer than a row (60 ms of G.711) sends the stream back to its facades.
// MSMI355X_NO_FUSE=1: every facade on its own bank (what the tests compare against).

constexpr int kRecvBlock = 480;               // samples (or code bytes) per staged row: 60 ms at 8 kHz, 10 ms at 48 kHz
constexpr int kRecvEntries = kMaxRounds + 2;  // blocks a stream can have on their way in one flush (launch rounds + a comfort-noise block)

struct RecvEntry {
	uint8_t kind;  // MI_PLC_RECEIVED (a packet / block that arrived), MI_PLC_CONCEAL (generated), 0 = host-made comfort-noise block
	uint8_t round; // launch round of a device entry
	int32_t n;     // samples
	mblk_t *m;     // RECEIVED: the packet (its meta data travel with the samples, alaw.c:214) or, without a decoder, the block itself; 0: the block
};

struct RecvBank;
struct RecvLeg {
	RecvBank *bank;
	int slot;
	MSFilter *dec, *plc, *fc, *mixer, *tail; // dec / fc / mixer (a forwarding local_mixer that is looked through) may be NULL; tail: whose output the blocks leave on
	int staged = 0;   // device rounds staged since the last launch
	int counted = 0;  // entries the PLC's walk has accounted for (ms_concealer_inc_sample_time)
	int nent = 0, nout = 0;
	RecvEntry ent[kRecvEntries]; // staged since the last launch
	RecvEntry out[kRecvEntries]; // launched, to be handed on by the flush
	uint64_t walk_stamp = 0;     // 1 + the ticker tick in which the PLC last ran in the walk
	std::atomic<bool> unfuse_wanted{false};
};

struct RecvBank : Pool {
	int rate, law; // law: MI_LAW_PCMA / MI_LAW_PCMU, -1 = no decoder (MSGenericPLC heads the chain)
	bool with_fc;
	mi_plc *plc = nullptr;
	mi_flowctl *fc = nullptr;
	uint8_t *h_codes = nullptr, *d_codes = nullptr; // [kMaxRounds][cap][kRecvBlock] pinned / [cap][kRecvBlock]
	int16_t *h_pcm, *d_pcm, *d_out = nullptr;       // [kMaxRounds][cap][kRecvBlock] pinned: staged PCM (no decoder) and every round's results; [cap][kRecvBlock]
	int32_t *h_len, *h_lensc, *d_len;               // [kMaxRounds][cap]; the same while a detaching graph's slots alone are flushed; [cap]
	uint8_t *h_mode, *h_modesc, *d_mode;            // MI_PLC_* per round and stream
	int32_t *h_olen, *d_olen = nullptr;             // [kMaxRounds][cap]: samples MSAudioFlowControl left of the block (0: dropped)
	std::vector<RecvLeg *> legs;
	std::vector<uint8_t> used; // the slot has had a stream since the bank was created (a fresh one needs no reset: recv_try_fuse)
	// MS_AUDIO_FLOW_CONTROL_DROP requests since the last launch and how many staged rounds of the stream precede each (FlowPool)
	std::vector<uint32_t> req_drop, req_total, arm_drop, arm_total;
	std::vector<int> req_round;
	bool have_req = false;
	bool zero_copy = true, no_early = false;
	bool staged_since = false, outstanding = false, early = false, early_any = false;
	int launched_rounds = 0;
	int walked = 0;
	uint32_t walk_epoch = 0;
	uint64_t launches = 0;

	RecvBank(int cap, int r, int l, bool flow) : rate(r), law(l), with_fc(flow) {
		Building b(this, cap);
		const size_t c = (size_t)capacity;
		if (!failed) MI_MUST(mi_plc_create(hub->ctx, capacity, rate, kRecvBlock, &plc));
		if (!failed && with_fc) MI_MUST(mi_flowctl_create(hub->ctx, capacity, kRecvBlock, &fc));
		if (law >= 0) {
			h_codes = pinned<uint8_t>(kMaxRounds * c * kRecvBlock);
			d_codes = devmem<uint8_t>(c * kRecvBlock);
		}
		h_pcm = pinned<int16_t>(kMaxRounds * c * kRecvBlock);
		d_pcm = devmem<int16_t>(c * kRecvBlock);
		if (with_fc) d_out = devmem<int16_t>(c * kRecvBlock);
		h_len = pinned<int32_t>(kMaxRounds * c);
		h_lensc = pinned<int32_t>(kMaxRounds * c);
		d_len = devmem<int32_t>(c);
		h_mode = pinned<uint8_t>(kMaxRounds * c);
		h_modesc = pinned<uint8_t>(kMaxRounds * c);
		d_mode = devmem<uint8_t>(c);
		h_olen = pinned<int32_t>(kMaxRounds * c);
		if (with_fc) d_olen = devmem<int32_t>(c);
		legs.assign(c, nullptr);
		used.assign(c, 0);
		req_drop.assign(c, 0), req_total.assign(c, 0), arm_drop.assign(c, 0), arm_total.assign(c, 0);
		req_round.assign(c, 0);
		zero_copy = zero_copy_rows();
		no_early = getenv("MSMI355X_NO_EARLY_LAUNCH") != nullptr;
	}
	~RecvBank() override {
		for (RecvLeg *l : legs)
			if (l) {
				drop_entries(l);
				delete l;
			}
		if (hub->ctx) mi_ctx_sync(hub->ctx);
		if (plc) mi_plc_destroy(plc);
		if (fc) mi_flowctl_destroy(fc);
	}
	static void drop_entries(RecvLeg *l) {
		for (int i = 0; i < l->nent; ++i)
			if (l->ent[i].m) freemsg(l->ent[i].m);
		for (int i = 0; i < l->nout; ++i)
			if (l->out[i].m) freemsg(l->out[i].m);
		l->nent = l->nout = l->staged = l->counted = 0;
	}
	bool scoped() const override { return true; }

	// MS_AUDIO_FLOW_CONTROL_DROP requests that fell before round r of their stream (last: everything left) go to the device
	void arm(int r, bool last) {
		if (!have_req || !fc) return;
		bool any = false, left = false;
		for (int s = 0; s < capacity; ++s) {
			arm_drop[(size_t)s] = arm_total[(size_t)s] = 0;
			if (req_drop[(size_t)s] == 0 && req_total[(size_t)s] == 0) continue;
			if (s < hi && parked(s)) {
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
	}

	bool enqueue() override {
		bool any = false;
		const bool was_early = early;
		if (early) { // already out since the end of the last graph walk
			early = false;
			any = early_any;
		}
		if (!was_early || staged_since) any |= enqueue_now(); // (what was staged after an early launch -- a stream that joined the bank later in that walk, a PLC run by the flush -- goes out now)
		outstanding = false; // the hub waits for the stream right behind this
		return any;
	}
	bool enqueue_now() {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity, UL = (size_t)hi;
		if (outstanding) sync_stream(); // (rare: a second launch before the first was collected) the length rows are about to be rewritten
		staged_since = false;
		int rounds = 0;
		for (size_t s = 0; s < UL; ++s) {
			RecvLeg *leg = legs[s];
			const bool in = leg && !parked((int)s);
			if (leg && !in && leg->nent) staged_since = true; // (not this flush's business: it leaves with the ticker's own)
			const int st = in ? leg->staged : 0;
			rounds = std::max(rounds, st);
			if (hub->scope) { // a detaching graph's slots alone: everybody else counts as empty in THIS launch and keeps what it staged
				for (int r = 0; r < kMaxRounds; ++r) {
					h_lensc[(size_t)r * c + s] = r < st ? h_len[(size_t)r * c + s] : 0;
					h_modesc[(size_t)r * c + s] = r < st ? h_mode[(size_t)r * c + s] : (uint8_t)MI_PLC_NONE;
				}
			} else {
				for (int r = leg ? leg->staged : 0; r < kMaxRounds; ++r) h_len[(size_t)r * c + s] = 0, h_mode[(size_t)r * c + s] = MI_PLC_NONE;
			}
			if (!in) continue;
			// what the stream staged is on its way now: the flush that collects this launch hands it on
			for (int i = 0; i < leg->nent && leg->nout < kRecvEntries; ++i) leg->out[leg->nout++] = leg->ent[i];
			leg->nent = leg->staged = leg->counted = 0;
		}
		if (failed) { // a broken context is given no more work: received blocks pass as they came (undecoded: silence), a concealment is silence
			for (int r = 0; r < rounds; ++r)
				for (size_t s = 0; s < UL; ++s)
					if (legs[s] && !parked((int)s) && (law >= 0 || h_mode[(size_t)r * c + s] == MI_PLC_CONCEAL)) memset(h_pcm + ((size_t)r * c + s) * kRecvBlock, 0, (size_t)kRecvBlock * 2);
			for (int r = 0; r < rounds; ++r)
				for (size_t s = 0; s < UL; ++s) h_olen[(size_t)r * c + s] = h_len[(size_t)r * c + s];
			launched_rounds = std::max(launched_rounds, rounds);
			return false;
		}
		for (int r = 0; r < rounds; ++r) {
			const int32_t *lrow = (hub->scope ? h_lensc : h_len) + (size_t)r * c;
			const uint8_t *mrow = (hub->scope ? h_modesc : h_mode) + (size_t)r * c;
			int16_t *res = h_pcm + (size_t)r * c * kRecvBlock; // where this round's blocks end up
			const int32_t *dl = lrow;
			const uint8_t *dm = mrow;
			if (!zero_copy) {
				MI_MUST(mi_copy_h2d_pinned(ctx, d_len, lrow, c * 4));
				MI_MUST(mi_copy_h2d_pinned(ctx, d_mode, mrow, c));
				dl = d_len, dm = d_mode;
			}
			int16_t *work = d_pcm;
			if (law >= 0) { // alaw_dec_process alaw.c:208-221: the packet's code bytes -> PCM, on the device from here on
				const uint8_t *codes = h_codes + (size_t)r * c * kRecvBlock;
				if (!zero_copy) {
					MI_MUST(mi_copy_h2d_pinned(ctx, d_codes, codes, UL * kRecvBlock));
					codes = d_codes;
				}
				MI_MUST(mi_g711_decode(ctx, law, codes, kRecvBlock, d_pcm, kRecvBlock, dl, kRecvBlock, UL));
				++launches;
			} else if (zero_copy) {
				work = res; // the blocks are edited where they lie in pinned memory (as PlcPool does)
			} else {
				MI_MUST(mi_copy_h2d_pinned(ctx, d_pcm, res, UL * kRecvBlock * 2));
			}
			MI_MUST(mi_plc_process(plc, work, kRecvBlock, dl, dm));
			++launches;
			if (fc) {
				arm(r, false);
				int16_t *to = zero_copy ? res : d_out;
				int32_t *ol = zero_copy ? h_olen + (size_t)r * c : d_olen;
				MI_MUST(mi_flowctl_process(fc, work, kRecvBlock, dl, kRecvBlock, to, kRecvBlock, ol));
				++launches;
				if (!zero_copy) {
					MI_MUST(mi_copy_d2h_pinned(ctx, res, d_out, UL * kRecvBlock * 2));
					MI_MUST(mi_copy_d2h_pinned(ctx, h_olen + (size_t)r * c, d_olen, c * 4));
				}
			} else {
				if (work != res) MI_MUST(mi_copy_d2h_pinned(ctx, res, work, UL * kRecvBlock * 2));
				for (size_t s = 0; s < UL; ++s) h_olen[(size_t)r * c + s] = lrow[s];
			}
		}
		if (fc) arm(rounds, true);
		launched_rounds = std::max(launched_rounds, rounds);
		outstanding |= rounds > 0;
		return rounds > 0;
	}
	void finish() override {
		if (failed && launched_rounds) g_late_events.fetch_add(1, std::memory_order_relaxed);
		launched_rounds = 0;
	}
	// one block per packet (alaw.c:208-221: a new block with the packet's meta data, which MSGenericPLC edits in place and
	// MSAudioFlowControl shortens or drops) and per concealment (msgenericplc.c:150-156: flagged), in the stream's order
	void emit(MSFilter *, int slot) override {
		RecvLeg *leg = legs[(size_t)slot];
		if (!leg) return;
		const size_t c = (size_t)capacity, s = (size_t)slot;
		MSQueue *q = leg->tail->outputs[0];
		for (int i = 0; i < leg->nout; ++i) {
			const RecvEntry &e = leg->out[i];
			if (e.kind == 0) { // comfort noise made on the host: as it is
				if (q) ms_queue_put(q, e.m);
				else freemsg(e.m);
				continue;
			}
			const int left = h_olen[(size_t)e.round * c + s];
			if (left <= 0 || !q) { // dropped entirely (flowcontrol.c:118,131,139)
				if (e.m) freemsg(e.m);
				continue;
			}
			const int16_t *row = h_pcm + ((size_t)e.round * c + s) * kRecvBlock;
			mblk_t *o;
			if (e.kind == MI_PLC_RECEIVED && law < 0) { // the block itself, edited in place
				o = e.m;
				memcpy(o->b_rptr, row, (size_t)left * 2);
				o->b_wptr = o->b_rptr + (size_t)left * 2;
			} else {
				o = allocb((size_t)left * 2, 0);
				memcpy(o->b_wptr, row, (size_t)left * 2);
				o->b_wptr += (size_t)left * 2;
				if (e.kind == MI_PLC_CONCEAL) o->reserved2 |= 1u << 2; // mblk_set_plc_flag msqueue.h:113
				if (e.m) {
					mblk_meta_copy(e.m, o);
					freemsg(e.m);
				}
			}
			ms_queue_put(q, o);
		}
		leg->nout = 0;
	}
	// a graph is being detached between two ticks: rows staged in the last walk whose launches have not left go now
	void launch_staged() {
		if (failed || !staged_since) return;
		const bool more = enqueue_now();
		early_any = early ? (early_any || more) : more;
		early = true;
	}
	void deliver_in_flight(int slot) {
		if (outstanding || early) {
			sync_stream();
			outstanding = false;
		}
		emit(owner[(size_t)slot], slot);
	}
};

RecvLeg *recv_leg_of_dec(MapFilter *d) { return (RecvLeg *)d->rleg; }
Pool *recv_pool(RecvLeg *leg) { return leg->bank; }
bool recv_wants_out(RecvLeg *leg) { return leg && leg->unfuse_wanted.load(); }
bool recv_idle(RecvLeg *leg) { return leg->nent == 0; } // nothing staged since the last launch
void recv_disqualify(RecvLeg *leg) {
	if (leg) leg->unfuse_wanted = true;
}

// every stream of the bank has had its PLC run in this tick's walk: everything the tick will stage IS staged, the launches leave now
void recv_walked(RecvBank *b, RecvLeg *leg) {
	if (b->no_early || b->failed || b->early || !b->hub->ticker || b->hub->in_flush) return;
	const uint32_t tick = b->hub->ticker->ticks;
	if (b->walk_epoch != tick) b->walk_epoch = tick, b->walked = 0;
	if (leg->walk_stamp == (uint64_t)tick + 1) return;
	leg->walk_stamp = (uint64_t)tick + 1;
	if (++b->walked < b->in_use) return;
	b->early_any = b->enqueue_now();
	b->early = true;
}

// a row for one more block of the stream, or NULL when the launch rounds of this flush are taken
int recv_new_round(RecvBank *b, RecvLeg *leg, int mode, int n) {
	if (leg->staged >= kMaxRounds || leg->nent >= kRecvEntries) return -1;
	const size_t c = (size_t)b->capacity;
	const int r = leg->staged++;
	b->h_len[(size_t)r * c + (size_t)leg->slot] = n;
	b->h_mode[(size_t)r * c + (size_t)leg->slot] = (uint8_t)mode;
	return r;
}

// ---- the decoder's fused half (alaw_dec_process alaw.c:208-221): every packet's code bytes into a row, as they are
void recv_stage_codes(MSFilter *f, MapFilter *d) {
	RecvLeg *leg = recv_leg_of_dec(d);
	RecvBank *b = leg->bank;
	const size_t c = (size_t)b->capacity;
	if (!d->bz) d->bz = ms_bufferizer_new(); // (whole packets beyond a tick's launch rounds wait here, in order)
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) putq(&d->bz->q, m);
	bool any = false;
	for (mblk_t *m; (m = peekq(&d->bz->q)) != NULL;) {
		const size_t n = msgdsize(m);
		if (n > (size_t)kRecvBlock) { // longer than a row: the stream goes back to its facades (its next walk), this packet with it
			leg->unfuse_wanted = true;
			break;
		}
		if (n == 0) { // an empty packet makes an empty block (alaw.c:213-219), which the PLC counts and forwards
			if (leg->nent >= kRecvEntries) break;
			getq(&d->bz->q);
			mblk_t *o = allocb(0, 0);
			mblk_meta_copy(m, o);
			freemsg(m);
			leg->ent[leg->nent++] = RecvEntry{0, 0, 0, o};
			any = true;
			continue;
		}
		const int r = recv_new_round(b, leg, MI_PLC_RECEIVED, (int)n);
		if (r < 0) break; // (more packets than launch rounds in one tick: the rest next tick)
		getq(&d->bz->q);
		copy_payload(m, b->h_codes + ((size_t)r * c + (size_t)leg->slot) * kRecvBlock);
		leg->ent[leg->nent++] = RecvEntry{MI_PLC_RECEIVED, (uint8_t)r, (int32_t)n, m};
		any = true;
	}
	if (any) {
		b->staged_since = true;
		request_flush(f);
	}
}

// ---- MSGenericPLC's fused half: generic_plc_process msgenericplc.c:59-167 on counts
void recv_plc_walk(MSFilter *f, PlcFilter *d) {
	RecvLeg *leg = d->rleg;
	RecvBank *b = leg->bank;
	const size_t c = (size_t)b->capacity, s = (size_t)leg->slot;
	const int nch = d->nchannels < 1 ? 1 : d->nchannels;
	bool any = false;
	if (!leg->dec) { // the chain's head: the blocks it is handed are staged as PCM
		for (mblk_t *m; (m = peekq(&f->inputs[0]->q)) != NULL;) {
			const size_t total = msgdsize(m) / 2;
			if (total > (size_t)kRecvBlock || m->b_cont) {
				leg->unfuse_wanted = true; // (a shape the batch does not take: back to the facade, which cuts / forwards it)
				break;
			}
			if (total == 0) {
				if (leg->nent >= kRecvEntries) break;
				getq(&f->inputs[0]->q);
				leg->ent[leg->nent++] = RecvEntry{0, 0, 0, m};
				continue;
			}
			const int r = recv_new_round(b, leg, MI_PLC_RECEIVED, (int)total);
			if (r < 0) break; // (the rest stays on the queue: next tick)
			getq(&f->inputs[0]->q);
			memcpy(b->h_pcm + ((size_t)r * c + s) * kRecvBlock, m->b_rptr, total * 2);
			leg->ent[leg->nent++] = RecvEntry{MI_PLC_RECEIVED, (uint8_t)r, (int32_t)total, m};
		}
	}
	// :63-116 for every block of this walk: the concealer's clock advances by the block's duration; a block that ends comfort noise says so
	for (; leg->counted < leg->nent; ++leg->counted) {
		RecvEntry &e = leg->ent[leg->counted];
		const unsigned int time = (unsigned int)((1000 * (size_t)e.n * 2) / ((size_t)d->rate * sizeof(int16_t) * (size_t)nch));
		d->concealer->inc_sample_time(f->ticker->time, time, true);
		if (e.kind == MI_PLC_RECEIVED) {
			if (d->cng_running) {
				b->h_mode[(size_t)e.round * c + s] = MI_PLC_RECEIVED | MI_PLC_CNG_RESUME; // :76-89
				d->cng_running = d->cng_set = false;
			}
		}
		any = true;
	}
	if (d->concealer->required(f->ticker->time)) { // :117-166
		const int buff = d->rate * nch * f->ticker->interval / 1000; // samples
		if (d->cng_set || d->cng_running) { // comfort noise: a silent block flagged as such, no concealer involved
			if (leg->nent < kRecvEntries) {
				mblk_t *o = allocb((size_t)buff * 2, 0);
				memset(o->b_wptr, 0, (size_t)buff * 2);
				o->b_wptr += (size_t)buff * 2;
				o->reserved2 |= 1u << 3; // mblk_set_cng_flag msqueue.h:116
				leg->ent[leg->nent++] = RecvEntry{0, 0, buff, o};
				leg->counted = leg->nent;
			}
			if (d->cng_set) {
				d->cng_set = false;
				d->cng_running = true;
			}
			any = true;
		} else if (buff <= kRecvBlock) {
			const int r = recv_new_round(b, leg, MI_PLC_CONCEAL, buff);
			if (r >= 0) {
				leg->ent[leg->nent++] = RecvEntry{MI_PLC_CONCEAL, (uint8_t)r, buff, nullptr};
				leg->counted = leg->nent;
			} else g_late_events.fetch_add(1, std::memory_order_relaxed); // (the rounds are taken by a burst of packets: this concealment is skipped, counted)
			any = true;
		}
		d->concealer->inc_sample_time(f->ticker->time, (uint32_t)f->ticker->interval, false);
	}
	if (any || leg->nent) {
		b->staged_since = true;
		request_flush(f);
	}
	recv_walked(b, leg);
}

// ---- MS_AUDIO_FLOW_CONTROL_DROP / SET_CONFIG on a fused stream's MSAudioFlowControl (hub locked)
void recv_flow_drop(RecvLeg *leg, uint32_t drop, uint32_t total) {
	RecvBank *b = leg->bank;
	const size_t s = (size_t)leg->slot;
	if (!b->fc || b->req_drop[s] || b->req_total[s]) return; // (a request is ignored while one is pending, as in FlowPool)
	b->req_drop[s] = drop, b->req_total[s] = total;
	b->req_round[s] = leg->staged;
	b->have_req = true;
}
void recv_flow_config(RecvLeg *leg, const MSAudioFlowControlConfig *cfg) {
	RecvBank *b = leg->bank;
	if (b->fc && !b->failed)
		MI_MUST(mi_flowctl_set_config(b->fc, leg->slot, 1, cfg->strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT, cfg->silent_threshold));
}

// ---- fusing ------------------------------------------------------------------------------------------------------------
bool is_g711_dec(const MSFilterDesc *d); // server_leg.inl

// a local_mixer (audiostream.c:1770-1772,1815) that can only forward: ours, one linked input (pin 0), one output (pin 0), not a
// conference -- audiomixer.c:244-286 hands that input's blocks on as they are.  Looked through while the chain is fused.
bool is_forwarding_mixer(MSFilter *g, MSTicker *ticker) {
	if (!g || g->desc != &ms_mi355x_audio_mixer_desc || g->ticker != ticker) return false;
	const MixerState *ms = (const MixerState *)g->data;
	if (ms->conf_mode != 0 || ms->fbank || ms->sbank || !g->inputs[0] || !g->outputs[0] || !ms->channels[0].output_enabled || !ms->held->empty()) return false;
	for (int i = 1; i < g->desc->ninputs; ++i)
		if (g->inputs[i] || g->outputs[i]) return false;
	return ms_bufferizer_get_avail(const_cast<MSBufferizer *>(&ms->channels[0].bufferizer)) == 0;
}

void plc_release(PlcFilter *d);
void flowctl_release(FlowFilter *d);

// `head`: a G.711 decoder of ours, or an MSGenericPLC whose input is not one.  true = the chain is fused (hub locked by the caller)
bool recv_try_fuse(MSFilter *head) {
	if (getenv("MSMI355X_NO_FUSE") != nullptr || !head->ticker || head->ticker->interval != 10) return false;
	MSFilter *dec = is_g711_dec(head->desc) ? head : nullptr, *mixer = nullptr;
	MSFilter *plcf = head;
	if (dec) {
		MapFilter *dd = (MapFilter *)dec->data;
		if (dd->sleg || dd->rleg || !dec->outputs[0] || !ms_queue_empty(dec->outputs[0])) return false;
		if (dd->pool && (!dd->pool->staged[(size_t)dd->slot].empty() || !dd->pool->ready[(size_t)dd->slot].empty())) return false;
		plcf = dec->outputs[0]->next.filter;
		if (is_forwarding_mixer(plcf, head->ticker)) {
			mixer = plcf;
			if (!ms_queue_empty(mixer->outputs[0])) return false;
			plcf = mixer->outputs[0]->next.filter;
		}
	}
	if (!plcf || plcf->desc != &ms_mi355x_generic_plc_desc || plcf->ticker != head->ticker || !plcf->inputs[0] || !ms_queue_empty(plcf->inputs[0])) return false;
	PlcFilter *pd = (PlcFilter *)plcf->data;
	if (pd->rleg || pd->rate <= 0 || (dec && pd->rate != 8000) || !plcf->outputs[0]) return false;
	if (pd->pool && (pd->pool->staged[(size_t)pd->slot] || !pd->pool->pending[(size_t)pd->slot].empty() || !pd->pool->done[(size_t)pd->slot].empty())) return false;
	if (pd->rate / 100 > kRecvBlock) return false;
	MSFilter *fcf = plcf->outputs[0]->next.filter, *tail = plcf;
	FlowFilter *fd = nullptr;
	if (fcf && fcf->desc == &ms_mi355x_audio_flow_control_desc && fcf->ticker == head->ticker) {
		fd = (FlowFilter *)fcf->data;
		if (!ms_queue_empty(plcf->outputs[0])) return false; // (the queues BETWEEN the chain's filters must be empty; what waits behind its tail -- a tick delivered at a detach -- stays in front of what the batch will emit)
		if (fd->rleg || (fd->pool && (fd->pool->staged[(size_t)fd->slot] || fd->pool->ready[(size_t)fd->slot]))) return false;
		tail = fcf;
	} else fcf = nullptr;
	if (!dec && !fcf) return false; // (MSGenericPLC alone is what its own bank does)
	const int rate = pd->rate, law = dec ? (((MapFilter *)dec->data)->law ? MI_LAW_PCMU : MI_LAW_PCMA) : -1;
	const bool flow = fcf != nullptr;
	RecvBank *b = bank<RecvBank>("rcv:" + std::to_string(rate) + ":" + std::to_string(law) + (flow ? ":fc" : ""), 1,
	                             [&](int cap) { return new RecvBank(cap * 4, rate, law, flow); }); // 64, 256, 1024, .. streams
	const int s = b ? b->acquire(head) : -1;
	if (s < 0) return false;
	note_slot(head);
	// generic_plc_preprocess :55-58: a fresh context; flowcontrol.c:166-169: a controller at rest.  A slot nobody has used since the bank was created
	// IS there (mi_plc_create / mi_flowctl_create leave every stream so, with the default configuration): no device call for it
	const bool was_used = b->used[(size_t)s] != 0;
	b->used[(size_t)s] = 1;
	bool ok = !was_used || mi_plc_reset(b->plc, s, 1) == MI_OK;
	if (ok && b->fc) {
		const bool dflt = fd->config.strategy != MSAudioFlowControlBasic && fd->config.silent_threshold == 0.02f; // ms_audio_flow_controller_init flowcontrol.c:37-41
		if (was_used) ok = mi_flowctl_reset(b->fc, s, 1) == MI_OK;
		if (ok && (was_used || !dflt))
			ok = mi_flowctl_set_config(b->fc, s, 1, fd->config.strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT, fd->config.silent_threshold) == MI_OK;
	}
	if (!ok) {
		mi_failed("fusing a stream's receiving side");
		b->release(s);
		return false;
	}
	b->req_drop[(size_t)s] = b->req_total[(size_t)s] = 0;
	RecvLeg *leg = new RecvLeg();
	leg->bank = b, leg->slot = s;
	leg->dec = dec, leg->plc = plcf, leg->fc = fcf, leg->mixer = mixer, leg->tail = tail;
	b->legs[(size_t)s] = leg;
	// the facades let go of their own slots (on this hub, which is held)
	if (dec) {
		MapFilter *dd = (MapFilter *)dec->data;
		map_release(dd);
		dd->rleg = leg;
	}
	plc_release(pd);
	pd->rleg = leg;
	if (fd) {
		flowctl_release(fd);
		fd->rleg = leg;
	}
	b->staged_since = true;
	ms_message("mi355x: receiving side %p fused: %s%sMSGenericPLC%s at %d Hz as one device-resident batch (bank of %d)", (void *)head, dec ? (law == MI_LAW_PCMU ? "MSUlawDec -> " : "MSAlawDec -> ") : "",
	           mixer ? "(local_mixer) -> " : "", fcf ? " -> MSAudioFlowControl" : "", rate, b->capacity);
	return true;
}

// Any filter of a possible chain, from its preprocess (the attaching thread): the chain's head is looked up and the chain fused if it
// qualifies -- which needs every member's ticker set, so the LAST member to be preprocessed is the one that succeeds
MSFilter *recv_head_of(MSFilter *g) {
	for (int hops = 0; g && hops < 4; ++hops) {
		if (is_g711_dec(g->desc)) return g;
		MSFilter *up = g->inputs[0] ? g->inputs[0]->prev.filter : NULL;
		if (g->desc == &ms_mi355x_generic_plc_desc) {
			if (!up || !(is_g711_dec(up->desc) || up->desc == &ms_mi355x_audio_mixer_desc)) return g;
		} else if (g->desc == &ms_mi355x_audio_mixer_desc) {
			if (!up || !is_g711_dec(up->desc)) return NULL;
		} else if (g->desc != &ms_mi355x_audio_flow_control_desc) return NULL;
		g = up;
	}
	return NULL;
}
void recv_chain_preprocessed(MSFilter *member) {
	MSFilter *head = recv_head_of(member);
	if (!head) return;
	if (head->desc == &ms_mi355x_generic_plc_desc) {
		MSFilter *up = head->inputs[0] ? head->inputs[0]->prev.filter : NULL;
		if (up && up->desc == &ms_mi355x_audio_mixer_desc) return; // (a mixer that works -- two linked inputs -- delivers with the flush: MSGenericPLC heads the chain only behind somebody else's filter)
	}
	recv_try_fuse(head);
}
// The chain leaves its batch: at detach (keep_running false; the tick in flight was delivered by the graph's scoped flush, what is
// left is handed on here) or because a member stopped qualifying while attached -- the facades then carry on with banks of their own,
// the PLC and the flow controller starting over as they do at an attach.  Any of the chain's facades may call; the first does the work.
void recv_release(RecvLeg *leg, bool keep_running) {
	if (!leg) return;
	RecvBank *b = leg->bank;
	HubLock lk(b->hub);
	const int s = leg->slot;
	if (b->legs[(size_t)s] != leg) return;
	if (leg->nent && !b->failed) { // staged in a walk whose launch has not left (a bank without early launch): it leaves now
		b->staged_since = true;
		b->launch_staged();
	}
	b->deliver_in_flight(s);
	RecvBank::drop_entries(leg);
	b->legs[(size_t)s] = nullptr;
	b->req_drop[(size_t)s] = b->req_total[(size_t)s] = 0;
	if (leg->dec) {
		MapFilter *dd = (MapFilter *)leg->dec->data;
		dd->rleg = nullptr;
		if (keep_running && dd->bz) { // packets that waited for a launch round go back in front of the decoder's queue
			mblk_t *m;
			std::vector<mblk_t *> later;
			while ((m = ms_queue_get(leg->dec->inputs[0])) != NULL) later.push_back(m);
			while ((m = getq(&dd->bz->q)) != NULL) ms_queue_put(leg->dec->inputs[0], m);
			for (mblk_t *l : later) ms_queue_put(leg->dec->inputs[0], l);
		} else if (dd->bz) flushq(&dd->bz->q, 0);
	}
	((PlcFilter *)leg->plc->data)->rleg = nullptr;
	if (leg->fc) ((FlowFilter *)leg->fc->data)->rleg = nullptr;
	if (keep_running) ms_warning("mi355x: receiving side %p left its fused batch; the facades carry on one by one", (void *)b->owner[(size_t)s]);
	b->release(s); // (may destroy the bank)
	delete leg;
}

// a graph is being detached (facade_detached): its streams' tick in flight is launched if it has not left, waited for and handed on
void deliver_recv_in_scope(TickerHub &h) {
	for (Pool *p : h.pools) {
		if (p->key.compare(0, 4, "rcv:") != 0) continue;
		RecvBank *b = static_cast<RecvBank *>(p);
		bool ours = false;
		for (int s = 0; s < b->hi && !ours; ++s) ours = b->owner[(size_t)s] && h.scope->count(b->owner[(size_t)s]);
		if (!ours) continue;
		b->launch_staged();
		for (int s = 0; s < b->hi; ++s)
			if (b->owner[(size_t)s] && h.scope->count(b->owner[(size_t)s])) b->deliver_in_flight(s);
	}
}
