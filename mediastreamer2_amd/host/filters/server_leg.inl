// filters/server_leg.inl -- a conference SERVER's member as one device-resident batch.
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after leg_chain.inl and codec.inl);
// not compiled on its own.
//
// MSAudioConference plumbs a remote endpoint -- an AudioStream whose graph it has cut, src/voip/audioconference.c:121-179 -- as
//     rtprecv -> decoder -> [plc -> flowcontrol -> dtmfgen ->] volrecv -> in_resampler -> MIXER pin k
//     MIXER pin k -> out_resampler -> encoder -> rtpsend                                  (:209-257; audiostream.c:1812-1832)
// There is no echo canceller on such a leg: what the conference's members cost the mixer's ticker is MSVolume (volrecv: the
// level the active-speaker election reads, :419-464), the two resamplers -- which forward their blocks untouched when the
// endpoint already runs at the conference's rate (msresample.c:126-135) --, the mix and the encoder.  Facade by facade that
// is three banks and the audio crossing PCIe six times.  When EVERY linked input pin of a conference mixer of this plugin is
// fed by   anything -> MSVolume (ours, no AGC, no echo-limiter peer) -> [forwarding MSResample ->] pin   the conference moves
// into a ServerBank:
//
//   * MSVolume stages the blocks it is handed (10 or 20 ms of PCM, whatever the decoder's ptime makes them) in pinned rows;
//     at the end of the graph walk (the mixer is walked behind all of its members, msticker.c:261-282) the bank uploads them,
//     meters and levels every block AS A BLOCK (volume_process without AGC, msvolume.c:505-513: mi_volume_process), queues them
//     on the mixer channels' device FIFO (channel_process_in, audiomixer.c:78-90) and mixes the conferences that are due
//     (mi_mixer_process_volume_fifo_flags with an identity volume batch: pop + mix);
//   * a pin whose output runs  [forwarding MSResample ->] MSAlawEnc / MSUlawEnc (ours)  gets its mix ENCODED in the same batch
//     (mi_g711_encode on the mixes where they lie): 80 bytes of G.711 per member and tick come back instead of 160 bytes of PCM,
//     and the encoder facade only packs them to its ptime (alaw.c:56-90: the codes of two ticks make a 20 ms packet; encoding is
//     sample by sample, so the packet is the one the facade would have made from the PCM);  any other pin gets its PCM block
//     from a pinned slab, as LegBank's do;
//   * the census (mixer_check_bypass, audiomixer.c:244-286), the channels' queues and flow control (:92-111) run on COUNTS on the
//     host, exactly as in LegBank::conf_tick (MSMI355X_CHECK_LEVELS compares the device queue with them every flush).
//
// Equal to the facades one by one bit for bit (tests/test_plugin_server_cpu.py, tests/test_gpu_plugin_server.py), with LegBank's
// stated exception (a lone contributor is mixed, not forwarded).  Endpoints at ANOTHER rate than the conference (G.711 endpoints
// in a 16 or 48 kHz conference: both resamplers of every member work) are served too: levelled at their own rate, up-sampled in
// the batch, their pins' mixes down-sampled and encoded (ServerBank::re / q).  With AGC on its MSVolume, a peer, a facade of this
// plugin feeding MSVolume (its blocks arrive with the flush, not in the walk) or MSMI355X_NO_FUSE=1 the conference keeps its facades.

struct ServerBank;
struct ServerLeg {
	ServerBank *bank;
	int slot, pin;
	uint32_t lv_from = 0; // MSMI355X_CHECK_LEVELS: the first read-back of the levels (ServerBank::lv_seq) that is this leg's, as FusedLeg::lv_from
	MSFilter *vol, *mixer;
	MSFilter *enc = nullptr;   // the pin's output is encoded in the batch (MSAlawEnc / MSUlawEnc of this plugin), else PCM
	MSFilter *irs = nullptr;   // the endpoint's in_resampler when it really resamples (the endpoint runs at another rate than the conference): its state lives in the bank
	MSFilter *dec = nullptr;   // the leg's HEAD is MSAlawDec / MSUlawDec of this plugin right in front of MSVolume: its packets are staged as they are and decoded in the batch
	int staged = 0;            // blocks staged since the last enqueue (launch rounds)
	int new_samples = 0;       // samples MSVolume put on the mixer's queue since the mixer last looked
	int chan_samples = 0;      // the mixer channel's bufferizer, samples (what f_chan holds)
	bool metered = false;
	bool fuse_checked = false;
};

struct ServerBank : Pool {
	int rate, ns, mm, nlegs, cap; // cap: samples a staged block may hold (a row of the staging arrays)
	// Endpoints at ANOTHER rate than the conference (G.711 endpoints in a 16 kHz conference: audioconference.c:209-257 puts a working
	// in_resampler in front of every pin and a working out_resampler behind it): MSVolume meters and levels at the endpoint's rate `re`,
	// the levelled blocks are up-sampled in the batch (MSResample's own kernel on the member's state, 10 ms at a time as the facade
	// frames them) before they are queued on the channel, and an encoded pin's mix is down-sampled before it is encoded.  q = rate / re.
	int re, q, nse;                  // the endpoints' rate, the ratio, samples of 10 ms at `re`
	mi_resampler *rs_in = nullptr, *rs_out = nullptr; // [nlegs] each
	int16_t *d_up = nullptr, *d_down = nullptr;       // [nlegs][cap * q] a round's up-sampled blocks; [nlegs][nse8] the pins' down-sampled mixes
	uint8_t *h_umask = nullptr, *d_umask = nullptr;   // [kMaxRounds][pieces][nlegs]: the member has a p-th 10 ms piece in its block of that round
	uint8_t *h_omask = nullptr, *d_omask = nullptr;   // [nlegs]: the pin's mix is down-sampled (and encoded) in this launch
	int16_t *h_down = nullptr;                        // [nlegs][nse8] pinned: the down-sampled mixes of pins whose out_resampler is followed by somebody else's filter (PCM at `re`)
	int32_t *h_un = nullptr, *d_un = nullptr;         // [kMaxRounds][nlegs]: the up-sampled counts
	int pieces = 1, nse8 = 0;
	mi_volume *vol = nullptr, *vol_id = nullptr;
	mi_fifo *f_chan = nullptr;
	mi_mixer *mix = nullptr;
	int16_t *h_in, *d_in;        // [kMaxRounds][nlegs][cap] / [nlegs][cap]
	int32_t *h_n, *d_n;          // [kMaxRounds][nlegs]
	// legs headed by a G.711 decoder: the packets' bytes [kMaxRounds][nlegs][cap], and the rounds' counts by kind -- [0] MSVolume-headed,
	// [1] A-law, [2] mu-law, [3] either law -- built at enqueue from h_n (only while the bank holds such legs: ndec)
	uint8_t *h_cin = nullptr;
	int32_t *h_nk[4] = {nullptr, nullptr, nullptr, nullptr};
	int ndec = 0;
	int16_t *d_mix, *d_scratch;  // [capacity][mm][ns]; [nlegs][ns]
	uint8_t *h_codes, *d_codes;  // [nlegs][ns]: the encoded pins' G.711 bytes of this flush
	int32_t *h_len[2], *d_len[2]; // [nlegs] per law: ns where the pin's mix is encoded with that law in this launch, else 0
	uint8_t *h_run, *d_run;      // [capacity]
	uint8_t *h_dgate, *d_dgate;  // [nlegs]
	int32_t *h_lv, *d_lv;        // MSMI355X_CHECK_LEVELS
	mi_volume_state *h_vstate, *h_vround;
	std::vector<uint8_t> vhas;
	int vrounds = 0;
	int16_t *h_copy;
	std::vector<MixSlab *> slabs;
	MixSlab *cur = nullptr;
	mblk_t *root = nullptr;
	std::vector<ServerLeg *> legs;
	std::vector<MSFilter *> orss; // [nlegs]: the working out_resampler in front of that encoder (endpoints at another rate), or NULL
	std::vector<MSFilter *> encs; // [nlegs]: the encoder behind output pin (c, pin), also where no member feeds that pin (a listener)
	std::vector<uint8_t> conf_ready, flags;
	std::vector<int> lone;
	std::vector<float> gains;
	bool ctl_dirty = true;
	std::vector<uint8_t> next_flags, next_conf; // (what a method set while blocks were waiting for the coming flush: LegBank's comment)
	std::vector<float> next_gains;
	bool next_any = false;
	std::vector<mi_volume_params> vparams;
	std::vector<mi_volume_state> vstate;
	std::vector<uint8_t> vp_dirty, vs_dirty;
	bool v_dirty = false;
	struct GainPatch {
		float gain, target;
		bool also_target;
	};
	std::vector<GainPatch> vpatch;
	std::vector<std::pair<int, int>> sdrops;
	std::vector<uint64_t> conf_time;
	std::vector<uint32_t> walk_tick;
	uint32_t walk_epoch = 0;
	int walked = 0;
	bool staged_since = false, outstanding = false, early = false, early_any = false, no_early = false;
	bool mixed = false, pcm_out = false, check_levels = false, lv_fresh = false, zero_copy = true;
	uint32_t lv_seq = 0;
	uint64_t launches = 0;

	ServerBank(int cap_conf, int r, int members, int endpoint_rate) : rate(r), mm(members), re(endpoint_rate) {
		Building b(this, cap_conf);
		ns = rate / 100;
		q = rate / re;
		nse = re / 100;
		nse8 = (nse + 8 + 7) & ~7; // (a row for 10 ms at `re` and the resampler's spare sample)
		nlegs = capacity * mm;
		cap = std::min((6 * nse + 7) & ~7, 2400 / q); // 60 ms (three 20 ms packets a jitter buffer lets go at once), and what the volume kernel's LDS staging takes (50 ms at 48 kHz); a longer block is cut (server_stage)
		pieces = std::max(1, cap / nse);
		const size_t L = (size_t)nlegs;
		if (!failed) MI_MUST(mi_volume_create(hub->ctx, nlegs, re, &vol));
		if (!failed) MI_MUST(mi_volume_create(hub->ctx, nlegs, rate, &vol_id));
		if (!failed) MI_MUST(mi_fifo_create(hub->ctx, nlegs, ((4 * ns + kMaxRounds * cap * q) + 7) & ~7, &f_chan));
		if (q > 1) {
			if (!failed) MI_MUST(mi_resampler_create(hub->ctx, nlegs, (uint32_t)re, (uint32_t)rate, 3, &rs_in));
			if (!failed) MI_MUST(mi_resampler_create(hub->ctx, nlegs, (uint32_t)rate, (uint32_t)re, 3, &rs_out));
			d_up = devmem<int16_t>(L * (size_t)cap * q);
			d_down = devmem<int16_t>(L * (size_t)nse8);
			h_umask = pinned<uint8_t>((size_t)kMaxRounds * 2 * pieces * L); // (per round two kinds of rows: MSVolume-headed from pinned memory, decoder-headed from the device)
			d_umask = devmem<uint8_t>((size_t)kMaxRounds * 2 * pieces * L);
			h_omask = pinned<uint8_t>(L);
			d_omask = devmem<uint8_t>(L);
			h_down = pinned<int16_t>(L * (size_t)nse8);
			h_un = pinned<int32_t>(kMaxRounds * 2 * L);
			d_un = devmem<int32_t>(kMaxRounds * 2 * L);
		}
		if (!failed) MI_MUST(mi_mixer_create(hub->ctx, capacity, mm, ns, &mix));
		h_in = pinned<int16_t>(kMaxRounds * L * cap);
		d_in = devmem<int16_t>(L * cap);
		h_n = pinned<int32_t>(kMaxRounds * L);
		d_n = devmem<int32_t>(kMaxRounds * L);
		d_mix = devmem<int16_t>(L * ns);
		d_scratch = devmem<int16_t>(L * ns);
		h_cin = pinned<uint8_t>(kMaxRounds * L * cap);
		for (int k = 0; k < 4; ++k) h_nk[k] = pinned<int32_t>(kMaxRounds * L);
		h_codes = pinned<uint8_t>(L * ns);
		d_codes = devmem<uint8_t>(L * ns);
		for (int law = 0; law < 2; ++law) {
			h_len[law] = pinned<int32_t>(L);
			d_len[law] = devmem<int32_t>(L);
		}
		h_run = pinned<uint8_t>((size_t)capacity);
		d_run = devmem<uint8_t>((size_t)capacity);
		h_dgate = pinned<uint8_t>(L);
		d_dgate = devmem<uint8_t>(L);
		h_lv = pinned<int32_t>(L);
		d_lv = devmem<int32_t>(L);
		h_vstate = pinned<mi_volume_state>(L);
		h_vround = pinned<mi_volume_state>((size_t)kLegMeterRounds * L);
		vhas.assign((size_t)kLegMeterRounds * L, 0);
		h_copy = pinned<int16_t>(L * ns);
		legs.assign(L, nullptr);
		encs.assign(L, nullptr);
		orss.assign(L, nullptr);
		conf_ready.assign((size_t)capacity, 0);
		lone.assign((size_t)capacity, -1);
		flags.assign(L, 0);
		gains.assign(L, 1.0f);
		next_flags.assign(L, 0);
		next_gains.assign(L, 1.0f);
		next_conf.assign((size_t)capacity, 0);
		mi_volume_params p;
		mi_volume_default_params(&p);
		vparams.assign(L, p);
		vstate.resize(L);
		vp_dirty.assign(L, 0);
		vs_dirty.assign(L, 0);
		vpatch.assign(L, GainPatch{1.f, 1.f, false});
		conf_time.assign((size_t)capacity, (uint64_t)-1);
		walk_tick.assign((size_t)capacity, 0);
		check_levels = getenv("MSMI355X_CHECK_LEVELS") != nullptr;
		zero_copy = zero_copy_rows();
		no_early = getenv("MSMI355X_NO_EARLY_LAUNCH") != nullptr;
	}
	~ServerBank() override {
		if (root) freeb(root);
		for (ServerLeg *l : legs) delete l;
		if (hub->ctx) mi_ctx_sync(hub->ctx);
		if (mix) mi_mixer_destroy(mix);
		if (vol) mi_volume_destroy(vol);
		if (vol_id) mi_volume_destroy(vol_id);
		if (rs_in) mi_resampler_destroy(rs_in);
		if (rs_out) mi_resampler_destroy(rs_out);
		if (f_chan) mi_fifo_destroy(f_chan);
		for (MixSlab *s : slabs)
			if (s->state.exchange(2, std::memory_order_acq_rel) == 0) mi_host_free(hub->ctx, s);
	}
	MixSlab *free_slab() {
		for (MixSlab *s : slabs)
			if (s->state.load(std::memory_order_acquire) == 0) return s;
		if (slabs.size() >= 4 || failed) return nullptr;
		const size_t bytes = (size_t)nlegs * ns * 2;
		void *p = mi_host_alloc(hub->ctx, 64 + bytes);
		if (!p) return nullptr;
		MixSlab *s = new (p) MixSlab();
		s->bytes = bytes;
		slabs.push_back(s);
		return s;
	}

	// one tick of a conference on counts: LegBank::conf_tick's `light` case (the channel's bufferizer holds the levelled blocks,
	// the tick reads 10 ms of them or nothing, audiomixer.c:78-90) with the census of mixer_check_bypass (:244-286)
	void conf_tick(int c, uint64_t now) {
		MSFilter *mx = owner[(size_t)c];
		MixerState *s = (MixerState *)mx->data;
		conf_ready[(size_t)c] = 0;
		lone[(size_t)c] = -1;
		int count = 0, who = -1;
		for (int pin = 0; pin < mm; ++pin) {
			ServerLeg *leg = legs[(size_t)(c * mm + pin)];
			if (!leg) continue;
			uint64_t &seen = s->channels[pin].last_activity;
			bool contributes;
			if (leg->new_samples > 0) {
				seen = now;
				contributes = true;
			} else if (seen == (uint64_t)-1) {
				seen = now; // first look at a silent pin only starts its clock
				contributes = false;
			} else {
				contributes = now - seen < BYPASS_MODE_TIMEOUT;
			}
			if (contributes) ++count, who = pin;
		}
		if (count == 0) return;
		if ((count == 1) != (s->bypass_mode != FALSE))
			ms_message("mi355x mixer %p: %s", (void *)mx, count == 1 ? "a single contributor (mixed on the device all the same)" : "two or more contributors");
		s->bypass_mode = count == 1;
		for (int pin = 0; pin < mm; ++pin) {
			ServerLeg *leg = legs[(size_t)(c * mm + pin)];
			if (!leg) continue;
			Channel *chan = &s->channels[pin];
			leg->chan_samples += leg->new_samples;
			leg->new_samples = 0;
			if (leg->chan_samples >= ns) leg->chan_samples -= ns;
			const int skip = channel_flow_control_level(chan, leg->chan_samples * 2, s->skip_threshold, now);
			if (skip > 0) {
				const int k = std::min(leg->chan_samples, skip / 2);
				ms_warning("mi355x mixer: pin %i kept more than two ticks queued for 5 s; %i samples discarded", pin, k);
				leg->chan_samples -= k;
				if (k > 0) sdrops.push_back({leg->slot, k});
			}
		}
		conf_ready[(size_t)c] = 1;
		lone[(size_t)c] = count == 1 ? who : -1;
	}

	void meter_round(size_t UL) {
		if (vrounds >= kLegMeterRounds || failed) return;
		MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vround + (size_t)vrounds * nlegs));
		++vrounds;
	}

	bool enqueue() override {
		bool any = false;
		const bool was_early = early;
		if (early) {
			early = false;
			any = early_any;
		}
		if (!was_early || staged_since) any |= enqueue_at(hub_time(hub));
		outstanding = false;
		return any;
	}
	// a round's levelled blocks go onto the channels' queue: as they are, or (endpoints at another rate) up-sampled first, 10 ms at a
	// time as MSResample's facade frames them -- the p-th launch serves every member whose block has a p-th piece
	void queue_blocks(int r, int kind, int16_t *rows, const int32_t *cnt_host, const int32_t *cnt_launch) {
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm;
		if (q == 1) {
			MI_MUST(mi_fifo_push(f_chan, rows, cap, cap, cnt_launch));
			++launches;
			return;
		}
		const size_t base = (size_t)(r * 2 + kind);
		uint8_t *hm = h_umask + base * pieces * L, *dm = d_umask + base * pieces * L;
		int32_t *hu = h_un + base * L, *du = d_un + base * L;
		int maxp = 0;
		for (size_t s = 0; s < L; ++s) {
			const int n = s < UL ? cnt_host[s] : 0, whole = n / nse;
			if (n % nse && legs[s]) { // MSResample's facade would carry the odd samples over to the next block: this conference goes back to its facades
				((MixerState *)legs[s]->mixer->data)->unfuse_wanted = true;
				g_late_events.fetch_add(1, std::memory_order_relaxed);
			}
			hu[s] = whole * ns;
			maxp = std::max(maxp, whole);
			for (int p = 0; p < pieces; ++p) hm[(size_t)p * L + s] = whole > p;
		}
		if (!zero_copy) {
			MI_MUST(mi_copy_h2d_pinned(ctx, dm, hm, (size_t)pieces * L));
			MI_MUST(mi_copy_h2d_pinned(ctx, du, hu, L * 4));
		}
		for (int p = 0; p < maxp; ++p) {
			MI_MUST(mi_resampler_process_masked(rs_in, rows + (size_t)p * nse, nse, cap, d_up + (size_t)p * ns, cap * q, nullptr, (zero_copy ? hm : dm) + (size_t)p * L));
			++launches;
		}
		MI_MUST(mi_fifo_push(f_chan, d_up, cap * q, cap * q, zero_copy ? hu : du));
		++launches;
	}
	bool enqueue_at(uint64_t now) {
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm;
		if (outstanding) sync_stream();
		staged_since = false;
		if (root) emitted();
		// (the mixer's controls go up behind the conferences' ticks below: a lone contributor's are overridden, push_controls)
		if (v_dirty) {
			for (size_t s = 0; s < UL; ++s) {
				if (vp_dirty[s] == 1) {
					MI_MUST(mi_volume_set_params(vol, (int)s, 1, &vparams[s]));
					vp_dirty[s] = 0;
				}
				if (vs_dirty[s] == 1) {
					vstate[s].gain = vpatch[s].gain;
					if (vpatch[s].also_target) vstate[s].target_gain = vpatch[s].target;
					MI_MUST(mi_volume_set_state(vol, (int)s, 1, &vstate[s]));
					vs_dirty[s] = 0;
				}
			}
			v_dirty = false;
		}
		// ---- the host's half: what every leg staged
		int rounds = 0;
		for (size_t s = 0; s < UL; ++s) {
			ServerLeg *leg = legs[s];
			const int st = leg ? leg->staged : 0;
			for (int r = st; r < kMaxRounds; ++r) h_n[(size_t)r * L + s] = 0;
			if (!leg) continue;
			rounds = std::max(rounds, st);
			// (at another rate than the conference's, whole 10 ms periods of a block are up-sampled and queued -- queue_blocks; a block that is not
			// a multiple of one sends the conference back to its facades and is counted there: the host's count follows what the device queues)
			for (int r = 0; r < st; ++r) leg->new_samples += q == 1 ? h_n[(size_t)r * L + s] : (h_n[(size_t)r * L + s] / nse) * ns;
			leg->metered |= st > 0;
			for (int r = 0; r + 1 < st && vrounds + r < kLegMeterRounds; ++r) vhas[(size_t)(vrounds + r) * L + s] = 1;
			leg->staged = 0;
		}
		sdrops.clear();
		bool ticked = false;
		for (int c = 0; c < capacity; ++c) { // a mixer ticks once per ticker time, whoever enqueues
			h_run[c] = 0;
			if (failed || c >= hi || !owner[(size_t)c] || conf_time[(size_t)c] == now) continue;
			conf_time[(size_t)c] = now;
			conf_tick(c, now);
			h_run[c] = conf_ready[(size_t)c];
			ticked |= conf_ready[(size_t)c] != 0;
		}
		if (failed) return false;
		push_controls();
		bool any = false, any_dev = false;
		// ---- the device's half: every block metered and levelled as a block, then on to the channel's queue
		if (ndec > 0 && rounds) { // the rounds' counts by kind of head
			for (int r = 0; r < rounds; ++r)
				for (size_t s = 0; s < UL; ++s) {
					const ServerLeg *leg = legs[s];
					const int n = h_n[(size_t)r * L + s];
					const int kind = (leg && leg->dec) ? 1 + ((MapFilter *)leg->dec->data)->law : 0;
					for (int k = 0; k < 3; ++k) h_nk[k][(size_t)r * L + s] = k == kind ? n : 0;
					h_nk[3][(size_t)r * L + s] = kind ? n : 0;
				}
		}
		if (rounds && !zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_n, h_n, (size_t)rounds * L * 4));
		for (int r = 0; ndec > 0 && r < rounds; ++r) { // banks with decoder-headed legs: decode, then level and queue the two kinds of rows apart
			const size_t ro = (size_t)r * L;
			bool any[4] = {false, false, false, false};
			for (size_t s = 0; s < UL; ++s)
				for (int k = 0; k < 4; ++k) any[k] |= h_nk[k][ro + s] > 0;
			for (int law = 0; law < 2; ++law)
				if (any[1 + law]) {
					MI_MUST(mi_g711_decode(ctx, law ? MI_LAW_PCMU : MI_LAW_PCMA, h_cin + ro * cap, (size_t)cap, d_in, (size_t)cap, h_nk[1 + law] + ro, cap, UL));
					++launches;
				}
			if (any[3]) {
				MI_MUST(mi_volume_process(vol, d_in, cap, cap, h_nk[3] + ro));
				++launches;
				queue_blocks(r, 1, d_in, h_nk[3] + ro, h_nk[3] + ro);
			}
			if (any[0]) {
				MI_MUST(mi_volume_process(vol, h_in + ro * cap, cap, cap, h_nk[0] + ro));
				++launches;
				queue_blocks(r, 0, h_in + ro * cap, h_nk[0] + ro, h_nk[0] + ro);
			}
			if (r + 1 < rounds) meter_round(UL);
			any_dev = mixed = true;
		}
		for (int r = 0; ndec == 0 && r < rounds; ++r) {
			// (zero copy: the launches read the block's n samples where they lie in pinned memory and level them in place -- what
			// crosses PCIe is the audio, not the rows' capacity)
			const int32_t *cnt = (zero_copy ? h_n : d_n) + (size_t)r * L;
			int16_t *rows = zero_copy ? h_in + (size_t)r * L * cap : d_in;
			if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_in, h_in + (size_t)r * L * cap, UL * cap * 2));
			MI_MUST(mi_volume_process(vol, rows, cap, cap, cnt));
			++launches;
			queue_blocks(r, 0, rows, h_n + (size_t)r * L, cnt);
			if (r + 1 < rounds) meter_round(UL);
			any = mixed = true;
		}
		any |= any_dev;
		if (ticked) {
			if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_run, h_run, (size_t)capacity));
			MI_MUST(mi_mixer_process_volume_fifo_flags(mix, vol_id, 0, f_chan, d_mix, MI_VOLMIX_DRY_SKIPS, zero_copy ? h_run : d_run));
			++launches;
			for (const auto &dk : sdrops) {
				memset(h_dgate, 0, L);
				h_dgate[(size_t)dk.first] = 1;
				if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_dgate, h_dgate, L));
				for (int left = dk.second; left > 0; left -= std::min(left, ns))
					MI_MUST(mi_fifo_pop(f_chan, std::min(left, ns), d_scratch, ns, nullptr, zero_copy ? h_dgate : d_dgate, 0));
				sync_stream();
			}
			// the mixes of this launch: encoded where the pin's output is an encoder of ours, PCM elsewhere
			bool any_law[2] = {false, false};
			bool down_now = false;
			pcm_out = false;
			if (q > 1) memset(h_omask, 0, L);
			for (size_t s = 0; s < UL; ++s) {
				h_len[0][s] = h_len[1][s] = 0;
				const int c = (int)s / mm, pin = (int)s % mm;
				// (only the conferences that ticked in THIS launch: one that ticked in the walk's early launch has its codes already -- and
				// its out_resamplers must not see the same mix twice)
				if (!h_run[c] || !conf_ready[(size_t)c] || !(flags[s] & MI_MIX_OUTPUT) || pin == lone[(size_t)c]) continue;
				if (MSFilter *e = encs[s]) {
					const int law = ((MapFilter *)e->data)->law;
					h_len[law][s] = nse; // (10 ms at the endpoint's rate: ns where the endpoints run at the conference's)
					any_law[law] = true;
				} else if (orss[s]) { // a working out_resampler followed by somebody else's filter (a CPU encoder): its PCM at the endpoints' rate
					if (q > 1) h_omask[s] = 2, down_now = true;
				} else pcm_out = true;
			}
			const int16_t *enc_src = d_mix;
			size_t enc_stride = (size_t)ns;
			if (q > 1 && (any_law[0] || any_law[1] || down_now)) { // these pins' mixes down to the endpoints' rate (their out_resamplers' states)
				for (size_t s = 0; s < L; ++s) h_omask[s] = s < UL && (h_len[0][s] > 0 || h_len[1][s] > 0 || h_omask[s] == 2);
				if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_omask, h_omask, L));
				MI_MUST(mi_resampler_process_masked(rs_out, d_mix, ns, ns, d_down, nse8, nullptr, zero_copy ? h_omask : d_omask));
				++launches;
				enc_src = d_down, enc_stride = (size_t)nse8;
			}
			for (int law = 0; law < 2; ++law) {
				if (!any_law[law]) continue;
				if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_len[law], h_len[law], L * 4));
				MI_MUST(mi_g711_encode(ctx, law ? MI_LAW_PCMU : MI_LAW_PCMA, enc_src, enc_stride, zero_copy ? h_codes : d_codes, (size_t)ns,
				                       zero_copy ? h_len[law] : d_len[law], nse, UL));
				++launches;
			}
			if ((any_law[0] || any_law[1]) && !zero_copy) MI_MUST(mi_copy_d2h_pinned(ctx, h_codes, d_codes, UL * ns));
			if (down_now) MI_MUST(mi_copy_d2h_pinned(ctx, h_down, d_down, UL * (size_t)nse8 * 2));
			if (pcm_out) {
				if (!cur) cur = free_slab();
				MI_MUST(mi_copy_d2h_pinned(ctx, cur ? (void *)cur->payload() : (void *)h_copy, d_mix, UL * ns * 2));
				++launches;
			}
			any = true;
		}
		if (any) MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vstate));
		if (check_levels && any) {
			lv_fresh = true, ++lv_seq;
			MI_MUST(mi_fifo_levels(f_chan, d_lv));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_lv, d_lv, L * 4));
		}
		outstanding |= any;
		return any;
	}

	void finish() override {
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm;
		if (failed) {
			std::fill(conf_ready.begin(), conf_ready.end(), 0);
			g_late_events.fetch_add(1, std::memory_order_relaxed);
			return;
		}
		if (mixed) {
			for (size_t s = 0; s < UL; ++s) {
				ServerLeg *leg = legs[s];
				if (!leg) continue;
				vstate[s] = h_vstate[s];
				if (leg->metered && hub->ticker) { // update_energy's extremum records, msvolume.c:405-406: one per block, in order
					VolumeData *vd = (VolumeData *)leg->vol->data;
					for (int r = 0; r < vrounds; ++r)
						if (vhas[(size_t)r * L + s]) {
							vd->max.record_max(hub_time(hub), h_vround[(size_t)r * L + s].energy);
							vd->min.record_min(hub_time(hub), h_vround[(size_t)r * L + s].energy);
						}
					vd->max.record_max(hub_time(hub), vstate[s].energy);
					vd->min.record_min(hub_time(hub), vstate[s].energy);
				}
				leg->metered = false;
			}
			std::fill(vhas.begin(), vhas.end(), 0);
			vrounds = 0;
			mixed = false;
		}
		if (pcm_out && cur && !root) {
			cur->state.store(1, std::memory_order_release);
			root = esballoc(cur->payload(), cur->bytes, 0, mix_slab_release);
		}
		const bool lv_now = lv_fresh; // (a flush that launched nothing read no levels)
		lv_fresh = false;
		if (check_levels && lv_now)
			for (size_t s = 0; s < UL; ++s)
				if (legs[s] && legs[s]->lv_from <= lv_seq && h_lv[s] != legs[s]->chan_samples + legs[s]->new_samples) {
					ms_error("mi355x server leg %d: the mixer channel's queue holds %d samples, the host's framing says %d", (int)s, h_lv[s],
					         legs[s]->chan_samples + legs[s]->new_samples);
					g_late_events.fetch_add(1, std::memory_order_relaxed);
				}
	}

	void emit(MSFilter *f, int c) override; // (needs the encoder facade: below)
	void emitted() override {
		if (root) freeb(root);
		root = nullptr;
		cur = nullptr;
		pcm_out = false;
	}
	void flushed() override {
		const size_t UL = (size_t)hi * mm;
		for (size_t s = 0; s < UL; ++s) {
			if (vp_dirty[s] == 2) vp_dirty[s] = 1, v_dirty = true;
			if (vs_dirty[s] == 2) vs_dirty[s] = 1, v_dirty = true;
		}
		if (!next_any) return;
		for (int c = 0; c < hi; ++c) {
			if (!next_conf[(size_t)c]) continue;
			const size_t at = (size_t)c * mm;
			std::copy(next_flags.begin() + at, next_flags.begin() + at + mm, flags.begin() + at);
			std::copy(next_gains.begin() + at, next_gains.begin() + at + mm, gains.begin() + at);
			next_conf[(size_t)c] = 0;
			ctl_dirty = true;
		}
		next_any = false;
	}
	// a slot's owner leaves while the bank's work for the coming tick is already out: LegBank::deliver_in_flight
	// A conference with a SINGLE contributor is in the reference's bypass mode (audiomixer.c:219-286): that pin's blocks go to the other
	// outputs AS THEY ARE -- no input gain, no regard for MS_AUDIO_MIXER_SET_ACTIVE (mixer_dispatch_output never looks at the channel).
	// The batch mixes such a conference all the same, with that pin's controls set to "active, gain 1" for as long as it is alone:
	// the sum of one is the block itself (but for a sample of -32768, which the sum saturates to -32767: the stated exception).
	std::vector<int> lone_ctl;           // per conference: the pin whose controls are overridden right now, -1 = none
	std::vector<uint8_t> eff_flags;
	std::vector<float> eff_gains;
	void push_controls() {
		if (!mix) return;
		bool moved = false;
		if (lone_ctl.size() != lone.size()) lone_ctl.assign(lone.size(), -1), moved = true;
		for (size_t c = 0; c < lone.size(); ++c) {
			if (!owner[c] && lone_ctl[c] >= 0) lone_ctl[c] = -1, moved = true; // (the slot was given up)
			if (owner[c] && conf_ready[c] && lone_ctl[c] != lone[c]) lone_ctl[c] = lone[c], moved = true; // (a conference that does not tick keeps what it had)
		}
		if (!ctl_dirty && !moved) return;
		eff_flags = flags, eff_gains = gains;
		for (size_t c = 0; c < lone_ctl.size(); ++c)
			if (lone_ctl[c] >= 0) {
				const size_t at = c * (size_t)mm + (size_t)lone_ctl[c];
				eff_flags[at] |= MI_MIX_ACTIVE;
				eff_gains[at] = 1.0f;
			}
		MI_MUST(mi_mixer_set_controls(mix, eff_flags.data(), eff_gains.data()));
		ctl_dirty = false;
	}
	// a graph is being detached between two ticks (deliver_server_in_scope): rows staged in the last walk whose launches have not left --
	// a bank without early launch, a conference that joined the bank mid-walk -- leave now, as the coming flush would send them
	// (the walks are over and the ticker's clock reads what that flush would read): the tick in flight includes them
	void launch_staged() {
		if (failed || !staged_since || !hub->ticker) return;
		const bool more = enqueue_at(hub_time(hub));
		early_any = early ? (early_any || more) : more;
		early = true;
	}
	void deliver_in_flight(MSFilter *owner_filter, int slot) {
		if (failed || (!outstanding && !early)) return;
		sync_stream();
		if (failed) return;
		outstanding = false;
		finish();
		emit(owner_filter, slot);
	}
	void settle_meters() {
		if (!outstanding && !early) return;
		if (failed) return;
		sync_stream();
		if (failed) return;
		for (size_t s = 0; s < (size_t)nlegs; ++s)
			if (legs[s] && !vs_dirty[s]) vstate[s] = h_vstate[s];
	}
};

// ---- the encoder facade's half: the codes of one tick, packed to the encoder's ptime (alaw_enc_process, alaw.c:56-90)
void enc_take_codes(MSFilter *e, const uint8_t *codes, int n) {
	MapFilter *d = (MapFilter *)e->data;
	int frame_per_packet = 2;
	if (d->ptime >= 10) frame_per_packet = d->ptime / 10;
	if (frame_per_packet <= 0) frame_per_packet = 1;
	if (frame_per_packet > 14) frame_per_packet = 14;
	const size_t packet = (size_t)80 * (size_t)frame_per_packet; // 160 bytes of PCM per 10 ms at 8 kHz -> 80 codes
	while (n > 0) {
		if (!d->pending) d->pending = allocb(packet, 0);
		const size_t room = packet - (size_t)(d->pending->b_wptr - d->pending->b_rptr);
		const size_t k = std::min(room, (size_t)n);
		memcpy(d->pending->b_wptr, codes, k);
		d->pending->b_wptr += k;
		codes += k, n -= (int)k;
		if ((size_t)(d->pending->b_wptr - d->pending->b_rptr) == packet) {
			mblk_set_timestamp_info(d->pending, d->ts);
			d->ts += (uint32_t)packet;
			if (e->outputs[0]) ms_queue_put(e->outputs[0], d->pending);
			else freemsg(d->pending);
			d->pending = nullptr;
		}
	}
}

void ServerBank::emit(MSFilter *f, int c) { // mixer_process :336-343 (conference mode): one block per enabled output
	if (!conf_ready[(size_t)c]) return;
	conf_ready[(size_t)c] = 0;
	MixerState *s = (MixerState *)f->data;
	const uint8_t *base = root ? cur->payload() : reinterpret_cast<const uint8_t *>(h_copy);
	for (int pin = 0; pin < mm && pin < MIXER_MAX_CHANNELS; ++pin) {
		MSQueue *q = f->outputs[pin];
		if (!q || !s->channels[pin].output_enabled || pin == lone[(size_t)c]) continue;
		const size_t at = (size_t)(c * mm + pin);
		if (MSFilter *e = encs[at]) {
			enc_take_codes(e, h_codes + at * ns, nse);
			continue;
		}
		if (MSFilter *ors = orss[at]) { // the block its out_resampler would have made of this tick's mix, on ITS output queue (ResamplePool::emit)
			if (this->q > 1 && ors->outputs[0]) { // (`q` is the pin's queue here)
				ResampleData *rd = (ResampleData *)ors->data;
				mblk_t *om = allocb((size_t)nse * 2, 0);
				memcpy(om->b_wptr, h_down + at * (size_t)nse8, (size_t)nse * 2);
				om->b_wptr += (size_t)nse * 2;
				mblk_set_timestamp_info(om, rd->ts); // msresample.c:168-169
				rd->ts += (uint32_t)nse;
				ms_queue_put(ors->outputs[0], om);
			}
			continue;
		}
		uint8_t *row = const_cast<uint8_t *>(base) + (at * ns) * 2;
		mblk_t *om;
		if (root) {
			om = dupb(root);
			om->b_rptr = row;
			om->b_wptr = row + (size_t)ns * 2;
		} else {
			om = allocb((size_t)ns * 2, 0);
			memcpy(om->b_wptr, row, (size_t)ns * 2);
			om->b_wptr += ns * 2;
		}
		ms_queue_put(q, om);
	}
}

// ---- the facades' fused halves -----------------------------------------------------------------------------------------
// MSVolume (no AGC): every block it is handed becomes a row of the bank's staging arrays (volume_process :505-512)
void server_stage(MSFilter *f, VolumeData *d) {
	ServerLeg *leg = d->sleg;
	ServerBank *b = leg->bank;
	const size_t L = (size_t)b->nlegs;
	if (b->failed) { // no batch to be had: the blocks are lost, counted (the queue must be emptied: the ticker calls process() while there is input)
		if (!ms_queue_empty(f->inputs[0])) g_late_events.fetch_add(1, std::memory_order_relaxed);
		ms_queue_flush(f->inputs[0]);
		return;
	}
	// whole blocks wait in `backlog` (block by block: MSVolume without AGC meters each as it came) when a tick brings more than the
	// launch rounds take; the input queue is always emptied (msticker.c:244-259 calls process() while there is input)
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) putq(&d->backlog->q, m);
	for (;;) {
		if (leg->staged >= kMaxRounds) break;
		int16_t *row = b->h_in + ((size_t)leg->staged * L + (size_t)leg->slot) * b->cap;
		int n = 0;
		const size_t spilled = ms_bufferizer_get_avail(d->spill);
		mblk_t *m;
		if (spilled) {
			n = (int)std::min(spilled / 2, (size_t)b->cap);
			ms_bufferizer_read(d->spill, (uint8_t *)row, (size_t)n * 2);
		} else if ((m = getq(&d->backlog->q)) != NULL) {
			n = (int)(msgdsize(m) / 2);
			if (n > b->cap) { // longer than a row: cut into row-sized blocks, as the facade's light path does
				ms_bufferizer_put(d->spill, m);
				continue;
			}
			copy_payload(m, (uint8_t *)row);
			freemsg(m);
		} else {
			break;
		}
		b->h_n[(size_t)leg->staged * L + (size_t)leg->slot] = n;
		leg->staged++;
	}
	if (leg->staged) {
		b->staged_since = true;
		request_flush(f);
	}
}

// MSAlawDec / MSUlawDec as the leg's head: every packet it is handed becomes a row of code bytes (alaw_dec_process, alaw.c:208-221:
// one output block per packet -- which MSVolume without AGC then meters as a block)
void server_stage_codes(MSFilter *f, MapFilter *d) {
	ServerLeg *leg = (ServerLeg *)d->sleg_leg;
	ServerBank *b = leg->bank;
	const size_t L = (size_t)b->nlegs;
	if (b->failed) {
		if (!ms_queue_empty(f->inputs[0])) g_late_events.fetch_add(1, std::memory_order_relaxed);
		ms_queue_flush(f->inputs[0]);
		return;
	}
	if (!d->bz) d->bz = ms_bufferizer_new(); // (a queue of whole packets beyond a tick's launch rounds)
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) putq(&d->bz->q, m);
	while (leg->staged < kMaxRounds) {
		mblk_t *m = getq(&d->bz->q);
		if (!m) break;
		const size_t n = msgdsize(m);
		if (n == 0 || n > (size_t)b->cap) { // an empty packet makes an empty block (nothing to meter or mix); an absurdly long one is refused as the facade refuses it
			if (n) ms_error("msmi355x plugin: %s: packet of %zu bytes refused", f->desc->name, n);
			freemsg(m);
			continue;
		}
		copy_payload(m, b->h_cin + ((size_t)leg->staged * L + (size_t)leg->slot) * b->cap);
		freemsg(m);
		b->h_n[(size_t)leg->staged * L + (size_t)leg->slot] = (int)n;
		leg->staged++;
	}
	if (leg->staged) {
		b->staged_since = true;
		request_flush(f);
	}
}

// every conference of the bank has been walked in this tick: the bank's work leaves now (leg_conf_walked)
void server_conf_walked(ServerBank *b, int c) {
	if (b->no_early || b->failed || b->early || !b->hub->ticker) return;
	const uint32_t tick = b->hub->ticker->ticks;
	if (b->walk_epoch != tick) b->walk_epoch = tick, b->walked = 0;
	if (b->walk_tick[(size_t)c] == tick) return;
	b->walk_tick[(size_t)c] = tick;
	if (++b->walked < b->in_use) return;
	b->early_any = b->enqueue_at(hub_time(b->hub) + (uint64_t)b->hub->ticker->interval);
	b->early = true;
}

// ---- fusing ------------------------------------------------------------------------------------------------------------
bool is_g711_enc(const MSFilterDesc *d) { return d == &ms_mi355x_alaw_enc_desc || d == &ms_mi355x_ulaw_enc_desc; }

// the encoder behind output pin `pin` of the mixer, through a forwarding out_resampler: ours, on the mixer's ticker, idle
bool is_working_resampler(MSFilter *f, MSTicker *ticker, uint32_t in, uint32_t out);
// re: the endpoints' rate (== rate unless their resamplers work); *ors: the working out_resampler in front of the encoder;
// *blocked: the pin's output is resampled by a facade of ours but does not end in an encoder the batch can serve -- the conference keeps its facades
MSFilter *server_find_encoder(MSFilter *mx, int pin, int rate, int re, MSFilter **ors, bool *blocked) {
	*ors = nullptr;
	MSQueue *q = mx->outputs[pin];
	MSFilter *g = q ? q->next.filter : NULL;
	if (g && is_pass_resampler(g, mx->ticker) && ms_queue_empty(q)) {
		q = g->outputs[0];
		g = q ? q->next.filter : NULL;
	} else if (g && g->desc == &ms_mi355x_resample_desc && re != rate) {
		*blocked = true; // (unless everything below holds)
		if (!is_working_resampler(g, mx->ticker, (uint32_t)rate, (uint32_t)re) || !ms_queue_empty(q)) return NULL;
		*ors = g;
		q = g->outputs[0];
		g = q ? q->next.filter : NULL;
	}
	if (*ors) *blocked = false; // (whatever follows it gets PCM at the endpoints' rate from the batch, on the out_resampler's own queue -- or, below, is an encoder the batch serves)
	if (re != 8000) return NULL; // (G.711 runs at 8 kHz)
	if (!g || !is_g711_enc(g->desc) || g->ticker != mx->ticker || !q || !ms_queue_empty(q)) return NULL;
	MapFilter *d = (MapFilter *)g->data;
	if (d->sleg || ms_bufferizer_get_avail(d->bz) || (d->pool && !d->pool->staged[(size_t)d->slot].empty())) return NULL;
	*blocked = false;
	return g;
}

bool is_g711_dec(const MSFilterDesc *d) { return d == &ms_mi355x_alaw_dec_desc || d == &ms_mi355x_ulaw_dec_desc; }

// an MSResample of ours that really resamples in -> out (mono, nothing of its own in flight): its state can move into the bank and back
bool is_working_resampler(MSFilter *f, MSTicker *ticker, uint32_t in, uint32_t out) {
	if (!f || f->desc != &ms_mi355x_resample_desc || f->ticker != ticker) return false;
	const ResampleData *rd = (const ResampleData *)f->data;
	if (rd->input_rate != in || rd->output_rate != out || in == out || rd->in_nchannels != 1 || rd->out_nchannels != 1 || rd->leg) return false;
	if (ms_bufferizer_get_avail(rd->bz)) return false;
	return !(rd->pool && (rd->pool->failed || rd->pool->staged[(size_t)rd->slot] || rd->pool->ready[(size_t)rd->slot]));
}
// (the rates the batch resamples between: whole ratios the up-sampler's kernel takes, 10 ms a whole number of samples on both sides)
bool server_rates_ok(int re, int rate) { return re > 0 && rate % re == 0 && (rate / re == 2 || rate / re == 3 || rate / re == 6) && re % 100 == 0; }

bool server_candidate(MSFilter *mx, MixerState *ms, int pin, MSFilter **vol_out, MSFilter **dec_out, MSFilter **irs_out, int *re_out) {
	MSQueue *q = mx->inputs[pin];
	MSFilter *vol = q->prev.filter;
	*irs_out = nullptr;
	*re_out = ms->rate;
	if (is_pass_resampler(vol, mx->ticker)) { // the endpoint's in_resampler, forwarding
		if (!ms_queue_empty(q)) return false;
		q = vol->inputs[0];
		vol = q ? q->prev.filter : NULL;
	} else if (vol && vol->desc == &ms_mi355x_resample_desc) { // ... or working: the endpoint runs at another rate (a G.711 endpoint in a 16 kHz conference)
		const int re = (int)((const ResampleData *)vol->data)->input_rate;
		if (!server_rates_ok(re, ms->rate) || !is_working_resampler(vol, mx->ticker, (uint32_t)re, (uint32_t)ms->rate) || !ms_queue_empty(q)) return false;
		*irs_out = vol;
		*re_out = re;
		q = vol->inputs[0];
		vol = q ? q->prev.filter : NULL;
	}
	const int re = *re_out;
	if (!vol || vol->desc != &ms_mi355x_volume_desc || vol->ticker != mx->ticker || !ms_queue_empty(q)) return false;
	// MSVolume must be handed its blocks IN the graph walk -- by a filter that is not one of this plugin's (dtmfgen stands in front
	// of volrecv in an AudioStream, audiostream.c:1826; a CPU decoder; a sound card): a facade of ours delivers with the flush, a tick
	// later, and the conference would tick before its members' blocks arrive.  Such a conference keeps its facades.
	// ... unless that facade is a G.711 decoder of ours which is itself handed its packets in the walk (rtprecv in front of it): then
	// the DECODER is the leg's head -- its packets are staged as they are and decoded in the batch (80 bytes per 10 ms up instead of 160)
	MSQueue *qin = vol->inputs[0];
	*dec_out = nullptr;
	if (!qin || !qin->prev.filter) return false;
	if (is_ours(qin->prev.filter->desc)) {
		MSFilter *dec = qin->prev.filter;
		MapFilter *dd = (MapFilter *)dec->data;
		MSQueue *qd = dec->inputs[0];
		if (!is_g711_dec(dec->desc) || dec->ticker != mx->ticker || re != 8000 || !ms_queue_empty(qin) || !qd || !qd->prev.filter || is_ours(qd->prev.filter->desc)) return false;
		if (dd->sleg || (dd->pool && (!dd->pool->staged[(size_t)dd->slot].empty() || !dd->pool->ready[(size_t)dd->slot].empty()))) return false;
		*dec_out = dec;
	}
	VolumeData *vd = (VolumeData *)vol->data;
	if (volume_is_peered(vd) || vd->sample_rate != re || vd->leg || vd->sleg || vd->p.agc_enabled) return false;
	if (ms_bufferizer_get_avail(vd->buffer) || ms_bufferizer_get_avail(vd->spill)) return false;
	const size_t held = ms_bufferizer_get_avail(&ms->channels[pin].bufferizer); // (from before this attach: it moves to the bank's channel queue)
	if (held % 16 || held > (size_t)(ms->rate / 100) * 2 * 3) return false;
	if (vd->pool && (vd->pool->staged[(size_t)vd->slot] || vd->pool->ready[(size_t)vd->slot])) return false; // (a block of its own in flight)
	*vol_out = vol;
	return true;
}

// Called (hub locked, ticker thread) by conf_try_fuse when the conference is not one of sending legs.  true = fused.
bool server_try_fuse(MSFilter *mx) {
	MixerState *ms = (MixerState *)mx->data;
	if (getenv("MSMI355X_NO_FUSE") != nullptr) return false;
	if (!ms->prepared || ms->conf_mode == 0 || ms->nchannels != 1 || !mx->ticker || mx->ticker->interval != 10 || ms->rate % 800) return false;
	std::vector<std::pair<int, MSFilter *>> cand;
	std::vector<MSFilter *> heads; // per candidate: the decoder that heads the leg, or NULL (MSVolume does)
	std::vector<MSFilter *> irss;  // per candidate: its working in_resampler, or NULL
	int maxpin = -1, re = -1;
	for (int pin = 0; pin < mx->desc->ninputs; ++pin) {
		if (!mx->inputs[pin]) continue;
		MSFilter *vol = nullptr, *dec = nullptr, *irs = nullptr;
		int re_pin = 0;
		if (!server_candidate(mx, ms, pin, &vol, &dec, &irs, &re_pin)) return false;
		if (re >= 0 && re_pin != re) return false; // (one endpoint rate per conference: a bank is one shape)
		re = re_pin;
		cand.push_back({pin, vol});
		heads.push_back(dec);
		irss.push_back(irs);
		maxpin = pin;
	}
	if (cand.empty()) return false;
	// the outputs: an encoder of ours behind the pin (through its out_resampler) is served by the batch, anything else gets PCM at the
	// conference's rate -- but an out_resampler of ours that WORKS must end in such an encoder (its PCM is not the batch's to make)
	std::vector<MSFilter *> enc_of((size_t)mx->desc->noutputs, nullptr), ors_of((size_t)mx->desc->noutputs, nullptr);
	for (int pin = 0; pin < mx->desc->noutputs; ++pin) {
		if (!mx->outputs[pin]) continue;
		bool blocked = false;
		enc_of[(size_t)pin] = server_find_encoder(mx, pin, ms->rate, re, &ors_of[(size_t)pin], &blocked);
		if (blocked) return false;
	}
	for (int pin = 0; pin < mx->desc->noutputs; ++pin)
		if (mx->outputs[pin]) maxpin = std::max(maxpin, pin);
	int mm = MIXER_MAX_CHANNELS;
	for (int m : {4, 8, 16, 32})
		if (maxpin < m) {
			mm = m;
			break;
		}
	const int rate = ms->rate;
	ServerBank *b = bank<ServerBank>("srv:" + std::to_string(rate) + ":" + std::to_string(re) + ":" + std::to_string(mm), 1,
	                                 [&](int cap) { return new ServerBank(std::max(1, cap * 4 / mm), rate, mm, re); });
	const int c = b ? b->acquire(mx) : -1;
	if (c < 0) return false;
	note_slot(mx);
	const int s0 = c * mm;
	bool ok = mi_fifo_reset_range(b->f_chan, s0, mm) == MI_OK && mi_volume_reset_max(b->vol, s0, mm) == MI_OK;
	if (b->q > 1) ok = ok && mi_resampler_reset(b->rs_in, s0, mm) == MI_OK && mi_resampler_reset(b->rs_out, s0, mm) == MI_OK;
	for (const auto &pv : cand) {
		const size_t s = (size_t)(s0 + pv.first);
		VolumeData *vd = (VolumeData *)pv.second->data;
		volume_keep_state(vd);
		b->vstate[s] = volume_start_state(vd);
		b->vparams[s] = vd->p;
		b->vparams[s].peer = -1;
	}
	ok = ok && mi_volume_set_params(b->vol, s0, mm, &b->vparams[(size_t)s0]) == MI_OK && mi_volume_set_state(b->vol, s0, mm, &b->vstate[(size_t)s0]) == MI_OK;
	if (!ok) {
		mi_failed("fusing a conference's remote members");
		b->release(c);
		return false;
	}
	for (int pin = 0; pin < mm; ++pin) {
		b->flags[(size_t)(s0 + pin)] = 0, b->gains[(size_t)(s0 + pin)] = 1.0f;
		b->encs[(size_t)(s0 + pin)] = nullptr;
		b->orss[(size_t)(s0 + pin)] = nullptr;
	}
	int nenc = 0;
	for (int pin = 0; pin < mm && pin < mx->desc->noutputs; ++pin) {
		if (MSFilter *e = enc_of[(size_t)pin]) {
			b->encs[(size_t)(s0 + pin)] = e;
			((MapFilter *)e->data)->sleg_bank = b;
			((MapFilter *)e->data)->sleg = true;
			++nenc;
		}
		if (MSFilter *ors = ors_of[(size_t)pin]) { // a working out_resampler: its state moves into the bank (msresample.c:117-120: the handle lives as long as the filter)
			ResampleData *rd = (ResampleData *)ors->data;
			if (rd->pool && rd->slots->size() == 1) resample_keep_from(rd, rd->pool->r, rd->slot, rd->input_rate, rd->output_rate);
			resample_restore_to(rd, b->rs_out, s0 + pin, (uint32_t)rate, (uint32_t)re, false);
			b->orss[(size_t)(s0 + pin)] = ors;
		}
	}
	for (size_t ci = 0; ci < cand.size(); ++ci) {
		const auto &pv = cand[ci];
		ServerLeg *leg = new ServerLeg();
		leg->lv_from = b->lv_seq + 1;
		leg->bank = b, leg->slot = s0 + pv.first, leg->pin = pv.first;
		leg->vol = pv.second, leg->mixer = mx;
		leg->enc = b->encs[(size_t)leg->slot];
		if (MSFilter *irs = irss[ci]) { // the in_resampler's state likewise
			ResampleData *rd = (ResampleData *)irs->data;
			if (rd->pool && rd->slots->size() == 1) resample_keep_from(rd, rd->pool->r, rd->slot, rd->input_rate, rd->output_rate);
			resample_restore_to(rd, b->rs_in, leg->slot, (uint32_t)re, (uint32_t)rate, false);
			leg->irs = irs;
		}
		if (MSFilter *dec = heads[ci]) {
			MapFilter *dd = (MapFilter *)dec->data;
			leg->dec = dec;
			dd->sleg = true, dd->sleg_bank = b, dd->sleg_leg = leg;
			b->ndec++;
		}
		b->legs[(size_t)leg->slot] = leg;
		VolumeData *vd = (VolumeData *)pv.second->data;
		if (const int held = (int)(ms_bufferizer_get_avail(&ms->channels[leg->pin].bufferizer) / 2)) { // the channel's queue as the last detach left it
			std::vector<int16_t> x((size_t)held);
			ms_bufferizer_read(&ms->channels[leg->pin].bufferizer, (uint8_t *)x.data(), (size_t)held * 2);
			if (fifo_give(b->hub->ctx, b->f_chan, b->nlegs, leg->slot, b->ns, b->d_scratch, b->d_n, x.data(), held)) leg->chan_samples = held;
			else mi_failed("moving a mixer channel's queue to the device");
		}
		if (vd->pool) {
			vd->pool->release(vd->slot);
			vd->pool = nullptr, vd->slot = -1;
		}
		vd->sleg = leg;
	}
	if (ms->pool) {
		ms->pool->staged[(size_t)ms->slot] = ms->pool->ready[(size_t)ms->slot] = 0;
		ms->pool->release(ms->slot);
		ms->pool = nullptr, ms->slot = -1;
	}
	ms->sbank = b, ms->sconf = c;
	b->conf_time[(size_t)c] = (uint64_t)-1;
	b->staged_since = true;
	ms->unfuse_wanted = false;
	mixer_push_controls(mx, ms);
	ms_message("mi355x: conference %p fused: %d remote members at %d Hz (the conference at %d Hz), %d of their outputs encoded in the batch (bank of %d x %d)", (void *)mx,
	           (int)cand.size(), re, rate, nenc, b->capacity, mm);
	return true;
}

void server_unfuse(MSFilter *mx, bool keep_running) {
	MixerState *ms = (MixerState *)mx->data;
	ServerBank *b = ms->sbank;
	if (!b) return;
	HubLock lk(b->hub);
	const int c = ms->sconf, mm = b->mm;
	b->deliver_in_flight(mx, c);
	b->settle_meters();
	if (!b->failed) { // what the channels' queues hold goes back into the mixer's own bufferizers, which outlive a detach (audiomixer.c:64-76,132-135,200-208)
		std::vector<int> want((size_t)mm, 0);
		for (int pin = 0; pin < mm; ++pin)
			if (ServerLeg *leg = b->legs[(size_t)(c * mm + pin)]) {
				want[(size_t)pin] = leg->chan_samples + leg->new_samples;
				leg->chan_samples = leg->new_samples = 0;
			}
		if (!fifo_take(b->hub->ctx, b->f_chan, b->nlegs, c * mm, b->ns, b->d_scratch, b->d_dgate, want,
		               [&](int s, const int16_t *x, int n) { bufferizer_put_samples(&ms->channels[s - c * mm].bufferizer, x, n); }))
			mi_failed("taking the mixer channels' queues back");
	}
	std::vector<ServerLeg *> gone;
	auto state_back = [&](MSFilter *rs, mi_resampler *from, int slot) { // a working resampler's state returns to its filter (and to the slot it kept)
		ResampleData *rd = (ResampleData *)rs->data;
		if (b->failed || !from) return;
		resample_keep_from(rd, from, slot, rd->input_rate, rd->output_rate);
		if (rd->pool && rd->slots->size() == 1 && !rd->pool->failed) resample_restore_to(rd, rd->pool->r, rd->slot, rd->input_rate, rd->output_rate, false);
	};
	for (int pin = 0; pin < mm; ++pin) {
		const size_t s = (size_t)(c * mm + pin);
		if (MSFilter *e = b->encs[s]) {
			((MapFilter *)e->data)->sleg = false;
			((MapFilter *)e->data)->sleg_bank = nullptr;
			b->encs[s] = nullptr;
		}
		if (MSFilter *ors = b->orss[s]) {
			state_back(ors, b->rs_out, (int)s);
			b->orss[s] = nullptr;
		}
		if (b->legs[s] && b->legs[s]->irs) state_back(b->legs[s]->irs, b->rs_in, (int)s);
		ServerLeg *leg = b->legs[s];
		if (!leg) continue;
		VolumeData *vd = (VolumeData *)leg->vol->data;
		if (leg->dec) { // the decoder takes back what it staged in a walk whose launches never left, and what waited behind that
			MapFilter *dd = (MapFilter *)leg->dec->data;
			for (int r = 0; r < leg->staged; ++r) {
				const int n = b->h_n[(size_t)r * b->nlegs + s];
				mblk_t *m = allocb((size_t)n, 0);
				memcpy(m->b_wptr, b->h_cin + ((size_t)r * b->nlegs + s) * b->cap, (size_t)n);
				m->b_wptr += n;
				if (leg->dec->inputs[0]) putq(&leg->dec->inputs[0]->q, m);
				else freemsg(m);
			}
			leg->staged = 0;
			if (dd->bz)
				for (mblk_t *m; (m = getq(&dd->bz->q)) != NULL;) {
					if (leg->dec->inputs[0]) putq(&leg->dec->inputs[0]->q, m);
					else freemsg(m);
				}
			dd->sleg = false, dd->sleg_bank = nullptr, dd->sleg_leg = nullptr;
			b->ndec--;
		}
		if (!b->failed) { // MSVolume's running state goes with the filter (volume.inl: VolumeData::kept)
			vd->kept = b->vstate[s];
			if (b->vs_dirty[s]) {
				vd->kept.gain = b->vpatch[s].gain;
				if (b->vpatch[s].also_target) vd->kept.target_gain = b->vpatch[s].target;
			}
			vd->has_kept = true;
		}
		for (int r = 0; r < leg->staged; ++r) { // blocks staged in a walk whose launches never left (a tick without early launch): back to the filter's queue
			const int n = b->h_n[(size_t)r * b->nlegs + s];
			mblk_t *m = allocb((size_t)n * 2, 0);
			memcpy(m->b_wptr, b->h_in + ((size_t)r * b->nlegs + s) * b->cap, (size_t)n * 2);
			m->b_wptr += n * 2;
			ms_bufferizer_put(vd->spill, m);
		}
		// (blocks still waiting in the backlog go to the facade's own path through its input queue's place: its spill)
		for (mblk_t *m; (m = getq(&vd->backlog->q)) != NULL;) ms_bufferizer_put(vd->spill, m);
		vd->sleg = nullptr;
		b->legs[s] = nullptr;
		gone.push_back(leg);
	}
	b->conf_ready[(size_t)c] = 0;
	ms->sbank = nullptr, ms->sconf = -1;
	ms->fuse_state = keep_running ? 2 : 0;
	ms->unfuse_wanted = false;
	for (int pin = 0; pin < mm; ++pin) b->flags[(size_t)(c * mm + pin)] = 0;
	b->ctl_dirty = true;
	if (keep_running) {
		mixer_prepare(mx, true);
		ms_warning("mi355x: conference %p left its fused batch (a member's configuration changed); the facades carry on one by one", (void *)mx);
	}
	b->release(c); // (may destroy the bank)
	for (ServerLeg *leg : gone) delete leg;
}

void server_release(ServerLeg *leg, bool keep_running) {
	if (leg) server_unfuse(leg->mixer, keep_running);
}
void server_disqualify(ServerLeg *leg) {
	if (leg) ((MixerState *)leg->mixer->data)->unfuse_wanted = true;
}
bool server_wants_out(ServerLeg *leg) { return leg && ((MixerState *)leg->mixer->data)->unfuse_wanted; }
Pool *server_pool(ServerLeg *leg) { return leg->bank; }
Pool *server_pool_of(ServerBank *b) { return b; }
Pool *server_pool_of_map(MapFilter *d) { return (ServerBank *)d->sleg_bank; }
mi_volume_state *server_vstate(ServerLeg *leg) { return &leg->bank->vstate[(size_t)leg->slot]; }
void server_push_volume(ServerLeg *leg, const mi_volume_params *p, const float *gain, const float *target) {
	ServerBank *b = leg->bank;
	const size_t s = (size_t)leg->slot;
	const uint8_t when = b->work_waiting() ? 2 : 1;
	b->vparams[s] = *p;
	b->vparams[s].peer = -1;
	b->vp_dirty[s] = when;
	if (gain) {
		b->vpatch[s] = {*gain, target ? *target : 0.f, target != nullptr};
		b->vs_dirty[s] = when;
	}
	b->v_dirty = true;
}
void server_push_mixer_controls(MSFilter *f, MixerState *s, bool from_method) {
	ServerBank *b = s->sbank;
	const bool later = from_method && b->work_waiting();
	std::vector<uint8_t> &fl_row = later ? b->next_flags : b->flags;
	std::vector<float> &g_row = later ? b->next_gains : b->gains;
	for (int pin = 0; pin < b->mm; ++pin) {
		const size_t at = (size_t)(s->sconf * b->mm + pin);
		uint8_t fl = 0;
		if (f->inputs[pin] && b->legs[at]) fl |= MI_MIX_LINKED;
		if (s->channels[pin].active) fl |= MI_MIX_ACTIVE;
		if (f->outputs[pin] && s->channels[pin].output_enabled) fl |= MI_MIX_OUTPUT;
		fl_row[at] = fl;
		g_row[at] = s->channels[pin].gain;
	}
	if (later) b->next_conf[(size_t)s->sconf] = 1, b->next_any = true;
	else b->next_conf[(size_t)s->sconf] = 0, b->ctl_dirty = true;
}
// the encoder of a fused pin was detached or destroyed: its conference leaves the batch
void server_encoder_gone(MSFilter *e) { // (an encoder OR a decoder of a fused member)
	MapFilter *d = (MapFilter *)e->data;
	if (!d->sleg || !d->sleg_bank) return;
	ServerBank *b = (ServerBank *)d->sleg_bank;
	MSFilter *mx = d->sleg_leg ? ((ServerLeg *)d->sleg_leg)->mixer : nullptr;
	if (!mx) {
		HubLock lk(b->hub);
		for (size_t s = 0; s < b->encs.size() && !mx; ++s)
			if (b->encs[s] == e) mx = b->owner[s / (size_t)b->mm];
	}
	if (mx) server_unfuse(mx, false);
}
void deliver_server_in_scope(TickerHub &h) {
	for (Pool *p : h.pools) {
		if (p->key.compare(0, 4, "srv:") != 0) continue;
		ServerBank *b = static_cast<ServerBank *>(p);
		bool ours = false; // (as in deliver_fused_in_scope: the whole bank is launched only when the detaching conference itself has staged blocks that have not left)
		for (int s = 0; s < b->hi && !ours; ++s) {
			if (!b->owner[(size_t)s] || !h.scope->count(b->owner[(size_t)s])) continue;
			for (int pin = 0; pin < b->mm && !ours; ++pin)
				if (const ServerLeg *leg = b->legs[(size_t)(s * b->mm + pin)]) ours = leg->staged > 0;
		}
		if (ours) b->launch_staged();
		for (int s = 0; s < b->hi; ++s)
			if (b->owner[(size_t)s] && h.scope->count(b->owner[(size_t)s])) b->deliver_in_flight(b->owner[(size_t)s], s);
	}
}
