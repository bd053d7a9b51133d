// filters/volume.inl -- MSVolume facade (src/audiofilters/msvolume.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

// ====================================================================== volume
struct Extremum { // OrtpExtremum (oRTP utils): windowed min/max, period in ms
	float current = 0, last_stable = 0;
	uint64_t t0 = (uint64_t)-1;
	int period;
	void reset() {
		current = last_stable = 0;
		t0 = (uint64_t)-1;
	}
	bool check_init(uint64_t now, float v) {
		if (t0 != (uint64_t)-1 && (int)(now - t0) > period) {
			last_stable = current;
			t0 = (uint64_t)-1;
		}
		if (t0 == (uint64_t)-1) {
			current = v;
			t0 = now;
			return true;
		}
		return false;
	}
	void record_min(uint64_t now, float v) {
		check_init(now, v);
		if (v < current) current = v;
	}
	void record_max(uint64_t now, float v) {
		check_init(now, v);
		if (v > current) current = v;
	}
};

struct VolumePool : Pool {
	int rate, cap_samples;
	mi_volume *v = nullptr;
	int16_t *h_buf, *d_buf;
	int32_t *h_n, *d_n;
	int32_t *h_nsc; // the rounds' count rows while a detaching graph's slots alone are flushed (the others' h_n rows stay as staged)
	mi_volume_state *h_state; // pinned: the meters come back with the blocks, no synchronisation of their own
	bool fetched = false;
	int rounds_fetched = 0;
	std::vector<int> staged, ready;
	// [kMaxRounds][capacity]: the row is a COPY for the meter -- the block itself went on in the walk, untouched (volume_process: an
	// MSVolume that is a meter and nothing else); emit() records its energy and makes no block of it
	std::vector<uint8_t> quiet;
	std::vector<mi_volume_params> params;
	std::vector<mi_volume_state> state;
	// params_dirty / state_dirty: 1 = to the device with the next enqueue, 2 = set by a method while the last walk's blocks were still
	// waiting for the coming flush (Pool::work_waiting): behind that flush (flushed()).  A gain method patches two fields of the running
	// state as the device holds it (gain_patch), it does not replace the state
	std::vector<uint8_t> params_dirty, state_dirty;
	struct GainPatch {
		float gain, target;
		bool also_gain, also_target, whole; // whole: `state[slot]` itself is to be written (a new slot's start state)
	};
	std::vector<GainPatch> gain_patch;
	std::vector<uint8_t> used;         // the slot has had a filter since the bank was created
	mi_volume_params params0;          // what mi_volume_create leaves in every slot: a new filter whose parameters and start state
	mi_volume_state state0;            // are these needs no device call at all (volume_attach_slot)
	VolumePool(int cap, int r) : rate(r) {
		Building b(this, cap);
		if (!failed) MI_MUST(mi_volume_create(hub->ctx, capacity, rate, &v));
		cap_samples = std::max(960, rate / 100 * 2);
		cap_samples = (cap_samples + 7) & ~7;
		const size_t c = (size_t)capacity;
		h_buf = pinned<int16_t>(kMaxRounds * c * cap_samples);
		h_n = pinned<int32_t>(kMaxRounds * c);
		h_nsc = pinned<int32_t>(kMaxRounds * c);
		d_buf = devmem<int16_t>(c * cap_samples);
		d_n = devmem<int32_t>(c);
		h_state = pinned<mi_volume_state>(kMaxRounds * c); // row r: the meters behind round r (every chunk's energy is recorded, msvolume.c:405-406)
		staged.assign(c, 0);
		ready.assign(c, 0);
		quiet.assign(kMaxRounds * c, 0);
		mi_volume_params p;
		mi_volume_default_params(&p);
		params.assign(c, p);
		params0 = p;
		state.resize(c);
		if (!failed) MI_MUST(mi_volume_get_state(v, 0, capacity, state.data()));
		state0 = state[0];
		used.assign(c, 0);
		params_dirty.assign(c, 0);
		state_dirty.assign(c, 0);
		gain_patch.assign(c, GainPatch{1.f, 1.f, false, false, false});
	}
	~VolumePool() override {
		if (v) mi_volume_destroy(v);
	}
	bool enqueue() override {
		mi_ctx *ctx = hub->ctx;
		const size_t c = (size_t)capacity, u = (size_t)hi; // rows [0, hi) are all that was ever handed out
		// (runs of neighbouring slots go up in ONE call each: every call waits for the stream, and a ticker's streams attached together are
		// thousands of neighbours -- one call per slot took sixteen tickers' first flush to a second)
		for (int s = 0; s < hi; ++s)
			if (state_dirty[(size_t)s] == 1) {
				GainPatch &g = gain_patch[(size_t)s];
				if (g.also_gain) state[(size_t)s].gain = g.gain;
				if (g.also_target) state[(size_t)s].target_gain = g.target;
				g = GainPatch{1.f, 1.f, false, false, false};
			}
		for (int s = 0; s < hi;) {
			if (params_dirty[(size_t)s] != 1) {
				++s;
				continue;
			}
			int e = s;
			while (e < hi && params_dirty[(size_t)e] == 1) params_dirty[(size_t)e++] = 0;
			MI_MUST(mi_volume_set_params(v, s, e - s, &params[(size_t)s]));
			s = e;
		}
		for (int s = 0; s < hi;) {
			if (state_dirty[(size_t)s] != 1) {
				++s;
				continue;
			}
			int e = s;
			while (e < hi && state_dirty[(size_t)e] == 1) state_dirty[(size_t)e++] = 0;
			MI_MUST(mi_volume_set_state(v, s, e - s, &state[(size_t)s]));
			s = e;
		}
		int maxr = 0;
		for (int s = 0; s < hi; ++s)
			if (!parked(s)) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			const int32_t *nrow = h_n + r * c;
			if (hub->scope) { // a detaching graph's slots alone: everybody else counts as empty in THIS launch and keeps what it staged
				for (int s = 0; s < capacity; ++s) h_nsc[r * c + s] = (s < hi && staged[(size_t)s] > r && !parked(s)) ? h_n[r * c + s] : 0;
				nrow = h_nsc + r * c;
			} else {
				for (int s = 0; s < capacity; ++s)
					if (s >= hi || staged[(size_t)s] <= r) h_n[r * c + s] = 0;
			}
			if (zero_copy_rows()) { // the launch reads and levels the blocks where they lie in pinned memory: what crosses PCIe is the audio, not the rows' capacity
				MI_MUST(mi_volume_process(v, h_buf + r * c * cap_samples, cap_samples, cap_samples, nrow));
			} else {
				MI_MUST(mi_copy_h2d_pinned(ctx, d_buf, h_buf + r * c * cap_samples, u * cap_samples * 2));
				MI_MUST(mi_copy_h2d_pinned(ctx, d_n, nrow, c * 4));
				MI_MUST(mi_volume_process(v, d_buf, cap_samples, cap_samples, d_n));
				MI_MUST(mi_copy_d2h_pinned(ctx, h_buf + r * c * cap_samples, d_buf, u * cap_samples * 2));
			}
			if (!failed) MI_MUST(mi_volume_get_state_async(v, 0, hi, h_state + r * c)); // meters for the app thread
		}
		fetched = maxr > 0 && !failed;
		rounds_fetched = fetched ? maxr : 0;
		return maxr > 0;
	}
	void finish() override {
		if (fetched && !failed) // (a new slot's start state that has not gone to the device yet is not overwritten)
			for (int s = 0; s < hi; ++s)
				if (!(state_dirty[(size_t)s] && gain_patch[(size_t)s].whole)) state[(size_t)s] = h_state[(size_t)(rounds_fetched - 1) * capacity + s];
		fetched = false;
		for (int s = 0; s < hi; ++s) {
			if (parked(s)) continue;
			ready[(size_t)s] = staged[(size_t)s]; // after a failed launch the staged blocks leave as they came (unity gain)
			staged[(size_t)s] = 0;
		}
	}
	bool scoped() const override { return true; }
	void flushed() override;
	void emit(MSFilter *f, int slot) override;
};

struct VolumeData;
void volume_framing_now(MSFilter *f);
void VolumePool::flushed() {
		for (int s = 0; s < hi; ++s) {
			if (parked(s)) continue;
			if (params_dirty[(size_t)s] == 2) {
				params_dirty[(size_t)s] = 1;
				if (owner[(size_t)s]) volume_framing_now(owner[(size_t)s]); // (the facade's framing follows its parameters: see VolumeData::chunks)
			}
			if (state_dirty[(size_t)s] == 2) state_dirty[(size_t)s] = 1;
		}
}

struct VolumeData { // struct Volume msvolume.c:48-86, host-side part
	mi_volume_params p;
	// how process() frames what it is handed -- 10 ms chunks with AGC or an echo-limiter peer (msvolume.c:480), else every block as it is
	// (:505).  Follows the parameters WHEN THEY GO LIVE: a method that arrives while the last walk's blocks are still waiting for the
	// coming flush (Pool::work_waiting) re-frames only behind that flush, as the reference's process() of that walk ran before the call
	bool chunks;
	float gain, target_gain; // pending values for a slot not yet acquired
	int sample_rate, nsamples;
	MSFilter *peer;
	MSBufferizer *buffer;
	MSBufferizer *spill; // light path: the part of an over-long block that did not fit this tick's rounds
	bool feeds_far_end;  // its blocks end up on a canceller's far-end pin (looked up at every attach: volume_passes)
	bool spill_quiet;    // ... which is a copy for the meter (the block itself went on in the walk: volume_passes)
	MSBufferizer *backlog; // a conference server's member (server_leg.inl): whole blocks beyond a tick's launch rounds, kept block by block
	Extremum min, max;
	// struct Volume lives as long as the filter: energy, the gain ramp, the noise gate's and the echo limiter's counters all survive a
	// detach / re-attach of the graph (msvolume.c:88-118 sets them once, :447-469 only resets the extrema).  Here the running state
	// lives in a bank slot: `kept` carries it from a slot that is given up (the graph re-plumbed, a conference fused or un-fused)
	// to the next one
	mi_volume_state kept;
	bool has_kept;
	VolumePool *pool;
	int slot;
	bool ng_soft_start;
	FusedLeg *leg;  // the filter is part of a fused call leg (filters/leg_chain.inl): its meter lives in that bank
	ServerLeg *sleg; // ... or a conference server's member (filters/server_leg.inl): volrecv as the head of the leg
	// ... or the echo-limiter PEER of a fused leg's MSVolume (volrecv of an AudioStream whose volsend names it, audiostream.c:2240):
	// metered in that leg's bank (LegBank::vol_peer); this facade hands its blocks on untouched, in the walk
	FusedLeg *meter_leg;
	bool fuse_checked; // looked for a conference of remote members to fuse with since the last attach
	// MSVolume filters that named this one as their echo-limiter peer (it must stay in a bank of its own kind).  Back-pointers,
	// under g_peer_mu: a peer that is destroyed FIRST -- audio_stream_free destroys volrecv before volsend, audiostream.c:357-358,
	// and volsend's peer is volrecv (:2240) -- is forgotten by those who named it; nobody ever reaches into a freed filter
	std::vector<MSFilter *> peered_by;
	std::atomic<bool> peer_gone; // the peer was destroyed: the slot's parameters follow at the next block
};
std::mutex g_peer_mu;
bool volume_is_peered(VolumeData *d) {
	std::lock_guard<std::mutex> g(g_peer_mu);
	return d->peer != NULL || !d->peered_by.empty();
}
mi_volume_state *leg_vstate(FusedLeg *leg);                      // leg_chain.inl
mi_volume_state *leg_pstate(FusedLeg *leg);                      // (the state of the leg's metered peer)
void leg_push_volume(FusedLeg *leg, const mi_volume_params *p, const float *gain, const float *target); // (gain: also the running state's)
mi_volume_state *server_vstate(ServerLeg *leg);                  // server_leg.inl
void server_push_volume(ServerLeg *leg, const mi_volume_params *p, const float *gain, const float *target);
MSFilter *leg_volume_sink(MSFilter *vol);

void volume_init(MSFilter *f) { // msvolume.c:88-118
	VolumeData *d = new VolumeData();
	mi_volume_default_params(&d->p);
	d->gain = d->target_gain = 1;
	d->chunks = false;
	d->spill_quiet = false;
	d->feeds_far_end = false;
	d->sample_rate = 8000;
	d->nsamples = 80;
	d->peer = NULL;
	d->buffer = ms_bufferizer_new();
	d->spill = ms_bufferizer_new();
	d->backlog = ms_bufferizer_new();
	d->max.period = 1000;
	d->min.period = 30000;
	d->pool = nullptr;
	d->slot = -1;
	d->leg = nullptr;
	d->sleg = nullptr;
	d->meter_leg = nullptr;
	d->fuse_checked = false;
	d->peer_gone = false;
	d->has_kept = false;
	f->data = d;
}

void volume_postprocess(MSFilter *f) { // detach: a fused conference goes back to its facades' own banks
	VolumeData *d = (VolumeData *)f->data;
	facade_detached(f);
	if (d->leg) leg_release(d->leg, false);
	if (d->meter_leg) leg_release(d->meter_leg, false);
	if (d->sleg) server_release(d->sleg, false);
	d->fuse_checked = false;
}

void volume_uninit(MSFilter *f) {
	VolumeData *d = (VolumeData *)f->data;
	if (d->leg) leg_release(d->leg, false);
	if (d->meter_leg) leg_release(d->meter_leg, false);
	if (d->sleg) server_release(d->sleg, false);
	{
		std::lock_guard<std::mutex> g(g_peer_mu);
		if (d->peer) { // (alive: had it died first it would have cleared this pointer below)
			std::vector<MSFilter *> &v = ((VolumeData *)d->peer->data)->peered_by;
			v.erase(std::remove(v.begin(), v.end(), f), v.end());
		}
		for (MSFilter *a : d->peered_by) {
			VolumeData *ad = (VolumeData *)a->data;
			ad->peer = NULL;
			ad->peer_gone.store(true, std::memory_order_release);
		}
		d->peered_by.clear();
	}
	if (d->pool && d->slot >= 0) {
		HubLock lk(f);
		d->pool->release(d->slot);
	}
	ms_bufferizer_destroy(d->buffer);
	ms_bufferizer_destroy(d->spill);
	ms_bufferizer_destroy(d->backlog);
	delete d;
}

// the slot's running state as of the last flush, before the slot is given up (hub locked)
void volume_keep_state(VolumeData *d) {
	if (!d->pool || d->slot < 0 || d->pool->failed) return;
	d->kept = d->pool->state[(size_t)d->slot];
	if (d->pool->state_dirty[(size_t)d->slot]) { // a gain method that has not reached the device yet
		const VolumePool::GainPatch &g = d->pool->gain_patch[(size_t)d->slot];
		if (g.also_gain) d->kept.gain = g.gain;
		if (g.also_target) d->kept.target_gain = g.target;
	}
	d->has_kept = true;
}
// what a new slot starts from: volume_init's state (msvolume.c:88-118) with the gains the methods set, or what the last slot held
mi_volume_state volume_start_state(const VolumeData *d) {
	mi_volume_state st;
	if (d->has_kept) return d->kept;
	memset(&st, 0, sizeof(st));
	st.gain = d->gain;
	st.target_gain = d->target_gain;
	st.ng_gain = 1;
	return st;
}

// MSVolume re-frames to 10 ms chunks with AGC or an echo-limiter peer (msvolume.c:480), else it takes every block as it is (:505)
bool volume_chunks(const VolumeData *d) { return d->p.agc_enabled != 0 || d->peer != NULL; }
void volume_framing_now(MSFilter *f) { ((VolumeData *)f->data)->chunks = volume_chunks((VolumeData *)f->data); }

// MSVolume as a METER and nothing else -- volrecv of a default AudioStream: no AGC, no gate, no DC removal, no peer, every gain exactly
// 1 -- leaves every sample as it came (volume_process :505-513 with a Q12 gain of 4096: the sample loop is skipped, msvolume.c:440).
// Where such a filter stands upstream of a canceller's far-end pin (audiostream.c:1812-1832) it hands its blocks on IN the walk, as the
// reference's does, and stages a copy for the meter: the far end of a call leg reaches the canceller in the walk it belongs to, where a
// block that came back with the next flush would find the microphone block it belongs to already cancelled against silence.  (Only
// there: in front of a mixer, a meter that is given a gain in mid-call would jump from no latency to a tick's and open a gap.)
// (its configuration alone: what the methods have set, whatever is still on its way to the device)
bool volume_meter_config(const VolumeData *d) {
	return !d->p.agc_enabled && !d->p.noise_gate_enabled && !d->p.remove_dc && d->peer == NULL && d->p.static_gain == 1.f && d->gain == 1.f && d->target_gain == 1.f;
}
FusedLeg *leg_fed_far_end_by(MSFilter *vol); // leg_chain.inl
// a method made this MSVolume more than a meter: a fused leg whose far end passes through it goes back to its facades (leg_far_end_in_walk)
void volume_far_end_changed(MSFilter *f, VolumeData *d) {
	if (volume_meter_config(d) || !f->outputs[0]) return;
	leg_disqualify(leg_fed_far_end_by(f));
}
bool volume_passes(const VolumeData *d) { // (hub locked)
	if (!d->feeds_far_end || d->chunks || d->p.agc_enabled || d->p.noise_gate_enabled || d->p.remove_dc || d->peer || d->p.static_gain != 1.f || !d->pool || d->slot < 0) return false;
	if ((ms_bufferizer_get_avail(d->spill) && !d->spill_quiet) || ms_bufferizer_get_avail(d->buffer)) return false;
	const VolumePool *p = d->pool;
	const size_t s = (size_t)d->slot;
	if (p->failed || p->params_dirty[s] == 2 || p->state_dirty[s] == 2 || (p->state_dirty[s] && !p->gain_patch[s].whole)) return false; // (a method's change on its way: wait for it)
	const mi_volume_state &st = p->state[s];
	return st.gain == 1.f && st.target_gain == 1.f && st.ng_gain == 1.f;
}

mi_volume_state *vstate(VolumeData *d) {
	if (d->leg) return leg_vstate(d->leg);
	if (d->meter_leg) return leg_pstate(d->meter_leg);
	if (d->sleg) return server_vstate(d->sleg);
	return (d->pool && d->slot >= 0) ? &d->pool->state[(size_t)d->slot] : nullptr;
}

void volume_push_params(VolumeData *d, bool f_method = true) {
	if (d->leg) {
		leg_push_volume(d->leg, &d->p, nullptr, nullptr);
		if (volume_chunks(d) != leg_frames_chunks(d->leg)) leg_disqualify(d->leg); // AGC switched: with it the reference meters 10 ms chunks, without it block by block -- another bank
		return;
	}
	if (d->meter_leg) { // a metered peer is a meter and nothing else: any other configuration goes back to a bank slot of its own
		leg_disqualify(d->meter_leg);
		return;
	}
	if (d->sleg) {
		server_push_volume(d->sleg, &d->p, nullptr, nullptr);
		if (d->p.agc_enabled) server_disqualify(d->sleg); // with AGC the reference meters re-framed 10 ms chunks: the facades' own banks
		return;
	}
	if (!d->pool || d->slot < 0) return;
	d->pool->params[(size_t)d->slot] = d->p;
	d->pool->params_dirty[(size_t)d->slot] = (f_method && d->pool->work_waiting()) ? 2 : 1;
	if (d->pool->params_dirty[(size_t)d->slot] == 1) d->chunks = volume_chunks(d);
}

void volume_attach_slot(MSFilter *f) {
	VolumeData *d = (VolumeData *)f->data;
	if (d->leg || d->sleg) return;
	if (d->pool) { // rate changed, moved to another ticker, or the bank failed: the slot goes back (under ITS hub's lock)
		HubLock old(f);
		if (d->pool->failed || d->pool->rate != d->sample_rate || d->pool->hub->ticker != f->ticker) {
			volume_keep_state(d);
			d->pool->release(d->slot);
			d->pool = nullptr;
			d->slot = -1;
		}
	}
	HubLock lk(f);
	bool fresh = false;
	if (!d->pool) {
		const int rate = d->sample_rate;
		d->pool = bank<VolumePool>("volume:" + std::to_string(rate), 1, [&](int cap) { return new VolumePool(cap, rate); });
		d->slot = d->pool ? d->pool->acquire(f) : -1;
		if (d->slot < 0) {
			d->pool = nullptr;
			return;
		}
		note_slot(f);
		// new slot: volume_init's state and whatever the methods set before attach, or the running state the last slot held
		d->pool->state[(size_t)d->slot] = volume_start_state(d);
		d->pool->state_dirty[(size_t)d->slot] = 1;
		d->pool->gain_patch[(size_t)d->slot] = VolumePool::GainPatch{1.f, 1.f, false, false, true};
		fresh = !d->pool->used[(size_t)d->slot];
		d->pool->used[(size_t)d->slot] = 1;
	}
	// the peer is addressed by its slot in the same pool
	d->p.peer = -1;
	if (d->peer) {
		VolumeData *pd = (VolumeData *)d->peer->data;
		if (pd->pool == d->pool && pd->slot >= 0) d->p.peer = pd->slot;
		else ms_warning("MSVolume[mi355x]: peer not in the same batch yet (different rate or not attached)");
	}
	volume_push_params(d, false); // (a slot's first parameters: at once)
	if (fresh && d->pool && d->slot >= 0) { // a slot nobody has used: what mi_volume_create left there may be exactly what this filter starts with
		VolumePool *p = d->pool;
		const size_t s = (size_t)d->slot;
		if (p->params_dirty[s] == 1 && memcmp(&p->params[s], &p->params0, sizeof(p->params0)) == 0) p->params_dirty[s] = 0;
		if (p->state_dirty[s] == 1 && p->gain_patch[s].whole && memcmp(&p->state[s], &p->state0, sizeof(p->state0)) == 0) {
			p->state_dirty[s] = 0;
			p->gain_patch[s] = VolumePool::GainPatch{1.f, 1.f, false, false, false};
		}
	}
}

bool equalizer_idle(MSFilter *g); // equalizer.inl: an MSEqualizer of ours that is switched off (it will hand its blocks on in the walk once attached)
void volume_preprocess(MSFilter *f) { // msvolume.c:447-469
	VolumeData *d = (VolumeData *)f->data;
	d->nsamples = (int)(0.01 * (float)d->sample_rate);
	d->min.reset();
	d->max.reset();
	d->feeds_far_end = false;
	{ // downstream through filters that are not ours (recv_tee ..) and a spk_equalizer that is not active (audiostream.c:1826-1829) to MSSpeexEC's pin 0?
		MSQueue *q = f->outputs[0];
		for (int hops = 0; q && hops < 12; ++hops) {
			MSFilter *g = q->next.filter;
			if (!g) break;
			if (g->desc == &ms_mi355x_speex_ec_desc || g->desc == &ms_mi355x_webrtc_aec_name_desc) {
				d->feeds_far_end = q->next.pin == 0;
				break;
			}
			if ((is_ours(g->desc) && !equalizer_idle(g)) || g->desc->noutputs != 1) break;
			q = g->outputs[0];
		}
	}
	// (no bank slot yet: a filter that joins a fused leg or conference at the attach never needs one of its own -- process() takes it
	// at the first block of a filter that did not fuse; a slot held from before the detach is kept or re-homed there as well.  But an
	// echo limiter's pair: the limiter addresses its peer by its slot in the SAME bank (volume_attach_slot), and two filters that take
	// their slots one after the other at the attach land side by side)
	if (volume_is_peered(d)) volume_attach_slot(f);
	if (!graph_ready(f)) return;
	HubLock lk(f);
	graph_preprocessed(f);
}

void volume_process(MSFilter *f) { // msvolume.c:471-514
	VolumeData *d = (VolumeData *)f->data;
	if (d->leg) { // fused leg: the chunks are popped, metered and mixed on the device; nothing arrives on this queue
		ms_queue_flush(f->inputs[0]);
		return;
	}
	if (!d->sleg && !d->fuse_checked && f->ticker && f->inputs[0] && !ms_queue_empty(f->inputs[0])) {
		// the first block since the attach: is this volrecv in front of a conference mixer whose members are all of that shape?
		d->fuse_checked = true;
		MSFilter *mx = leg_volume_sink(f);
		if (mx && mx->desc == &ms_mi355x_audio_mixer_desc) {
			HubLock lk(f, d->pool);
			conf_try_fuse(mx);
		}
	}
	if (d->meter_leg && leg_wants_out(d->meter_leg)) leg_release(d->meter_leg, true); // (it stands upstream of the leg's canceller: often the first of the leg to be walked)
	if (d->meter_leg) { // the echo-limiter peer of a fused leg: handed on as it came, a copy staged for the leg's meter
		HubLock lk(f, leg_pool(d->meter_leg));
		if (d->meter_leg) {
			leg_stage_peer(f, d);
			return;
		}
	}
	if (d->sleg && server_wants_out(d->sleg)) server_release(d->sleg, true); // a member stopped qualifying: the first of them to be walked takes the conference out, before anything of this walk is staged
	if (d->sleg) { // a conference server's member: the block goes into the leg's row of the conference's bank
		HubLock lk(f, server_pool(d->sleg));
		server_stage(f, d);
		return;
	}
	HubLock lk(f, d->pool);
	if (!d->pool) volume_attach_slot(f);
	if (!d->pool) {
		ms_queue_flush(f->inputs[0]);
		return;
	}
	if ((d->peer && d->p.peer < 0) || d->peer_gone.exchange(false, std::memory_order_acq_rel)) volume_attach_slot(f);
	VolumePool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	mblk_t *m;
	if (d->chunks) { // :480-503 re-framed to 10 ms chunks
		const size_t nbytes = (size_t)d->nsamples * 2;
		ms_bufferizer_put_from_queue(d->buffer, f->inputs[0]);
		while (ms_bufferizer_get_avail(d->buffer) >= nbytes) {
			if (p->staged[s] >= kMaxRounds) { // a burst of more chunks than launch rounds: what is staged goes out now
				p->flush();
				p->emit_all();
			}
			ms_bufferizer_read(d->buffer, (uint8_t *)(p->h_buf + (p->staged[s] * c + s) * p->cap_samples), nbytes);
			p->quiet[p->staged[s] * c + s] = 0;
			p->h_n[p->staged[s] * c + s] = d->nsamples;
			p->staged[s]++;
		}
	} else { // :505-512 light path: one chunk per mblk.  A block longer than a batch row (20 ms and more than 960 samples)
		// is cut into row-sized chunks -- no sample is dropped; the meter then sees those chunks, not the whole block.
		const bool pass = volume_passes(d); // a meter only: the blocks go on now, the rows are the meter's copies
		for (;;) {
			if (p->staged[s] >= kMaxRounds) {
				if (ms_bufferizer_get_avail(d->spill) == 0 && ms_queue_empty(f->inputs[0])) break;
				p->flush(); // more chunks than launch rounds in one tick: what is staged goes out now
				p->emit_all();
			}
			int16_t *row = p->h_buf + (p->staged[s] * c + s) * p->cap_samples;
			int n = 0;
			const size_t spilled = ms_bufferizer_get_avail(d->spill);
			bool row_quiet = pass;
			if (spilled) {
				n = (int)std::min(spilled / 2, (size_t)p->cap_samples);
				ms_bufferizer_read(d->spill, (uint8_t *)row, (size_t)n * 2);
				row_quiet = d->spill_quiet;
				if (ms_bufferizer_get_avail(d->spill) == 0) d->spill_quiet = false;
			} else if ((m = ms_queue_get(f->inputs[0])) != NULL) {
				n = (int)(msgdsize(m) / 2);
				if (n > p->cap_samples) {
					if (pass) { // (a copy is cut for the meter like any over-long block; the block itself goes on whole)
						mblk_t *cp = allocb((size_t)n * 2, 0);
						copy_payload(m, cp->b_wptr);
						cp->b_wptr += (size_t)n * 2;
						ms_bufferizer_put(d->spill, cp);
						d->spill_quiet = true;
						if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
						else freemsg(m);
					} else ms_bufferizer_put(d->spill, m); // served chunk by chunk from the top of the loop
					continue;
				}
				copy_payload(m, (uint8_t *)row);
				if (pass && f->outputs[0]) ms_queue_put(f->outputs[0], m);
				else freemsg(m);
			} else {
				break;
			}
			p->quiet[p->staged[s] * c + s] = row_quiet;
			p->h_n[p->staged[s] * c + s] = n;
			p->staged[s]++;
		}
	}
	if (p->staged[s]) request_flush(f);
}

void VolumePool::emit(MSFilter *f, int slot) {
	VolumeData *d = (VolumeData *)f->data;
	const size_t c = (size_t)capacity, s = (size_t)slot;
	for (int r = 0; r < ready[s]; ++r) {
		const int n = h_n[r * c + s];
		if (quiet[r * c + s]) continue; // (the block went on in the walk)
		mblk_t *om = allocb((size_t)n * 2, 0);
		memcpy(om->b_wptr, h_buf + (r * c + s) * cap_samples, (size_t)n * 2);
		om->b_wptr += n * 2;
		if (f->outputs[0]) ms_queue_put(f->outputs[0], om);
		else freemsg(om);
	}
	if (ready[s] && f->ticker) { // meters (update_energy msvolume.c:405-406): every chunk's energy, in order
		for (int r = 0; r + 1 < std::min(ready[s], rounds_fetched); ++r) {
			d->max.record_max(ticker_now(f->ticker), h_state[r * c + s].energy);
			d->min.record_min(ticker_now(f->ticker), h_state[r * c + s].energy);
		}
		d->max.record_max(ticker_now(f->ticker), state[s].energy);
		d->min.record_min(ticker_now(f->ticker), state[s].energy);
	}
	ready[s] = 0;
}

float linear_to_dbm0(float linear) { // ms_volume_linear_to_dbm0 msvolume.c:565-568
	if (linear == 0) return MS_VOLUME_DB_LOWEST;
	return 10 * log10f(linear);
}

int volume_get(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	mi_volume_state *st = vstate(d);
	*(float *)arg = linear_to_dbm0(st ? st->energy : 0.f);
	return 0;
}
int volume_get_linear(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	mi_volume_state *st = vstate(d);
	*(float *)arg = st ? st->energy : 0.f;
	return 0;
}
int volume_get_min(MSFilter *f, void *arg) {
	*(float *)arg = linear_to_dbm0(((VolumeData *)f->data)->min.current);
	return 0;
}
int volume_get_max(MSFilter *f, void *arg) {
	*(float *)arg = linear_to_dbm0(((VolumeData *)f->data)->max.current);
	return 0;
}
void volume_set_gains(MSFilter *f, VolumeData *d, bool also_target) {
	// methods run on the application's thread: the hub first (the ticker thread un-fuses and deletes legs under it), THEN d->leg
	HubLock lk(f);
	if (d->leg) {
		leg_push_volume(d->leg, &d->p, &d->gain, also_target ? &d->target_gain : nullptr);
		return;
	}
	if (d->meter_leg) { // (its running state comes back with the leg's un-fusing: the gains follow there)
		leg_disqualify(d->meter_leg);
		return;
	}
	if (d->sleg) {
		server_push_volume(d->sleg, &d->p, &d->gain, also_target ? &d->target_gain : nullptr);
		return;
	}
	if (d->has_kept) { // (a state waiting for its next slot follows the methods too)
		d->kept.gain = d->gain;
		if (also_target) d->kept.target_gain = d->target_gain;
	}
	if (!d->pool || d->slot < 0) { // no slot yet: volume_attach_slot picks d->gain / d->target_gain up
		volume_far_end_changed(f, d);
		return;
	}
	{
		VolumePool *p = d->pool;
		const size_t s = (size_t)d->slot;
		VolumePool::GainPatch &g = p->gain_patch[s];
		g.gain = d->gain, g.also_gain = true;
		if (also_target) g.target = d->target_gain, g.also_target = true;
		if (g.whole) { // (the slot's start state has not gone to the device yet: the method edits it)
			p->state[s].gain = d->gain;
			if (also_target) p->state[s].target_gain = d->target_gain;
		}
		if (p->state_dirty[s] != 1) p->state_dirty[s] = p->work_waiting() ? 2 : 1;
	}
	volume_push_params(d);
	volume_far_end_changed(f, d);
}
int volume_set_gain(MSFilter *f, void *arg) { // :270-276
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f); // (recursive: volume_set_gains takes it again; the ticker thread reads d->p and the gains under it)
	d->gain = d->target_gain = d->p.static_gain = *(float *)arg;
	volume_set_gains(f, d, true);
	return 0;
}
int volume_set_db_gain(MSFilter *f, void *arg) { // :262-268 (power ratio)
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->gain = d->p.static_gain = (float)pow(10, (*(float *)arg) / 10);
	volume_set_gains(f, d, false);
	return 0;
}
int volume_get_gain(MSFilter *f, void *arg) {
	*(float *)arg = ((VolumeData *)f->data)->p.static_gain;
	return 0;
}
int volume_get_gain_db(MSFilter *f, void *arg) {
	*(float *)arg = linear_to_dbm0(((VolumeData *)f->data)->p.static_gain);
	return 0;
}
int volume_set_peer(MSFilter *f, void *arg) { // :292-297 stores the MSFilter*
	VolumeData *d = (VolumeData *)f->data;
	MSFilter *peer = (MSFilter *)arg;
	{
		std::lock_guard<std::mutex> g(g_peer_mu);
		if (d->peer) {
			std::vector<MSFilter *> &v = ((VolumeData *)d->peer->data)->peered_by;
			v.erase(std::remove(v.begin(), v.end(), f), v.end());
		}
		d->peer = peer;
		if (peer) ((VolumeData *)peer->data)->peered_by.push_back(f);
	}
	if (peer) { // the echo limiter reads its peer's meter: both in one plain bank, or the peer metered beside a fused leg (set at the next attach)
		HubLock lk(peer);
		leg_disqualify(((VolumeData *)peer->data)->leg);
		leg_disqualify(((VolumeData *)peer->data)->meter_leg);
		server_disqualify(((VolumeData *)peer->data)->sleg);
	}
	HubLock lk(f);
	leg_disqualify(d->leg);
	server_disqualify(d->sleg);
	if (d->pool) volume_attach_slot(f);
	return 0;
}
int volume_set_rate(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	if (d->sample_rate != *(int *)arg) leg_disqualify(d->leg), server_disqualify(d->sleg);
	d->sample_rate = *(int *)arg;
	return 0;
}
#define VOL_FLOAT_SETTER(name, field, check)                       \
	int name(MSFilter *f, void *arg) {                             \
		VolumeData *d = (VolumeData *)f->data;                     \
		const float val = *(float *)arg;                           \
		if (!(check)) {                                            \
			ms_error("MSVolume: parameter out of range");          \
			return -1;                                             \
		}                                                          \
		HubLock lk(f); /* (the ticker thread reads d->p under it) */ \
		d->p.field = val;                                          \
		volume_push_params(d);                                     \
		volume_far_end_changed(f, d);                              \
		return 0;                                                  \
	}
VOL_FLOAT_SETTER(volume_set_ea_threshold, ea_thres, val >= 0 && val <= 1) // :305-314
VOL_FLOAT_SETTER(volume_set_ea_speed, vol_upramp, val >= 0 && val <= .5)  // :324-333
VOL_FLOAT_SETTER(volume_set_ea_force, force, true)
VOL_FLOAT_SETTER(volume_set_ea_transmit, ea_transmit_thres, true)
VOL_FLOAT_SETTER(volume_set_ng_threshold, ng_threshold, true)
int volume_set_ea_sustain(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->p.sustain_time = *(int *)arg;
	volume_push_params(d);
	volume_far_end_changed(f, d);
	return 0;
}
int volume_set_agc(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->p.agc_enabled = *(int *)arg;
	volume_push_params(d);
	volume_far_end_changed(f, d);
	return 0;
}
int volume_enable_noise_gate(MSFilter *f, void *arg) { // :352-359
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->p.noise_gate_enabled = *(bool_t *)arg;
	if (d->p.noise_gate_enabled) d->gain = d->target_gain = d->p.ng_floorgain;
	volume_set_gains(f, d, d->p.noise_gate_enabled != 0);
	return 0;
}
int volume_set_ng_floorgain(MSFilter *f, void *arg) { // :367-378
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->p.ng_floorgain = *(float *)arg;
	if (d->p.ng_floorgain < 0.005f) d->p.ng_floorgain = 0.005f;
	if (d->p.noise_gate_enabled) d->gain = d->target_gain = d->p.ng_floorgain;
	volume_set_gains(f, d, d->p.noise_gate_enabled != 0);
	return 0;
}
int volume_remove_dc(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	HubLock lk(f);
	d->p.remove_dc = *(int *)arg;
	volume_push_params(d);
	volume_far_end_changed(f, d);
	return 0;
}
MSFilterMethod volume_methods[] = {{MS_VOLUME_GET, volume_get},
                                   {MS_VOLUME_GET_LINEAR, volume_get_linear},
                                   {MS_VOLUME_SET_GAIN, volume_set_gain},
                                   {MS_VOLUME_SET_PEER, volume_set_peer},
                                   {MS_VOLUME_SET_EA_THRESHOLD, volume_set_ea_threshold},
                                   {MS_VOLUME_SET_EA_SPEED, volume_set_ea_speed},
                                   {MS_VOLUME_SET_EA_FORCE, volume_set_ea_force},
                                   {MS_VOLUME_SET_EA_SUSTAIN, volume_set_ea_sustain},
                                   {MS_VOLUME_SET_EA_TRANSMIT_THRESHOLD, volume_set_ea_transmit},
                                   {MS_FILTER_SET_SAMPLE_RATE, volume_set_rate},
                                   {MS_VOLUME_ENABLE_AGC, volume_set_agc},
                                   {MS_VOLUME_ENABLE_NOISE_GATE, volume_enable_noise_gate},
                                   {MS_VOLUME_SET_NOISE_GATE_THRESHOLD, volume_set_ng_threshold},
                                   {MS_VOLUME_SET_NOISE_GATE_FLOORGAIN, volume_set_ng_floorgain},
                                   {MS_VOLUME_SET_DB_GAIN, volume_set_db_gain},
                                   {MS_VOLUME_GET_GAIN, volume_get_gain},
                                   {MS_VOLUME_GET_GAIN_DB, volume_get_gain_db},
                                   {MS_VOLUME_REMOVE_DC, volume_remove_dc},
                                   {MS_VOLUME_GET_MIN, volume_get_min},
                                   {MS_VOLUME_GET_MAX, volume_get_max},
                                   {0, NULL}};
