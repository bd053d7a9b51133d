// filters/leg_chain.inl -- the call leg's chain of facades as ONE device-resident batch.
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the four facades it
// joins); not compiled on its own.
//
// The reference's sending graph of a call leg is  ... -> read_resampler -> ec -> volsend -> ... -> mixer
// (src/voip/audiostream.c:1798-1810), one MSFilter each, one process() each per tick.  Facade by facade that is four banks,
// four uploads and downloads and the audio crossing PCIe eight times.  When a conference mixer of this plugin finds that
// EVERY linked input pin is fed by   [MSResample (16k->48k, 8k->48k, 8k->16k) ->] MSSpeexEC pin 1 -> MSVolume -> pin
// -- with or without the resampler (the reference creates it only when the card's rate differs), with or without AGC (off
// unless the application asks), the same shape on every pin, all facades of this plugin on the mixer's ticker, freshly
// attached -- the whole conference moves into a LegBank; a leg whose MSVolume feeds anything else (an AudioStream's sending
// side) moves into a mixer-less one on its own:
//
//   * one slot per LEG (conference * mm + pin) shared by every per-leg object: the resampler's history, the canceller, its
//     three queues as device FIFOs (MSSpeexEC's `echo` and `delayed_ref` bufferizers, MSVolume's chunk bufferizer + the
//     mixer channel's), the meter;
//   * per tick and hub: the legs' 10 ms microphone blocks (320 B at 16 kHz) and far-end blocks (960 B) are staged in pinned
//     rows the launches read where they lie, TWO launches run per round -- aec_tick_kernel with the resampler folded in
//     (mi_aec_process_fifos_resampled_masked; mi_aec_process_fifos_masked without one) and volmix_kernel
//     (mi_mixer_process_volume_fifo_flags) -- and the conferences' mixes (960 B per leg) are written into a pinned slab whose
//     rows are handed downstream as they lie (esballoc + dupb: no copy, one data block per flush).  Without AGC MSVolume
//     meters the canceller's frames one by one (LegBank::light: a levelling launch per frame of the tick, a fourth queue);
//   * the facades' framing state machines stay on the host and decide as the reference does, on COUNTS: how many frames
//     MSSpeexEC's while loop runs (speexec.c:256), when it feeds silence into its delay line (:261-272) and what goes to
//     the speaker pin (host audio, as before), whether MSVolume has a whole chunk (msvolume.c:480-486), who contributes to
//     the mixer (audiomixer.c:244-286) and what its channels' flow control drops (:92-111).  The device FIFOs hold exactly
//     what those bufferizers would (MSMI355X_CHECK_LEVELS=1 reads the levels back every flush and compares).
//
// Results equal the facades run one by one (tests/test_gpu_plugin_fused.py: bit for bit, incl. far-end under-runs, 20 ms
// packets and a late joiner), with one stated exception: a conference with a single contributor is mixed like any other
// (the reference forwards that pin's blocks unsaturated, audiomixer.c:219-242: only a sample of -32768 differs; the
// contributor's own pin gets no block either way).
// Anything else -- another rate pair, pins of different shapes, MSVolume with an echo-limiter peer, non-conference mode,
// MSMI355X_NO_FUSE=1 -- keeps the facades on their own banks.  A fused conference falls back to them at run time when a
// member's configuration stops qualifying (bypass mode switched on, AGC switched, ...).
//
// MSAudioConference re-plumbs a conference around every join and leave (detach, link / unlink one endpoint, attach:
// src/voip/audioconference.c:322-374), so every member's filters see postprocess + preprocess.  What the reference's filters keep
// across that goes with the FILTERS, not with the bank slot (conf_unfuse / conf_try_fuse): the tick already out is waited for and
// delivered (deliver_in_flight), MSVolume's running state and the samples its bufferizer holds short of a chunk move to the next
// slot (leg_keep_volume / volume_start_state, take_remainders / give_remainder), MSResample's position and history too
// (resample_keep_from / resample_restore_to: the speex handle outlives a detach); the canceller starts over as its preprocess does.

constexpr int kLegRefOver = 3; // far-end ticks beyond the first that one flush carries per leg (a burst after a network hiccup)
constexpr int kLegLightRounds = 8; // frames MSVolume (no AGC) can meter in one enqueue: kMaxRounds blocks of 10 ms in frames
constexpr int kLegMeterRounds = 8; // rounds before the last of a flush whose meter state is read back (kLegLightRounds, kLegMaxChunks fit)
// 10 ms chunks a conference member's result queue is sized for BESIDES a tick's bursts (LegBank::out_cap).  A running batch keeps one or two;
// what a re-plumbed conference brings back (take_remainders) grows by up to one with every detach + attach -- the reference's mixer skips
// the walk in which none of its pins delivers (audiomixer.c:244-286: a restarted canceller's first frame may not complete a chunk) and its
// channels' flow control only trims after 5 s (:92-111) -- so a conference an application re-plumbs many times within seconds comes back
// with several.  The queue takes kLegHeldChunks - 4 of them at an attach (leg_candidate) and keeps the four a running batch was sized
// for.  (It was 4 in all, and the attach's bound the whole queue: PLUGIN_BENCH_CHURN brought conferences back with seven chunks in a
// queue of eight and a half, the canceller's next two frames found no room -- the kernel then runs nothing for the leg and counts it --
// and the host's framing had counted them: MSMI355X_CHECK_LEVELS, round 6.)
constexpr int kLegHeldChunks = 16;
constexpr int kLegMaxChunks = 5; // 10 ms chunks MSVolume can complete in one flush of a leg without a mixer (kMaxRounds blocks of 10 ms + what it held)

// The walk of a ticker with thousands of legs touches a few KB of scattered host memory per leg (filters, queues, blocks, the
// facades' states) -- tens of MB per ticker, far more than a core's caches hold from one tick to the next: the walk is a chain
// of cache misses.  A leg's facades therefore ask for the NEXT leg's lines while they work on their own (legs are walked in the
// order they were attached, which is the order of their slots).  MSMI355X_NO_PREFETCH=1: off (A/B).
inline bool leg_prefetch_on() {
	static const bool on = getenv("MSMI355X_NO_PREFETCH") == nullptr;
	return on;
}
// how many legs ahead the facades ask for (MSMI355X_PREFETCH_AHEAD, default 3): a leg's walk takes ~1.3 us, a miss on a busy host up to
// ~1 us -- one leg ahead is too late there (profiles/r05_prefetch_ahead.txt: 49 152 legs, three interleaved runs each on a box with load
// 30-40: 1.55 us per leg-tick and 4-21 late ticks of 400 at 1, 1.39 us and 0-5 at 3; within the noise at 32 768)
inline int leg_prefetch_ahead() {
	static const int n = [] {
		const char *e = getenv("MSMI355X_PREFETCH_AHEAD");
		const int v = e ? atoi(e) : 3;
		return v < 1 ? 1 : (v > 8 ? 8 : v);
	}();
	return n;
}
inline void pf(const void *p) {
	if (p) __builtin_prefetch(p, 0, 1);
}
inline void pf2(const void *p, size_t bytes) { // a struct or row of `bytes`
	if (!p) return;
	for (size_t o = 0; o < bytes; o += 64) __builtin_prefetch(static_cast<const char *>(p) + o, 0, 1);
}

// A conference leaves its device-resident batch, or joins one: what the reference's filters hold ACROSS a detach -- the mixer
// channels' bufferizers (audiomixer.c:64-76,132-135: channel_prepare / channel_unprepare touch only the tick buffer and the
// clocks; the queues go at uninit), MSVolume's bufferizer (msvolume.c has no postprocess) -- moves between the device queues and
// the filters' own host bufferizers.  Rare and synchronous.
// take: streams [s0, s0 + want.size()) hand want[k] of their samples to got(), in pieces of at most `row` samples: one
// all-or-nothing pop per distinct piece length and round (d_rows: [nlegs][row], d_gate: [nlegs], both idle between flushes)
bool fifo_take(mi_ctx *ctx, mi_fifo *fifo, int nlegs, int s0, int row, int16_t *d_rows, uint8_t *d_gate, std::vector<int> want,
               const std::function<void(int, const int16_t *, int)> &got) {
	const int count = (int)want.size();
	std::vector<uint8_t> gate((size_t)nlegs);
	std::vector<int16_t> rows((size_t)count * (size_t)row);
	for (;;) {
		std::vector<int> lens;
		for (int k = 0; k < count; ++k) {
			const int n = std::min(want[(size_t)k], row);
			if (n > 0 && std::find(lens.begin(), lens.end(), n) == lens.end()) lens.push_back(n);
		}
		if (lens.empty()) return true;
		for (int n : lens) {
			std::fill(gate.begin(), gate.end(), 0);
			bool any = false;
			for (int k = 0; k < count; ++k) any |= (gate[(size_t)(s0 + k)] = std::min(want[(size_t)k], row) == n) != 0;
			if (!any) continue;
			if (mi_copy_h2d(ctx, d_gate, gate.data(), (size_t)nlegs) != MI_OK || mi_fifo_pop(fifo, n, d_rows, row, nullptr, d_gate, 0) != MI_OK ||
			    mi_copy_d2h(ctx, rows.data(), d_rows + (size_t)s0 * row, rows.size() * 2) != MI_OK || mi_ctx_sync(ctx) != MI_OK)
				return false;
			for (int k = 0; k < count; ++k)
				if (gate[(size_t)(s0 + k)]) {
					got(s0 + k, rows.data() + (size_t)k * row, n);
					want[(size_t)k] -= n;
				}
		}
	}
}
// give: n samples appended to stream s (d_cnt: [nlegs] int32, idle)
bool fifo_give(mi_ctx *ctx, mi_fifo *fifo, int nlegs, int s, int row, int16_t *d_rows, int32_t *d_cnt, const int16_t *x, int n) {
	std::vector<int32_t> cnt((size_t)nlegs, 0);
	for (int at = 0; at < n; at += row) {
		const int k = std::min(row, n - at);
		cnt[(size_t)s] = k;
		if (mi_copy_h2d(ctx, d_rows + (size_t)s * row, x + at, (size_t)k * 2) != MI_OK || mi_copy_h2d(ctx, d_cnt, cnt.data(), (size_t)nlegs * 4) != MI_OK ||
		    mi_fifo_push(fifo, d_rows, row, row, d_cnt) != MI_OK || mi_ctx_sync(ctx) != MI_OK)
			return false;
	}
	return true;
}
void bufferizer_put_samples(MSBufferizer *bz, const int16_t *x, int n) {
	mblk_t *m = allocb((size_t)n * 2, 0);
	memcpy(m->b_wptr, x, (size_t)n * 2);
	m->b_wptr += n * 2;
	ms_bufferizer_put(bz, m);
}

struct LegBank;
struct FusedLeg {
	LegBank *bank;
	int slot, pin;
	uint32_t lv_from = 0; // MSMI355X_CHECK_LEVELS: the first read-back of the queues' levels (LegBank::lv_seq) that is this leg's -- an earlier one shows the slot before it was taken
	MSFilter *rs, *ec, *vol, *mixer;
	void *rs_data = nullptr, *ec_data = nullptr, *vol_data = nullptr; // the facades' states (f->data), known here so that a prefetch needs no cold load
	int staged_mic = 0;  // 10 ms blocks staged since the last flush (launch rounds)
	int framed_mic = 0;  // ... of which the framing below has already counted (a leg headed by MSSpeexEC frames as it stages)
	int pre_frames = 0;  // frames that framing found since the last flush
	int staged_ref = 0;  // far-end samples staged since the last flush
	int inject = 0;      // samples of silence the framing put behind them (speexec.c:261-272)
	int echo_level = 0;  // MSSpeexEC's `echo` bufferizer, samples: what f_mic holds
	int dref_level = 0;  // its `delayed_ref`: what f_ref holds, staged samples included
	int vol_rem = 0;     // MSVolume's bufferizer: cleaned samples short of a 10 ms chunk
	int chan_chunks = 0; // the mixer channel's bufferizer: whole chunks waiting (f_out holds vol_rem + chan_chunks * ns)
	int newchunks = 0;   // chunks MSVolume would have put on the mixer's queue in this flush
	// MSVolume WITHOUT AGC (LegBank::light): the canceller's frames pass it one by one, the mixer channel holds samples
	int lt_frames = 0;    // frames MSVolume meters in this enqueue
	int new_samples = 0;  // samples it put on the mixer's queue since the mixer last looked
	int chan_samples = 0; // the mixer channel's bufferizer, samples (what f_chan holds)
	bool metered = false;
	// MSVolume's echo limiter (msvolume.c:201-238): volsend reads the energy of its peer, volrecv -- which this leg METERS beside its
	// chain (LegBank::vol_peer): the peer facade hands its blocks on untouched in the walk and stages a copy of each for the meter
	MSFilter *eq = nullptr; // mic_equalizer between the leg's MSResample and MSSpeexEC (audiostream.c:1801): runs in the bank (LegBank::eq)
	// the G.711 encoder behind the leg's MSVolume (audiostream.c:1803-1809: volsend -> [outbound_mixer, forwarding] -> encoder), a leg
	// without a conference mixer: its chunks are ENCODED in the batch (LegBank::enc_law), 80 bytes of G.711 per 10 ms come back
	// instead of 160 of PCM and the encoder facade only packs them to its ptime (enc_take_codes, alaw.c:56-90)
	MSFilter *enc = nullptr, *omix = nullptr;
	MSFilter *peer = nullptr;
	int peer_staged = 0; // blocks of the peer staged since the last launch
	bool peer_metered = false;
	std::atomic<bool> unfuse_wanted{false}; // (a leg without a mixer: set by a method on any thread under the hub's lock, honoured by the head's next process())
	uint32_t far_tick = 0;      // ticker tick in which the far end was last taken (a bank without mixers leaves early on these)
};

struct MixSlab { // one flush's conference mixes in pinned memory, referenced by the blocks handed downstream
	std::atomic<int> state{0}; // 0 free, 1 a data block is alive on it, 2 its bank is gone (freed when the block goes)
	size_t bytes = 0;
	uint8_t *payload() { return reinterpret_cast<uint8_t *>(this) + 64; }
	static MixSlab *of(void *payload) { return reinterpret_cast<MixSlab *>(static_cast<uint8_t *>(payload) - 64); }
};
static_assert(sizeof(MixSlab) <= 64, "slab header");
void mix_slab_release(void *payload) { // db_freefn of the slab's data block (the last dupb of a flush was freed)
	MixSlab *s = MixSlab::of(payload);
	if (s->state.exchange(0, std::memory_order_acq_rel) == 2) mi_host_free(nullptr, s);
}

// MSSpeexEC's speaker-pin frames of one flush (one frame per microphone frame, speexec.c:261-284): cut from ONE host buffer per
// bank and flush -- esballoc for the buffer, a dupb per frame pointing at its piece -- instead of an allocb per frame (two
// allocations and a memset each, ~2 per leg and tick).  Plain host memory: the device never sees it.
struct SpkSlab {
	std::atomic<int> state{0}; // 0 free, 1 a data block is alive on it, 2 its bank is gone (freed when the block goes)
	size_t bytes = 0, used = 0;
	uint8_t *payload() { return reinterpret_cast<uint8_t *>(this) + 64; }
	static SpkSlab *of(void *payload) { return reinterpret_cast<SpkSlab *>(static_cast<uint8_t *>(payload) - 64); }
};
static_assert(sizeof(SpkSlab) <= 64, "slab header");
void spk_slab_release(void *payload) {
	SpkSlab *s = SpkSlab::of(payload);
	if (s->state.exchange(0, std::memory_order_acq_rel) == 2) free(s);
}

int channel_flow_control_level(Channel *chan, int level, int threshold, uint64_t now); // mixer.inl
void enc_take_codes(MSFilter *e, const uint8_t *codes, int n); // server_leg.inl
bool is_g711_enc(const MSFilterDesc *d);
bool is_forwarding_mixer(MSFilter *g, MSTicker *ticker); // recv_leg.inl
void leg_speaker_frame(MSFilter *f, SpeexECState *s, FusedLeg *leg, size_t nbytes, bool immediate);
void conf_unfuse(MSFilter *mixer, bool keep_running);
void leg_conf_walked(LegBank *b, int c);
void leg_far_walked(LegBank *b, FusedLeg *leg);

struct LegBank : Pool {
	uint32_t in_rate, rate;
	int F, flen, delay, mm, ns, in_len, den, nlegs;
	int mic_cap, ref_cap, out_cap;
	mi_resampler *rs = nullptr;
	mi_aec *aec = nullptr;
	mi_fifo *f_mic = nullptr, *f_ref = nullptr, *f_out = nullptr;
	mi_volume *vol = nullptr;
	mi_mixer *mix = nullptr;
	int16_t *h_mic, *d_mic;     // [kMaxRounds][nlegs][in_len] / [nlegs][in_len]
	uint8_t *h_gate, *d_gate;   // [kMaxRounds][nlegs] / [nlegs]
	int16_t *h_ref, *d_ref;     // [nlegs][ns]: the first tick of far end a leg staged
	int16_t *h_refx, *d_refx;   // [nlegs][kLegRefOver * ns]: what came beyond it (rare)
	int32_t *h_cnt, *d_cnt;     // [3][nlegs]: far-end samples in h_ref, in h_refx, silence injected behind them
	int32_t *d_zero;            // [nlegs] zeros: the canceller's launch appends no far end of its own
	int16_t *d_mix, *d_scratch; // [capacity][mm][ns]; [nlegs][ns]
	int32_t *h_lv, *d_lv;       // MSMI355X_CHECK_LEVELS: [3][nlegs]
	mi_volume_state *h_vstate;  // pinned [nlegs]
	// MSVolume records EVERY chunk's energy in its extrema (update_energy, msvolume.c:405-406): when a flush levels more than one
	// chunk of a leg, the state behind each round but the last comes back too (kLegMeterRounds rows of [nlegs]; vhas: the leg had a
	// chunk in that round)
	mi_volume_state *h_vround = nullptr;
	std::vector<uint8_t> vhas;
	int vrounds = 0;
	// the legs' echo-limiter peers (FusedLeg::peer), metered block by block as volrecv without AGC meters (msvolume.c:505-513) BEFORE the
	// chain's MSVolume runs in the same enqueue -- volsend reads what volrecv's process() of the same tick left, as in the reference,
	// where volrecv stands upstream of the canceller (audiostream.c:1812-1826).  Created when the first such leg joins.
	// A bank whose legs carry a mic_equalizer (audiostream.c:1801: between read_resampler and ec): the canceller's launch cannot fold the
	// up-sampler in then -- the round's microphone rows are up-sampled (MSResample's own kernel on the leg's state), equalized where they
	// lie (one FIR block per microphone block, equalizer.c:279-288) and handed to the canceller's launch as blocks at its rate: two
	// launches more per round, everything still device-resident
	mi_equalizer *eq = nullptr;
	int16_t *d_up = nullptr;                    // [nlegs][ns] a round's up-sampled, equalized microphone blocks
	int32_t *h_ecnt = nullptr, *d_ecnt = nullptr; // [kMaxRounds][nlegs]: ns where the leg has a microphone block in that round, else 0
	std::vector<EqualizerPool::Op> eq_later;    // methods that wait for the coming flush (Pool::work_waiting)
	std::vector<uint8_t> eq_used;               // the slot's FIR memory may hold an earlier leg's samples (a batch is created cleared: a first user needs no clearing)
	mi_volume *vol_peer = nullptr;
	int pcap = 0;                    // samples of a staged peer block (longer ones are cut, as the facade's light path cuts them)
	int16_t *h_pk = nullptr, *d_pk = nullptr;   // [kMaxRounds][nlegs][pcap] pinned; [nlegs][pcap]
	int32_t *h_pn = nullptr, *d_pn = nullptr;   // [kMaxRounds][nlegs]; [nlegs]
	mi_volume_state *h_pround = nullptr;        // [kMaxRounds][nlegs]: the peers' meters behind every round
	std::vector<mi_volume_state> pstate;        // the peers' running state as of the last flush
	int prounds = 0;                 // rounds of the launch that is out
	int npeers = 0;
	int16_t *h_copy;            // the mixes when every slab is still held downstream: emitted by copy
	std::vector<MixSlab *> slabs;
	MixSlab *cur = nullptr;     // the slab this flush downloads into (null: h_copy)
	mblk_t *root = nullptr;     // its data block, alive from finish() to emitted()
	std::vector<FusedLeg *> legs;
	std::vector<uint8_t> conf_ready;
	std::vector<int> lone; // the single contributor's pin of a conference that ticked with one, else -1
	std::vector<uint8_t> flags;
	std::vector<float> gains;
	bool ctl_dirty = true;
	// what a method set while the last walk's blocks were still waiting for the coming flush (Pool::work_waiting): live when that flush is
	// through (flushed()).  vp_dirty / vs_dirty: 1 = goes to the device with the next enqueue, 2 = waits for flushed() first
	std::vector<uint8_t> next_flags, next_conf;
	std::vector<float> next_gains;
	bool next_any = false;
	std::vector<mi_volume_params> vparams;
	std::vector<mi_volume_state> vstate;
	std::vector<uint8_t> vp_dirty, vs_dirty;
	// ... and in a conference with AGC the chunks MSVolume has already handed to the mixer's channel are levelled here only when the
	// mixer takes them (volmix_kernel pops, meters and mixes): what a method sets must pass those by -- v_delay: chunks of the leg
	// that were produced before the call and are still to be taken; the change goes to the device when it is down to zero
	std::vector<int> v_delay;
	bool v_dirty = false;
	std::vector<std::pair<int, int>> drops; // (leg slot, chunks) the mixer channels' flow control discards this flush
	std::vector<uint64_t> conf_time;        // ticker time of a conference's last tick (one per tick, whoever enqueues)
	std::vector<uint32_t> walk_tick;        // ticker tick in which a conference's mixer was last walked
	uint32_t walk_epoch = 0;
	uint8_t *h_run, *d_run;                 // [capacity]: conferences that tick in this launch of the volume + mix kernel
	bool staged_since = false;              // something was staged (or a conference joined) since the last enqueue
	bool outstanding = false;               // an enqueue has not been waited for yet
	// The staging rows and the mixes' slab are pinned host memory the device addresses itself: by default the launches read
	// and write them where they lie (a few hundred bytes per leg, once) and the tick path makes no copy at all -- four
	// launches and the meters' read-back.  MSMI355X_ZERO_COPY=0: staged through device buffers by copy launches (A/B).
	bool zero_copy = true;
	bool mixed = false, check_levels = false, lv_fresh = false;
	uint32_t lv_seq = 0; // read-backs of the levels so far (FusedLeg::lv_from)
	double trace_ms = 0;          // MSMI355X_TRACE_SLOW_MS: an enqueue that takes longer says where (stderr)
	uint64_t tr[8] = {0};
	std::vector<std::pair<const char *, uint64_t>> trc; // ... and call by call inside the device's half
	static uint64_t trace_now() {
		struct timespec ts;
		clock_gettime(CLOCK_MONOTONIC, &ts);
		return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
	}
	void mark(int i) {
		if (trace_ms > 0) tr[i] = trace_now();
	}
	void step(const char *what) {
		if (trace_ms > 0) trc.emplace_back(what, trace_now());
	}
	uint64_t launches = 0;
	std::vector<std::pair<MSQueue *, mblk_t *>> spk; // speaker-pin frames of this flush (MSSpeexEC pin 0: host audio), handed on in finish()
	std::vector<SpkSlab *> spk_slabs;                // ... cut from one of these (a ring: a slab returns when its last frame is freed downstream)
	SpkSlab *spk_cur = nullptr;
	mblk_t *spk_root = nullptr;
	int walked = 0;                                  // conferences whose mixer has run in this tick's graph walk
	bool early = false, early_any = false;           // this tick's work was enqueued at the end of the walk (leg_conf_walked)
	bool no_early = false;
	// A bank WITHOUT mixers (plain): legs  MSResample -> MSSpeexEC -> MSVolume -> any other filter  -- the sending side of an
	// AudioStream (audiostream.c:1798-1810) whose streams share a ticker.  One slot = one leg (mm = 1), owned by its MSVolume,
	// which hands the levelled 10 ms chunks on as they come out of mi_volume_process_fifo_flags (up to kLegMaxChunks a flush).
	bool plain = false;
	// A bank whose MSVolumes run WITHOUT AGC (the reference's default: audio_stream_enable_automatic_gain_control is off unless
	// asked for): volume_process then meters and levels every incoming block as it is -- the canceller's frames of F samples -- and
	// has no bufferizer (msvolume.c:505-513).  The frames are levelled where they lie in the output queue, one launch per frame
	// of the tick (mi_volume_process_fifo_flags on F samples), and go on to the mixer channel's queue on the device (f_chan), from
	// which the conference is mixed (volmix_kernel with an identity volume batch: pop + mix); without a mixer they are handed on
	// frame by frame (chunk = F).
	bool light = false;
	int enc_law = -1;              // plain: every leg's chunks leave through MSAlawEnc (MI_LAW_PCMA) / MSUlawEnc (MI_LAW_PCMU) of ours; -1: as PCM
	uint8_t *h_codes = nullptr, *d_codes = nullptr; // [kLegMaxChunks][nlegs][chunk]: the encoded chunks of this flush
	int32_t *h_elen = nullptr, *d_elen = nullptr;   // [kLegMaxChunks][nlegs]: chunk where the leg has a chunk in that round, else 0
	int chunk = 0;                 // samples of a block MSVolume hands on: 10 ms with AGC, a canceller frame without
	mi_volume *vol_id = nullptr;   // identity batch (gain 1, nothing enabled): volmix_kernel's volume half for the levelled queue
	mi_fifo *f_chan = nullptr;     // the mixer channels' bufferizers
	int16_t *d_lev = nullptr;      // [nlegs][F] a round's levelled frames
	int32_t *h_fcnt = nullptr, *d_fcnt = nullptr; // [kLegLightRounds][nlegs]: F where the leg has a frame in that round, else 0
	uint8_t *h_dgate = nullptr, *d_dgate = nullptr; // [nlegs]: the leg the channels' flow control drops samples of (rare)
	std::vector<std::pair<int, int>> sdrops;      // (leg slot, samples) of those drops
	std::vector<int> nout, nready; // chunks a leg's MSVolume completes in this flush / has ready to hand on
	struct GainPatch {
		float gain, target;
		bool also_target;
	};
	std::vector<GainPatch> vpatch; // MS_VOLUME_SET_GAIN & co. on a fused leg: the two fields, set on the state as the device holds it
	// A leg that joins the bank costs NO device call when its slot has never been used: every batch object is created with every
	// stream at its start state (mi_aec_create / mi_fifo_create / mi_resampler_create / mi_volume_create end in their own reset), so
	// only a slot that HAD a leg is reset; MSVolume's parameters and start state go up with the next enqueue, neighbours in one call,
	// and not at all where they are what the batch was created with; the delay line's zeroes (speexec.c:205-208) are pushed by the
	// first enqueue as silence behind nothing (FusedLeg::inject).  A ticker's thousands of legs fuse at an attach: each of those calls
	// waits for the stream, and took the attach into seconds.
	std::vector<uint8_t> used;
	mi_volume_params vparams0;
	mi_volume_state vstate0;

	static int frames_up(int v, int frame) { return (v + frame - 1) / frame * frame; }
	static int out_cap_for(int ns, int F) { return frames_up(kLegHeldChunks * ns + kMaxRounds * 2 * F, F); }
	LegBank(int cap_conf, uint32_t ir, uint32_t r, int frame, int filter_length, int delay_samples, int members, bool no_mixer = false, bool no_agc = false,
	        bool with_eq = false, int law = -1)
	    : in_rate(ir), rate(r), F(frame), flen(filter_length), delay(delay_samples), mm(members), plain(no_mixer), light(no_agc), enc_law(law) {
		Building b(this, cap_conf);
		ns = (int)rate / 100;
		chunk = light ? F : ns;
		in_len = (int)in_rate / 100;
		den = (int)(rate / in_rate);
		nlegs = capacity * mm;
		mic_cap = frames_up(2 * ns + 2 * F, F);
		out_cap = plain ? frames_up(4 * ns + kMaxRounds * 2 * F, F) : out_cap_for(ns, F); // (a leg without a mixer holds less than a chunk between blocks: leg_fuse_plain_at)
		ref_cap = frames_up(delay + (3 + kLegRefOver) * ns + kMaxRounds * 2 * F, F);
		if (!failed && in_rate != rate) MI_MUST(mi_resampler_create(hub->ctx, nlegs, in_rate, rate, 3, &rs)); // (no MSResample in front: the microphone arrives at the canceller's rate)
		if (!failed) MI_MUST(mi_aec_create(hub->ctx, nlegs, (int)rate, F, flen, &aec));
		if (!failed && with_eq) {
			MI_MUST(mi_equalizer_create(hub->ctx, nlegs, (int)rate, &eq));
			d_up = devmem<int16_t>((size_t)nlegs * ns);
			h_ecnt = pinned<int32_t>(kMaxRounds * (size_t)nlegs);
			d_ecnt = devmem<int32_t>(kMaxRounds * (size_t)nlegs);
		}
		if (!failed) MI_MUST(mi_fifo_create(hub->ctx, nlegs, mic_cap, &f_mic));
		if (!failed) MI_MUST(mi_fifo_create(hub->ctx, nlegs, ref_cap, &f_ref));
		if (!failed) MI_MUST(mi_fifo_create(hub->ctx, nlegs, out_cap, &f_out));
		if (!failed) MI_MUST(mi_volume_create(hub->ctx, nlegs, (int)rate, &vol));
		if (!failed && !plain) MI_MUST(mi_mixer_create(hub->ctx, capacity, mm, ns, &mix));
		const size_t L = (size_t)nlegs;
		if (light && !plain) {
			if (!failed) MI_MUST(mi_volume_create(hub->ctx, nlegs, (int)rate, &vol_id));
			if (!failed) MI_MUST(mi_fifo_create(hub->ctx, nlegs, out_cap, &f_chan));
			d_lev = devmem<int16_t>(L * F);
			h_fcnt = pinned<int32_t>(kLegLightRounds * L);
			d_fcnt = devmem<int32_t>(kLegLightRounds * L);
			h_dgate = pinned<uint8_t>(L);
			d_dgate = devmem<uint8_t>(L);
		}
		if (enc_law >= 0) {
			h_codes = pinned<uint8_t>((size_t)kLegMaxChunks * L * chunk);
			d_codes = devmem<uint8_t>((size_t)kLegMaxChunks * L * chunk);
			h_elen = pinned<int32_t>((size_t)kLegMaxChunks * L);
			d_elen = devmem<int32_t>((size_t)kLegMaxChunks * L);
		}
		h_mic = pinned<int16_t>(kMaxRounds * L * in_len);
		d_mic = devmem<int16_t>(L * in_len);
		h_gate = pinned<uint8_t>(kMaxRounds * L);
		d_gate = devmem<uint8_t>(kMaxRounds * L);
		h_ref = pinned<int16_t>(L * ns);
		d_ref = devmem<int16_t>(L * ns);
		h_refx = pinned<int16_t>(L * kLegRefOver * ns);
		d_refx = devmem<int16_t>(L * kLegRefOver * ns);
		h_cnt = pinned<int32_t>(3 * L);
		d_cnt = devmem<int32_t>(3 * L);
		d_zero = devmem<int32_t>(L);
		d_mix = devmem<int16_t>((plain ? kLegMaxChunks : 1) * L * ns);
		d_scratch = devmem<int16_t>(L * ns);
		h_lv = pinned<int32_t>(4 * L);
		d_lv = devmem<int32_t>(4 * L);
		h_vstate = pinned<mi_volume_state>(L);
		h_vround = pinned<mi_volume_state>((size_t)kLegMeterRounds * L);
		vhas.assign((size_t)kLegMeterRounds * L, 0);
		h_copy = pinned<int16_t>((plain ? kLegMaxChunks : 1) * L * ns);
		nout.assign(L, 0);
		nready.assign(L, 0);
		h_run = pinned<uint8_t>((size_t)capacity);
		d_run = devmem<uint8_t>((size_t)capacity);
		conf_time.assign((size_t)capacity, (uint64_t)-1);
		walk_tick.assign((size_t)capacity, 0);
		if (!failed) MI_MUST(mi_memset(hub->ctx, d_zero, 0, L * 4));
		legs.assign(L, nullptr);
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
		v_delay.assign(L, 0);
		vpatch.assign(L, GainPatch{1.f, 1.f, false});
		used.assign(L, 0);
		vparams0 = p;
		memset(&vstate0, 0, sizeof(vstate0));
		if (!failed) MI_MUST(mi_volume_get_state(vol, 0, 1, &vstate0));
		check_levels = getenv("MSMI355X_CHECK_LEVELS") != nullptr;
		if (const char *e = getenv("MSMI355X_TRACE_SLOW_MS")) trace_ms = atof(e);
		zero_copy = zero_copy_rows();
		no_early = getenv("MSMI355X_NO_EARLY_LAUNCH") != nullptr; // A/B switch: everything leaves at the flush
		// the first ticks' slabs are made here (the attaching thread opens the bank), not by those ticks: two of each -- a flush's blocks
		// are still held downstream when the next flush needs its slab
		for (int i = 0; i < 2 && !failed && enc_law < 0; ++i) {
			MixSlab *s = free_slab();
			if (s) s->state.store(1, std::memory_order_release); // (so that the second call makes a second one)
		}
		for (MixSlab *s : slabs) s->state.store(0, std::memory_order_release);
		for (int i = 0; i < 2 && !failed; ++i) (void)new_spk_slab();
	}
	~LegBank() override {
		if (root) freeb(root);
		for (auto &qm : spk) freemsg(qm.second);
		if (spk_root) freeb(spk_root);
		for (SpkSlab *s : spk_slabs)
			if (s->state.exchange(2, std::memory_order_acq_rel) == 0) free(s);
		for (FusedLeg *l : legs) delete l;
		if (hub->ctx) mi_ctx_sync(hub->ctx);
		if (mix) mi_mixer_destroy(mix);
		if (vol) mi_volume_destroy(vol);
		if (vol_peer) mi_volume_destroy(vol_peer);
		if (eq) mi_equalizer_destroy(eq);
		if (vol_id) mi_volume_destroy(vol_id);
		if (f_chan) mi_fifo_destroy(f_chan);
		for (mi_fifo *f : {f_mic, f_ref, f_out})
			if (f) mi_fifo_destroy(f);
		if (aec) mi_aec_destroy(aec);
		if (rs) mi_resampler_destroy(rs);
		for (MixSlab *s : slabs) // a slab whose blocks are still held downstream outlives the bank: its last block frees it
			if (s->state.exchange(2, std::memory_order_acq_rel) == 0) mi_host_free(hub->ctx, s);
	}
	// a frame of `nbytes` for the speaker pin: a piece of this flush's slab (NULL: no slab to be had, the caller allocates)
	mblk_t *spk_frame(size_t nbytes) {
		if (!spk_cur) {
			for (SpkSlab *s : spk_slabs)
				if (s->state.load(std::memory_order_acquire) == 0) {
					spk_cur = s;
					break;
				}
			if (!spk_cur && spk_slabs.size() < 6) spk_cur = new_spk_slab();
			if (!spk_cur) return nullptr;
			spk_cur->used = 0;
			spk_cur->state.store(1, std::memory_order_release);
			spk_root = esballoc(spk_cur->payload(), spk_cur->bytes, 0, spk_slab_release);
		}
		if (spk_cur->used + nbytes > spk_cur->bytes) return nullptr;
		mblk_t *m = dupb(spk_root);
		m->b_rptr = spk_cur->payload() + spk_cur->used;
		m->b_wptr = m->b_rptr + nbytes;
		spk_cur->used += nbytes;
		return m;
	}
	SpkSlab *new_spk_slab() {
		const size_t bytes = (size_t)nlegs * kMaxRounds * 2 * (size_t)F * 2; // every leg's frames of a burst of kMaxRounds blocks
		void *p = malloc(64 + bytes);
		if (!p) return nullptr;
		memset(p, 0, 64 + bytes); // (touched here, by whoever creates it -- the attaching thread for a bank's first -- not page by page in a tick)
		SpkSlab *s = new (p) SpkSlab();
		s->bytes = bytes;
		spk_slabs.push_back(s);
		return s;
	}
	void spk_flush_done() { // the flush's own reference: the slab returns to the ring when the last frame downstream is freed
		if (spk_root) freeb(spk_root);
		spk_root = nullptr;
		spk_cur = nullptr;
	}
	MixSlab *free_slab() {
		for (MixSlab *s : slabs)
			if (s->state.load(std::memory_order_acquire) == 0) return s;
		if (slabs.size() >= 4 || failed) return nullptr;
		const size_t bytes = (size_t)(plain ? kLegMaxChunks : 1) * nlegs * ns * 2;
		void *p = mi_host_alloc(hub->ctx, 64 + bytes);
		if (!p) return nullptr;
		MixSlab *s = new (p) MixSlab();
		s->bytes = bytes;
		slabs.push_back(s);
		return s;
	}

	// legs [s0, s0 + count) are about to be taken: their per-slot objects at the start state (see `used`)
	bool start_slots(int s0, int count) {
		bool any = false;
		for (int s = s0; s < s0 + count; ++s) any |= used[(size_t)s] != 0, used[(size_t)s] = 1;
		if (!any) return true;
		return (!rs || mi_resampler_reset(rs, s0, count) == MI_OK) && mi_aec_reset(aec, s0, count) == MI_OK && mi_fifo_reset_range(f_mic, s0, count) == MI_OK &&
		       mi_fifo_reset_range(f_ref, s0, count) == MI_OK && mi_fifo_reset_range(f_out, s0, count) == MI_OK && mi_volume_reset_max(vol, s0, count) == MI_OK &&
		       (!f_chan || mi_fifo_reset_range(f_chan, s0, count) == MI_OK);
	}
	// ... and leg s's MSVolume: parameters and running state for the slot, on the device with the next enqueue (or never: what the batch was created with)
	void start_volume(size_t s, const mi_volume_params &p, const mi_volume_state &st, bool was_used) {
		vparams[s] = p, vstate[s] = st;
		vpatch[s] = GainPatch{st.gain, st.target_gain, true};
		const bool same = !was_used && memcmp(&p, &vparams0, sizeof(p)) == 0 && memcmp(&st, &vstate0, sizeof(st)) == 0;
		vp_dirty[s] = vs_dirty[s] = same ? 0 : 1;
		v_delay[s] = 0;
		v_dirty |= !same;
	}

	// ---- the canceller's framing for everything a leg staged since the last flush: the while loop of speexec.c:256-305
	// immediate: called from MSSpeexEC's own process() (it is the leg's head): the speaker frames leave in that tick, as they do
	// from the reference's filter; otherwise (behind an MSResample, whose output reaches the canceller a tick later) with the flush
	int ec_frames(FusedLeg *leg, bool immediate = false) {
		SpeexECState *es = (SpeexECState *)leg->ec->data;
		leg->echo_level += (leg->staged_mic - leg->framed_mic) * ns;
		leg->framed_mic = leg->staged_mic;
		int nfr = 0;
		while (leg->echo_level >= F) {
			leg->echo_level -= F;
			es->echostarted = TRUE;
			leg_speaker_frame(leg->ec, es, leg, (size_t)F * 2, immediate);
			leg->dref_level -= F; // the frame the canceller reads from the head of the delay line
			++nfr;
		}
		return nfr;
	}

	// ---- one tick of a conference on counts: mixer_process (audiomixer.c:288-346) with the census of mixer_check_bypass
	// (:244-286) and the channels' flow control (:92-111), deciding from what MSVolume would have put on the pins' queues
	void conf_tick(int c, uint64_t now) {
		MSFilter *mx = owner[(size_t)c];
		MixerState *s = (MixerState *)mx->data;
		conf_ready[(size_t)c] = 0;
		lone[(size_t)c] = -1;
		int count = 0, who = -1;
		for (int pin = 0; pin < mm; ++pin) {
			FusedLeg *leg = legs[(size_t)(c * mm + pin)];
			if (!leg) continue;
			uint64_t &seen = s->channels[pin].last_activity;
			bool contributes;
			if (light ? leg->new_samples > 0 : leg->newchunks > 0) {
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
		if (count == 0) return; // nobody has delivered for a second: nothing leaves (and nothing was queued)
		if ((count == 1) != (s->bypass_mode != FALSE))
			ms_message("mi355x mixer %p: %s", (void *)mx, count == 1 ? "a single contributor (mixed on the device all the same)" : "two or more contributors");
		s->bypass_mode = count == 1;
		for (int pin = 0; pin < mm; ++pin) {
			FusedLeg *leg = legs[(size_t)(c * mm + pin)];
			if (!leg) continue;
			Channel *chan = &s->channels[pin];
			if (light) { // the channel's bufferizer holds the levelled frames, the tick reads 10 ms of them or nothing (:78-90)
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
				continue;
			}
			leg->chan_chunks += leg->newchunks; // ms_bufferizer_put_from_queue, channel_process_in :78-90
			leg->newchunks = 0;
			if (leg->chan_chunks > 0) { // ... and the read of one tick (the device pops it: the queue holds a whole chunk)
				leg->chan_chunks--;
				leg->metered = true;
				if (v_delay[(size_t)leg->slot] > 0) --v_delay[(size_t)leg->slot];
			}
			const int skip = channel_flow_control_level(chan, leg->chan_chunks * ns * 2, s->skip_threshold, now);
			if (skip > 0) {
				const int k = std::min(leg->chan_chunks, skip / (ns * 2));
				ms_warning("mi355x mixer: pin %i kept more than two ticks queued for 5 s; %i ms discarded", pin, k * 10);
				leg->chan_chunks -= k;
				if (k > 0) drops.push_back({leg->slot, k});
			}
		}
		conf_ready[(size_t)c] = 1;
		lone[(size_t)c] = count == 1 ? who : -1;
	}

	// the device's half up to the cleaned frames, in the order the reference's process() works: far end queued, then the frames
	bool enqueue_cancellers(bool any_ref, bool any_refx, bool any_inj, int rounds) {
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm;
		bool any = false;
		trc.clear();
		step("start");
		const bool zc = zero_copy;
		const int32_t *cnt = zc ? h_cnt : d_cnt;
		if (!zc && (any_ref || any_refx || any_inj)) MI_MUST(mi_copy_h2d_pinned(ctx, d_cnt, h_cnt, 3 * L * 4));
		step("counts up");
		if (any_ref) {
			if (!zc) MI_MUST(mi_copy_h2d_pinned(ctx, d_ref, h_ref, UL * ns * 2));
			step("far end up");
			MI_MUST(mi_fifo_push(f_ref, zc ? h_ref : d_ref, ns, ns, cnt));
			step("far end queued");
			++launches, any = true;
		}
		if (any_refx) {
			if (!zc) MI_MUST(mi_copy_h2d_pinned(ctx, d_refx, h_refx, UL * kLegRefOver * ns * 2));
			MI_MUST(mi_fifo_push(f_ref, zc ? h_refx : d_refx, kLegRefOver * ns, kLegRefOver * ns, cnt + L));
			++launches, any = true;
		}
		if (any_inj) {
			MI_MUST(mi_fifo_push_silence(f_ref, cnt + 2 * L));
			++launches, any = true;
		}
		if (rounds && !zc) MI_MUST(mi_copy_h2d_pinned(ctx, d_gate, h_gate, (size_t)rounds * L));
		if (rounds && !zc && eq) MI_MUST(mi_copy_h2d_pinned(ctx, d_ecnt, h_ecnt, (size_t)rounds * L * 4));
		step("gates up");
		for (int r = 0; r < rounds; ++r) {
			const int16_t *mic_r = h_mic + (size_t)r * L * in_len;
			if (!zc) MI_MUST(mi_copy_h2d_pinned(ctx, d_mic, mic_r, UL * in_len * 2));
			step("microphones up");
			if (eq) { // (a bank with equalizers has an MSResample in every leg: leg_candidate)
				const uint8_t *gate_r = (zc ? h_gate : d_gate) + (size_t)r * L;
				MI_MUST(mi_resampler_process_masked(rs, zc ? mic_r : d_mic, in_len, in_len, d_up, ns, nullptr, gate_r));
				MI_MUST(mi_equalizer_process_masked(eq, d_up, ns, ns, (zc ? h_ecnt : d_ecnt) + (size_t)r * L));
				MI_MUST(mi_aec_process_fifos_masked(aec, f_mic, d_up, ns, f_ref, d_ref, ns, d_zero, ns, f_out, MI_AEC_MAX_TICK_FRAMES, MI_AEC_POSTFILTER, nullptr, gate_r));
				launches += 2;
			} else if (rs)
				MI_MUST(mi_aec_process_fifos_resampled_masked(aec, rs, zc ? mic_r : d_mic, in_len, in_len, f_mic, f_ref, d_ref, ns, d_zero, f_out, MI_AEC_MAX_TICK_FRAMES,
				                                              MI_AEC_POSTFILTER, nullptr, (zc ? h_gate : d_gate) + (size_t)r * L));
			else // the microphone block as it came: queued by the same launch, no up-sampler in front
				MI_MUST(mi_aec_process_fifos_masked(aec, f_mic, zc ? mic_r : d_mic, in_len, f_ref, d_ref, ns, d_zero, ns, f_out, MI_AEC_MAX_TICK_FRAMES,
				                                    MI_AEC_POSTFILTER, nullptr, (zc ? h_gate : d_gate) + (size_t)r * L));
			step("cancellers launched");
			launches += 2, any = true; // (the canceller's launch and the turn-over of its leg lists behind it)
		}
		return any;
	}
	// Legs [s0, s0 + count) leave the bank: what the reference's filters would still hold goes back to them.
	//  - The chunks waiting whole in f_out have passed MSVolume in the reference and sit in the mixer channel's bufferizer, which
	//    outlives a detach (audiomixer.c:64-76,132-135,200-208: postprocess frees the tick buffers, the queues go at uninit).  When the
	//    facades carry on one by one (keep_running) they are levelled now and put into that bufferizer (MixerState::channels, which the
	//    facade's mixer reads); at a detach they go back RAW, in front of the samples short of a chunk, into MSVolume's bufferizer
	//    (msvolume.c:480-486; no postprocess touches it): the attach that follows fuses again and the queue is rebuilt as it was
	//    (give_remainder) -- levelled when the mixer takes them, as everything in this bank is.
	//  - Without AGC (light) the channel's queue is f_chan, levelled already: into the mixer channel's bufferizer either way.
	void take_remainders(int s0, int count, bool keep_running) {
		if (failed || (plain && light)) return; // (without AGC MSVolume holds nothing between blocks; a leg without a mixer has no channel)
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs;
		auto channel_of = [&](int s) -> MSBufferizer * {
			FusedLeg *leg = legs[(size_t)s];
			return &((MixerState *)leg->mixer->data)->channels[leg->pin].bufferizer;
		};
		// the queues come back in ONE round trip for the whole range (mi_fifo_export_range: a conference that is re-plumbed with every
		// join and leave, audioconference.c:322-374, used to pay one pop, copy and wait per member and piece length)
		std::vector<int32_t> lvl((size_t)count, 0);
		std::vector<int16_t> x((size_t)count * (size_t)out_cap);
		if (light) {
			if (mi_fifo_export_range(f_chan, s0, count, x.data(), out_cap, lvl.data()) != MI_OK || mi_fifo_reset_range(f_chan, s0, count) != MI_OK) {
				mi_failed("taking the mixer channels' queues back");
				return;
			}
			for (int s = s0; s < s0 + count; ++s)
				if (FusedLeg *leg = legs[(size_t)s]) {
					const int n = std::min(lvl[(size_t)(s - s0)], leg->chan_samples + leg->new_samples);
					if (n > 0) bufferizer_put_samples(channel_of(s), x.data() + (size_t)(s - s0) * out_cap, n);
					leg->chan_samples = leg->new_samples = 0;
				}
			return;
		}
		if (keep_running) {
			int rounds = 0;
			for (int s = s0; s < s0 + count; ++s)
				if (legs[(size_t)s]) rounds = std::max(rounds, legs[(size_t)s]->chan_chunks + legs[(size_t)s]->newchunks);
			std::vector<int16_t> rows((size_t)count * chunk);
			for (int k = 0; k < rounds; ++k) {
				for (int s = s0; s < s0 + count; ++s)
					if (legs[(size_t)s] && legs[(size_t)s]->chan_chunks + legs[(size_t)s]->newchunks > k) MI_MUST(mi_volume_process_fifo_range(vol, f_out, d_scratch, chunk, chunk, s, 1));
				MI_MUST(mi_copy_d2h(ctx, rows.data(), d_scratch + (size_t)s0 * chunk, rows.size() * 2));
				sync_stream();
				if (failed) return;
				for (int s = s0; s < s0 + count; ++s)
					if (legs[(size_t)s] && legs[(size_t)s]->chan_chunks + legs[(size_t)s]->newchunks > k) bufferizer_put_samples(channel_of(s), rows.data() + (size_t)(s - s0) * chunk, chunk);
			}
			if (rounds) {
				MI_MUST(mi_volume_get_state_async(vol, 0, (int)L, h_vstate));
				sync_stream();
				if (failed) return;
				for (int s = s0; s < s0 + count; ++s)
					if (legs[(size_t)s] && !vs_dirty[(size_t)s]) vstate[(size_t)s] = h_vstate[s];
			}
			for (int s = s0; s < s0 + count; ++s)
				if (legs[(size_t)s]) legs[(size_t)s]->chan_chunks = legs[(size_t)s]->newchunks = 0;
		}
		if (mi_fifo_export_range(f_out, s0, count, x.data(), out_cap, lvl.data()) != MI_OK || mi_fifo_reset_range(f_out, s0, count) != MI_OK) {
			mi_failed("taking MSVolume's queued samples back");
			return;
		}
		for (int s = s0; s < s0 + count; ++s)
			if (FusedLeg *leg = legs[(size_t)s]) {
				const int n = std::min(lvl[(size_t)(s - s0)], (leg->chan_chunks + leg->newchunks) * chunk + leg->vol_rem);
				if (n > 0) bufferizer_put_samples(((VolumeData *)leg->vol->data)->buffer, x.data() + (size_t)(s - s0) * out_cap, n);
				leg->chan_chunks = leg->newchunks = leg->vol_rem = 0;
			}
	}
	// ... and the other way round when legs join the bank at slots [s0, s0 + count): the samples in each one's MSVolume's bufferizer (whole
	// chunks in front of them: what the mixer channel held at the detach), the mixer channel's bufferizer of a bank without AGC -- staged
	// per leg (give_remainder) between give_begin and give_end, which sends them to the device in one round trip (mi_fifo_import_range)
	std::vector<int16_t> imp_x;
	std::vector<int32_t> imp_n;
	int imp_s0 = 0, imp_count = 0;
	bool imp_any = false;
	void give_begin(int s0, int count) {
		imp_s0 = s0, imp_count = count, imp_any = false;
		imp_n.assign((size_t)count, 0);
	}
	bool give_remainder(int s, VolumeData *vd, FusedLeg *leg) {
		MSBufferizer *bz = vd->buffer;
		if (light) {
			if (plain) return true;
			bz = &((MixerState *)leg->mixer->data)->channels[leg->pin].bufferizer;
		}
		const int n = (int)(ms_bufferizer_get_avail(bz) / 2);
		if (n <= 0) return true;
		if (n > out_cap || (!light && (n & 7))) return false;
		if (!imp_any) imp_x.assign((size_t)imp_count * (size_t)out_cap, 0);
		imp_any = true;
		ms_bufferizer_read(bz, (uint8_t *)(imp_x.data() + (size_t)(s - imp_s0) * out_cap), (size_t)n * 2);
		imp_n[(size_t)(s - imp_s0)] = n;
		if (light) {
			leg->chan_samples = n;
		} else {
			leg->chan_chunks = plain ? 0 : n / chunk;
			leg->vol_rem = n - leg->chan_chunks * chunk;
		}
		return true;
	}
	bool give_end() { // (every stream of the range is written: the others are empty, as start_slots left them.  The canceller appends whole frames at a tail it takes to be frame-aligned: f_out's queues end on the ring's end)
		if (!imp_any) return true;
		imp_any = false;
		return mi_fifo_import_range(light ? f_chan : f_out, imp_s0, imp_count, imp_x.data(), out_cap, imp_n.data(), light ? 0 : 1) == MI_OK;
	}
	uint8_t *d_dgate_any() {
		if (!d_takegate) d_takegate = devmem<uint8_t>((size_t)nlegs);
		return d_takegate;
	}
	uint8_t *d_takegate = nullptr;
	// A slot's owner leaves while the bank's work for the coming tick is already out (it left at the end of the last graph walk):
	// the reference's filters would have handed that tick's audio on in the walk itself, so it goes out now -- the speaker frames
	// of every leg (LegBank::finish), the owner's own mix or chunks; the others' follow with the hub's flush as usual.
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
	bool want_peers() { // (hub locked) the meter batch and its rows, on first use
		if (vol_peer) return true;
		if (failed) return false;
		const size_t Ln = (size_t)nlegs;
		pcap = (std::max(960, 2 * ns) + 7) & ~7; // (VolumePool::cap_samples: the facade's own rows, so that over-long blocks are cut alike)
		if (mi_volume_create(hub->ctx, nlegs, (int)rate, &vol_peer) != MI_OK || mi_volume_set_peer_batch(vol, vol_peer) != MI_OK) {
			mi_failed("the echo limiter's peer batch");
			return false;
		}
		h_pk = pinned<int16_t>(kMaxRounds * Ln * pcap);
		d_pk = devmem<int16_t>(Ln * pcap);
		h_pn = pinned<int32_t>(kMaxRounds * Ln);
		d_pn = devmem<int32_t>(Ln);
		h_pround = pinned<mi_volume_state>(kMaxRounds * Ln);
		pstate.resize(Ln);
		if (!failed) {
			memset(h_pn, 0, kMaxRounds * Ln * 4);
			MI_MUST(mi_volume_get_state(vol_peer, 0, nlegs, pstate.data()));
		}
		return !failed;
	}
	// the peers' blocks of this walk: metered round by round, ahead of everything the chain's MSVolume does in this enqueue
	bool enqueue_peers() {
		if (!vol_peer || npeers == 0 || failed) return false;
		const size_t Ln = (size_t)nlegs, UL = (size_t)hi * mm;
		int rounds = 0;
		for (size_t s = 0; s < UL; ++s) {
			FusedLeg *leg = legs[s];
			const int st = (leg && leg->peer) ? leg->peer_staged : 0;
			for (int r = st; r < kMaxRounds; ++r) h_pn[(size_t)r * Ln + s] = 0;
			if (st) leg->peer_metered = true, leg->peer_staged = 0;
			rounds = std::max(rounds, st);
		}
		for (int r = 0; r < rounds; ++r) {
			if (zero_copy) {
				MI_MUST(mi_volume_process(vol_peer, h_pk + (size_t)r * Ln * pcap, pcap, pcap, h_pn + (size_t)r * Ln));
			} else {
				MI_MUST(mi_copy_h2d_pinned(hub->ctx, d_pk, h_pk + (size_t)r * Ln * pcap, UL * pcap * 2));
				MI_MUST(mi_copy_h2d_pinned(hub->ctx, d_pn, h_pn + (size_t)r * Ln, Ln * 4));
				MI_MUST(mi_volume_process(vol_peer, d_pk, pcap, pcap, d_pn));
			}
			MI_MUST(mi_volume_get_state_async(vol_peer, 0, (int)UL, h_pround + (size_t)r * Ln));
			++launches;
		}
		prounds = rounds;
		return rounds > 0;
	}
	void finish_peers() { // update_energy's extremum records, msvolume.c:405-406: one per block, in order
		if (!prounds || failed) {
			prounds = 0;
			return;
		}
		const size_t Ln = (size_t)nlegs, UL = (size_t)hi * mm;
		for (size_t s = 0; s < UL; ++s) {
			FusedLeg *leg = legs[s];
			if (!leg || !leg->peer || !leg->peer_metered) continue;
			leg->peer_metered = false;
			VolumeData *pd = (VolumeData *)leg->peer->data;
			for (int r = 0; r < prounds; ++r) {
				if (h_pn[(size_t)r * Ln + s] <= 0) continue;
				pstate[s] = h_pround[(size_t)r * Ln + s];
				if (hub->ticker) {
					pd->max.record_max(hub_time(hub), pstate[s].energy);
					pd->min.record_min(hub_time(hub), pstate[s].energy);
				}
			}
		}
		prounds = 0;
	}
	// a graph is being detached between two ticks (deliver_*_in_scope): rows staged in the last walk whose launches have not left --
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
		// (this may be the APPLICATION's thread -- a postprocess, msticker.c:221 -- while the ticker walks the bank's other graphs: only the
		// owner's graph is handed anything here, the other legs' speaker frames follow with the ticker's own flush; TickerHub::scope)
		const std::unordered_set<MSFilter *> *outer = hub->scope;
		std::unordered_set<MSFilter *> own;
		if (!outer) {
			graph_of(owner_filter, own);
			hub->scope = &own;
		}
		finish();
		emit(owner_filter, slot);
		hub->scope = outer;
	}
	// vstate as the device holds it NOW (a leg is about to leave with its MSVolume's running state): launches that are out and not
	// waited for yet are waited for, their read-back taken
	void settle_meters() {
		if (!outstanding && !early) return;
		if (failed) return;
		sync_stream();
		if (failed || !mixed) return;
		for (size_t s = 0; s < (size_t)nlegs; ++s)
			if (legs[s] && !vs_dirty[s]) vstate[s] = h_vstate[s];
	}
	// the meters behind a levelling round that is not the flush's last (read back with the round's results: finish() records them)
	void meter_round(size_t UL) {
		if (vrounds >= kLegMeterRounds || failed) return;
		MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vround + (size_t)vrounds * nlegs));
		++vrounds;
	}
	bool enqueue_plain(bool any_ref, bool any_refx, bool any_inj, int rounds) {
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs, UL = (size_t)hi;
		int maxc = 0;
		for (size_t s = 0; s < UL; ++s) {
			FusedLeg *leg = legs[s];
			if (!leg || failed || nout[s] > 0) continue; // (nout > 0: an earlier enqueue of this flush already levelled this leg's chunks)
			nout[s] = std::min(leg->newchunks, kLegMaxChunks); // (more than that in one flush: the rest waits in the queue)
			leg->newchunks -= nout[s];
			leg->metered |= nout[s] > 0;
			maxc = std::max(maxc, nout[s]);
			for (int r = 0; r + 1 < nout[s] && vrounds + r < kLegMeterRounds; ++r) vhas[(size_t)(vrounds + r) * L + s] = 1;
		}
		if (failed) return false;
		bool any = enqueue_cancellers(any_ref, any_refx, any_inj, rounds);
		if (maxc && enc_law >= 0) { // the chunks stay on the device, what comes back is their G.711 (alaw_enc_process alaw.c:56-90)
			for (int r = 0; r < maxc; ++r)
				for (size_t s = 0; s < L; ++s) h_elen[(size_t)r * L + s] = (s < UL && r < nout[s]) ? chunk : 0;
			if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_elen, h_elen, (size_t)maxc * L * 4));
			for (int r = 0; r < maxc; ++r) {
				MI_MUST(mi_volume_process_fifo_flags(vol, f_out, d_mix + (size_t)r * L * chunk, chunk, chunk, MI_VOLMIX_DRY_SKIPS));
				MI_MUST(mi_g711_encode(ctx, enc_law, d_mix + (size_t)r * L * chunk, (size_t)chunk, (zero_copy ? h_codes : d_codes) + (size_t)r * L * chunk, (size_t)chunk,
				                       (zero_copy ? h_elen : d_elen) + (size_t)r * L, chunk, UL));
				launches += 2;
				if (r + 1 < maxc) meter_round(UL);
			}
			if (!zero_copy) MI_MUST(mi_copy_d2h_pinned(ctx, h_codes, d_codes, ((size_t)(maxc - 1) * L + UL) * chunk));
			MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vstate));
			mixed = true;
			any = true;
		} else if (maxc) {
			if (!cur) cur = free_slab();
			uint8_t *dst = cur ? cur->payload() : reinterpret_cast<uint8_t *>(h_copy);
			int16_t *rows = zero_copy ? reinterpret_cast<int16_t *>(dst) : d_mix;
			for (int r = 0; r < maxc; ++r) { // (rows of round r start behind what a leg still has ready from an earlier enqueue of this flush)
				MI_MUST(mi_volume_process_fifo_flags(vol, f_out, rows + (size_t)r * L * chunk, chunk, chunk, MI_VOLMIX_DRY_SKIPS));
				++launches;
				if (r + 1 < maxc) meter_round(UL);
			}
			if (!zero_copy) MI_MUST(mi_copy_d2h_pinned(ctx, dst, d_mix, ((size_t)(maxc - 1) * L + UL) * chunk * 2));
			MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vstate));
			mixed = true;
			any = true;
		}
		if (check_levels && any) {
			lv_fresh = true, ++lv_seq;
			MI_MUST(mi_fifo_levels(f_mic, d_lv));
			MI_MUST(mi_fifo_levels(f_ref, d_lv + L));
			MI_MUST(mi_fifo_levels(f_out, d_lv + 2 * L));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_lv, d_lv, 3 * L * 4));
		}
		outstanding |= any;
		return any;
	}

	bool enqueue() override {
		bool any = false;
		const bool was_early = early;
		if (early) { // already out since the end of the last graph walk
			early = false;
			any = early_any;
		}
		// (what was staged after an early enqueue -- a conference that joined the bank later in that walk -- goes out now)
		if (!was_early || staged_since) any |= enqueue_at(hub_time(hub));
		outstanding = false; // the hub waits for the stream right behind this
		return any;
	}
	bool enqueue_at(uint64_t now) {
		mi_ctx *ctx = hub->ctx;
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm; // legs of the conference slots ever handed out
		mark(0);
		if (outstanding) sync_stream(); // (rare: a second enqueue in one flush) the staging arrays are about to be rewritten
		staged_since = false;
		if (root) emitted();
		// ---- pending control changes (methods called since the last flush)
		// (the mixer's controls go up behind the conferences' ticks below: a lone contributor's are overridden, push_controls)
		if (v_dirty) { // (runs of neighbouring legs go up in ONE call each: a ticker's legs fused at an attach are thousands of neighbours, a call waits for the stream)
			bool held = false;
			auto go = [&](size_t s, const std::vector<uint8_t> &dirty) { return dirty[s] == 1 && v_delay[s] <= 0; };
			for (size_t s = 0; s < UL; ++s) {
				if (v_delay[s] > 0 && (vp_dirty[s] == 1 || vs_dirty[s] == 1)) held = true; // chunks from before the call are still to be taken: not yet
				if (go(s, vs_dirty)) { // (vstate is what the device holds: read back with the last launch's results, nothing launched since)
					vstate[s].gain = vpatch[s].gain;
					if (vpatch[s].also_target) vstate[s].target_gain = vpatch[s].target;
				}
			}
			for (size_t s = 0; s < UL;) {
				if (!go(s, vp_dirty)) {
					++s;
					continue;
				}
				size_t e = s;
				while (e < UL && go(e, vp_dirty)) vp_dirty[e++] = 0;
				MI_MUST(mi_volume_set_params(vol, (int)s, (int)(e - s), &vparams[s]));
				s = e;
			}
			for (size_t s = 0; s < UL;) {
				if (!go(s, vs_dirty)) {
					++s;
					continue;
				}
				size_t e = s;
				while (e < UL && go(e, vs_dirty)) vs_dirty[e++] = 0;
				MI_MUST(mi_volume_set_state(vol, (int)s, (int)(e - s), &vstate[s]));
				s = e;
			}
			v_dirty = held; // (entries at 2 wait for flushed(), which raises v_dirty again)
		}
		// ---- the host's half: framing decisions leg by leg
		int rounds = 0, light_rounds = 0;
		bool any_ref = false, any_refx = false, any_inj = false;
		const bool pfon = leg_prefetch_on();
		for (size_t s = 0; s < UL; ++s) {
			FusedLeg *leg = legs[s];
			if (pfon) { // (a pass of this loop is ~50 ns: far enough ahead for a miss) eight legs ahead: the leg; four: its canceller's state (the far end's queue for the speaker frames)
				if (s + 8 < UL) pf2(legs[s + 8], sizeof(FusedLeg));
				if (s + 4 < UL && legs[s + 4]) pf2(legs[s + 4]->ec_data, sizeof(SpeexECState));
			}
			h_cnt[s] = h_cnt[L + s] = h_cnt[2 * L + s] = 0;
			if (light && !plain && !leg)
				for (int r = 0; r < kLegLightRounds; ++r) h_fcnt[(size_t)r * L + s] = 0;
			for (int r = 0; r < kMaxRounds; ++r) h_gate[(size_t)r * L + s] = leg && r < leg->staged_mic;
			if (eq)
				for (int r = 0; r < kMaxRounds; ++r) h_ecnt[(size_t)r * L + s] = (leg && r < leg->staged_mic) ? ns : 0;
			if (!leg) continue;
			rounds = std::max(rounds, leg->staged_mic);
			const int nfr = failed ? 0 : ec_frames(leg) + leg->pre_frames;
			leg->pre_frames = leg->framed_mic = 0;
			h_cnt[s] = std::min(leg->staged_ref, ns);
			h_cnt[L + s] = leg->staged_ref - h_cnt[s];
			h_cnt[2 * L + s] = leg->inject;
			any_ref |= h_cnt[s] > 0;
			any_refx |= h_cnt[L + s] > 0;
			any_inj |= leg->inject > 0;
			leg->staged_mic = leg->staged_ref = leg->inject = 0;
			if (light && plain) {
				leg->newchunks += nfr; // (handed on frame by frame: a "chunk" of this bank is a frame)
			} else if (light) {
				leg->lt_frames = std::min(nfr, kLegLightRounds);
				leg->new_samples += leg->lt_frames * F;
				leg->metered |= nfr > 0;
				light_rounds = std::max(light_rounds, leg->lt_frames);
				for (int r = 0; r < kLegLightRounds; ++r) h_fcnt[(size_t)r * L + s] = r < leg->lt_frames ? F : 0;
				for (int r = 0; r + 1 < leg->lt_frames && vrounds + r < kLegMeterRounds; ++r) vhas[(size_t)(vrounds + r) * L + s] = 1;
			} else {
				leg->vol_rem += nfr * F; // MSVolume's re-framing to 10 ms chunks (msvolume.c:480-486)
				leg->newchunks += leg->vol_rem / ns;
				leg->vol_rem %= ns;
			}
		}
		drops.clear();
		sdrops.clear();
		mark(1);
		const bool pany = enqueue_peers();
		if (plain) {
			const bool a = enqueue_plain(any_ref, any_refx, any_inj, rounds);
			outstanding |= pany;
			return a || pany;
		}
		bool ticked = false;
		for (int c = 0; c < capacity; ++c) { // a mixer ticks once per ticker time, whoever enqueues
			h_run[c] = 0;
			if (failed || c >= hi || !owner[(size_t)c] || conf_time[(size_t)c] == now) continue;
			conf_time[(size_t)c] = now;
			conf_tick(c, now);
			h_run[c] = conf_ready[(size_t)c];
			ticked |= conf_ready[(size_t)c] != 0;
		}
		mixed |= ticked;
		if (failed) return false;
		push_controls();
		mark(2);
		bool any = enqueue_cancellers(any_ref, any_refx, any_inj, rounds);
		mark(3);
		if (light && light_rounds) { // MSVolume without AGC: every frame of the tick metered and levelled as a block of its own, then on to the channel's queue
			if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_fcnt, h_fcnt, (size_t)light_rounds * L * 4));
			for (int r = 0; r < light_rounds; ++r) {
				MI_MUST(mi_volume_process_fifo_flags(vol, f_out, d_lev, F, F, MI_VOLMIX_DRY_SKIPS));
				MI_MUST(mi_fifo_push(f_chan, d_lev, F, F, (zero_copy ? h_fcnt : d_fcnt) + (size_t)r * L));
				launches += 2;
				if (r + 1 < light_rounds) meter_round(UL);
			}
			if (!ticked) MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vstate));
			mixed = true;
			any = true;
		}
		if (ticked) {
			if (!cur) cur = free_slab();
			int16_t *host_rows = reinterpret_cast<int16_t *>(cur ? (void *)cur->payload() : (void *)h_copy);
			if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_run, h_run, (size_t)capacity));
			MI_MUST(mi_mixer_process_volume_fifo_flags(mix, light ? vol_id : vol, 0, light ? f_chan : f_out, zero_copy ? host_rows : d_mix, MI_VOLMIX_DRY_SKIPS,
			                                           zero_copy ? h_run : d_run));
			for (const auto &dk : sdrops) { // samples the channels' flow control discards (rare: a pin that kept two ticks queued for 5 s)
				memset(h_dgate, 0, L);
				h_dgate[(size_t)dk.first] = 1;
				if (!zero_copy) MI_MUST(mi_copy_h2d_pinned(ctx, d_dgate, h_dgate, L));
				for (int left = dk.second; left > 0; left -= std::min(left, ns))
					MI_MUST(mi_fifo_pop(f_chan, std::min(left, ns), d_scratch, ns, nullptr, zero_copy ? h_dgate : d_dgate, 0));
				sync_stream(); // (the gate row is rewritten for the next one)
			}
			mark(4);
			++launches;
			for (const auto &dk : drops) // chunks the channels' flow control discards: metered (MSVolume saw them), never mixed
				for (int k = 0; k < dk.second; ++k) {
					MI_MUST(mi_volume_process_fifo_range(vol, f_out, d_scratch, ns, ns, dk.first, 1));
					++launches;
				}
			mark(5);
			if (!zero_copy) MI_MUST(mi_copy_d2h_pinned(ctx, host_rows, d_mix, UL * ns * 2));
			MI_MUST(mi_volume_get_state_async(vol, 0, (int)UL, h_vstate));
			mark(6);
			any = true;
			if (trace_ms > 0 && (double)(tr[6] - tr[0]) * 1e-6 > trace_ms)
				fprintf(stderr, "mi355x leg bank %p tick %u (%zu slabs): enqueue took %.2f ms: controls + framing %.2f, conference ticks %.2f, far end + cancellers %.2f (%d rounds), run mask + volmix %.2f, slab %.2f, downloads %.2f\n",
				        (void *)this, hub->ticker ? (unsigned)hub->ticker->ticks : 0u, slabs.size(), (double)(tr[6] - tr[0]) * 1e-6, (double)(tr[1] - tr[0]) * 1e-6, (double)(tr[2] - tr[1]) * 1e-6, (double)(tr[3] - tr[2]) * 1e-6, rounds,
				        (double)(tr[4] - tr[3]) * 1e-6, (double)(tr[5] - tr[4]) * 1e-6, (double)(tr[6] - tr[5]) * 1e-6);
			if (trace_ms > 0 && (double)(tr[6] - tr[0]) * 1e-6 > trace_ms)
				for (size_t i = 1; i < trc.size(); ++i) fprintf(stderr, "    %-22s %.3f ms\n", trc[i].first, (double)(trc[i].second - trc[i - 1].second) * 1e-6);
		}
		if (check_levels && any) {
			lv_fresh = true, ++lv_seq;
			MI_MUST(mi_fifo_levels(f_mic, d_lv));
			MI_MUST(mi_fifo_levels(f_ref, d_lv + L));
			MI_MUST(mi_fifo_levels(f_out, d_lv + 2 * L));
			if (f_chan) MI_MUST(mi_fifo_levels(f_chan, d_lv + 3 * L));
			MI_MUST(mi_copy_d2h_pinned(ctx, h_lv, d_lv, 4 * L * 4));
		}
		any |= pany;
		outstanding |= any;
		return any;
	}

	// The kernels refuse what does not fit a queue and count it (mi_fifo_overflows: the canceller runs nothing for a leg whose result queue has
	// no room, aec_tick.hpp); the host's framing takes the room for granted -- it is what the queues are sized by (out_cap, kLegHeldChunks).
	// Once in 128 flushes (a second and a quarter; the stream is idle here, the read is 4 bytes) the counters are read back: one that moved
	// means the two disagree from here on for some leg -- said, and counted as a late event (MSMI355X_CHECK_LEVELS names the leg)
	uint32_t ovf_seq = 0;
	int32_t ovf_seen[3] = {0, 0, 0};
	void check_overflows() {
		if ((++ovf_seq & 127u) || failed) return;
		mi_fifo *const q[3] = {f_mic, f_ref, f_out};
		static const char *const name[3] = {"microphone", "far-end", "result"};
		for (int k = 0; k < 3; ++k) {
			int32_t n = 0;
			if (!q[k] || mi_fifo_overflows(q[k], &n) != MI_OK || n == ovf_seen[k]) continue;
			ms_error("mi355x leg bank %p: the device refused %d block(s) for want of room in the legs' %s queues: the host's framing no longer describes them", (void *)this,
			         (int)(n - ovf_seen[k]), name[k]);
			ovf_seen[k] = n;
			g_late_events.fetch_add(1, std::memory_order_relaxed);
		}
	}
	void finish() override {
		const size_t L = (size_t)nlegs, UL = (size_t)hi * mm;
		// MSSpeexEC's speaker pin: one frame per microphone frame (speexec.c:261-284).  While a detaching graph is being delivered
		// (TickerHub::scope, the application's thread) only ITS legs' frames go: the others' readers may be walking on the ticker thread,
		// their frames follow with the ticker's own flush
		if (hub->scope) {
			std::vector<std::pair<MSQueue *, mblk_t *>> later;
			for (auto &qm : spk) {
				if (qm.first->prev.filter && hub->scope->count(qm.first->prev.filter)) ms_queue_put(qm.first, qm.second);
				else later.push_back(qm);
			}
			spk.swap(later);
		} else {
			for (auto &qm : spk) ms_queue_put(qm.first, qm.second);
			spk.clear();
		}
		if (spk.empty()) spk_flush_done();
		if (failed) {
			std::fill(conf_ready.begin(), conf_ready.end(), 0);
			g_late_events.fetch_add(1, std::memory_order_relaxed);
			return;
		}
		finish_peers();
		check_overflows();
		if (mixed) {
			const bool pfon = leg_prefetch_on();
			for (size_t s = 0; s < UL; ++s) {
				FusedLeg *leg = legs[s];
				if (pfon && s + 8 < UL) pf2(legs[s + 8], sizeof(FusedLeg));
				if (pfon && s + 4 < UL && legs[s + 4]) pf2(legs[s + 4]->vol_data, sizeof(VolumeData));
				if (!leg) continue;
				vstate[s] = h_vstate[s];
				if (leg->metered && hub->ticker) { // update_energy's extremum records, msvolume.c:405-406: one per chunk, in order
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
			if (cur) {
				cur->state.store(1, std::memory_order_release);
				root = esballoc(cur->payload(), cur->bytes, 0, mix_slab_release);
			}
			mixed = false;
			if (plain)
				for (size_t s = 0; s < UL; ++s) nready[s] = nout[s], nout[s] = 0;
		}
		const bool lv_now = lv_fresh; // (a flush that launched nothing read no levels)
		lv_fresh = false;
		if (check_levels && lv_now)
			for (size_t s = 0; s < UL; ++s) {
				FusedLeg *leg = legs[s];
				if (!leg || leg->lv_from > lv_seq) continue; // (taken since the levels were read: an application's thread may re-plumb between a launch and its flush)
				const int want_out = light ? (plain ? leg->newchunks * F : 0) : leg->vol_rem + (leg->chan_chunks + leg->newchunks) * ns;
				if (light && !plain && h_lv[3 * L + s] != leg->chan_samples + leg->new_samples) {
					ms_error("mi355x fused leg %d: the mixer channel's queue holds %d samples, the host's framing says %d", (int)s, h_lv[3 * L + s], leg->chan_samples + leg->new_samples);
					g_late_events.fetch_add(1, std::memory_order_relaxed);
				}
				if (h_lv[s] != leg->echo_level || h_lv[L + s] != leg->dref_level || h_lv[2 * L + s] != want_out) {
					ms_error("mi355x fused leg %d: device queues (%d, %d, %d) differ from the host's framing (%d, %d, %d)", (int)s, h_lv[s], h_lv[L + s],
					         h_lv[2 * L + s], leg->echo_level, leg->dref_level, want_out);
					g_late_events.fetch_add(1, std::memory_order_relaxed);
				}
			}
	}

	void emit(MSFilter *f, int c) override { // mixer_process :336-343 (conference mode): one block per enabled output
		if (plain && enc_law >= 0) { // ... to the leg's encoder, as codes: packed to its ptime there
			FusedLeg *leg = legs[(size_t)c];
			for (int r = 0; leg && leg->enc && r < nready[(size_t)c]; ++r) enc_take_codes(leg->enc, h_codes + ((size_t)r * nlegs + (size_t)c) * chunk, chunk);
			nready[(size_t)c] = 0;
			return;
		}
		if (plain) { // the leg's MSVolume hands its levelled chunks on (volume_process :500-502)
			const uint8_t *base = root ? cur->payload() : reinterpret_cast<const uint8_t *>(h_copy);
			for (int r = 0; r < nready[(size_t)c]; ++r) {
				uint8_t *row = const_cast<uint8_t *>(base) + (((size_t)r * nlegs + (size_t)c) * chunk) * 2;
				mblk_t *om;
				if (root) {
					om = dupb(root);
					om->b_rptr = row;
					om->b_wptr = row + (size_t)chunk * 2;
				} else {
					om = allocb((size_t)chunk * 2, 0);
					memcpy(om->b_wptr, row, (size_t)chunk * 2);
					om->b_wptr += chunk * 2;
				}
				if (f->outputs[0]) ms_queue_put(f->outputs[0], om);
				else freemsg(om);
			}
			nready[(size_t)c] = 0;
			return;
		}
		if (!conf_ready[(size_t)c]) return;
		conf_ready[(size_t)c] = 0;
		MixerState *s = (MixerState *)f->data;
		const uint8_t *base = root ? cur->payload() : reinterpret_cast<const uint8_t *>(h_copy);
		for (int pin = 0; pin < mm && pin < MIXER_MAX_CHANNELS; ++pin) {
			MSQueue *q = f->outputs[pin];
			if (!q || !s->channels[pin].output_enabled || pin == lone[(size_t)c]) continue;
			uint8_t *row = const_cast<uint8_t *>(base) + ((size_t)(c * mm + pin) * ns) * 2;
			mblk_t *om;
			if (root) { // the row as it lies in the slab
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
	void emitted() override { // the flush's own reference: the slab returns to the ring when the last block downstream is freed
		if (root) freeb(root);
		root = nullptr;
		cur = nullptr;
	}
	void flushed() override { // the coming flush is through: what the methods set while its blocks were waiting goes live
		const size_t UL = (size_t)hi * mm;
		if (!eq_later.empty()) {
			if (eq && !failed)
				for (const EqualizerPool::Op &o : eq_later) EqualizerPool::apply(eq, o);
			eq_later.clear();
		}
		for (size_t s = 0; s < UL; ++s) {
			if (vp_dirty[s] != 2 && vs_dirty[s] != 2) continue;
			if (vp_dirty[s] == 2) vp_dirty[s] = 1;
			if (vs_dirty[s] == 2) vs_dirty[s] = 1;
			v_delay[s] = chunks_waiting(s);
			v_dirty = true;
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
	// chunks leg s's MSVolume has produced that the mixer has not taken yet (a conference with AGC: they are levelled when taken)
	int chunks_waiting(size_t s) const {
		const FusedLeg *leg = legs[s];
		return (leg && !plain && !light) ? leg->chan_chunks + leg->newchunks : 0;
	}
};

// one speaker frame per microphone frame, on counts: ec_emit_speaker_frame with the delay line on the device
void leg_speaker_frame(MSFilter *f, SpeexECState *s, FusedLeg *leg, size_t nbytes, bool immediate) {
	const int fs = (int)(nbytes / 2);
	LegBank *b = leg->bank;
	auto hand_on = [&](mblk_t *m) {
		if (!f->outputs[0]) freemsg(m);
		else if (immediate) ms_queue_put(f->outputs[0], m); // (inside MSSpeexEC's process())
		else b->spk.push_back({f->outputs[0], m});          // (handed on with the flush's results, see LegBank::finish)
	};
	auto frame = [&](bool zeroed) { // (from the flush's slab; a bank that has none to give allocates as before)
		mblk_t *m = b->spk_frame(nbytes);
		if (!m) return ec_block(nbytes);
		if (zeroed) memset(m->b_rptr, 0, nbytes);
		return m;
	};
	if (leg->dref_level < s->nominal_ref_samples + fs) {
		leg->inject += fs; // behind everything the far end delivered so far (ms_bufferizer_put(&s->delayed_ref, silence))
		leg->dref_level += fs;
		hand_on(frame(true));
		if (!s->using_zeroes) ms_warning("Not enough ref samples, using zeroes");
		s->using_zeroes = TRUE;
		return;
	}
	if (s->using_zeroes) ms_message("Samples are back.");
	s->using_zeroes = FALSE;
	// The frame as a WINDOW on the far end's own block where it lies within one (about half of them: 256-sample frames out of 480-sample
	// blocks): a second reference to the block's data with its own read and write positions, the queue then skips those bytes -- no copy, no
	// line of a frame buffer touched.  A frame that straddles two blocks is copied together as before.
	mblk_t *head = peekq(&s->ref.base.q);
	if (head && head != &s->ref.base.q._q_stopper && !head->b_cont && (size_t)(head->b_wptr - head->b_rptr) >= nbytes) {
		mblk_t *w = dupb(head);
		w->b_wptr = w->b_rptr + nbytes;
		w->reserved1 = w->reserved2 = 0, w->ttl_or_hl = 0; // (a frame the reference allocates anew carries no timestamp or marker of the far end's block: speexec.c:273-284)
		ms_bufferizer_skip_bytes(&s->ref.base, (int)nbytes);
		hand_on(w);
		return;
	}
	mblk_t *m = frame(false);
	if (ms_bufferizer_read(&s->ref.base, m->b_rptr, nbytes) == 0) {
		ms_error("mi355x echo canceller: the far-end bufferizer ran dry; silence sent to the speaker");
		memset(m->b_rptr, 0, nbytes);
	}
	hand_on(m);
}

// A graph is being detached (facade_detached, filters.cpp): its fused conferences' and legs' tick in flight is waited for and handed
// on -- speaker frames, mixes / chunks -- before any of its facades lets go (the scoped flush then carries those blocks on through
// whatever facades of the graph sit behind: an encoder, a resampler)
void deliver_fused_in_scope(TickerHub &h) {
	for (Pool *p : h.pools) {
		if (p->key.compare(0, 3, "leg") != 0) continue;
		LegBank *b = static_cast<LegBank *>(p);
		// (a launch is the whole bank's: it leaves from here -- possibly the application's thread, in the middle of the ticker's walk of the bank's
		// OTHER graphs, with only part of them staged -- only when the detaching graph itself staged something that has not left; between two ticks
		// its work is out already and there is nothing to launch)
		bool ours = false;
		for (int s = 0; s < b->hi && !ours; ++s) {
			if (!b->owner[(size_t)s] || !h.scope->count(b->owner[(size_t)s])) continue;
			for (int pin = 0; pin < b->mm && !ours; ++pin)
				if (const FusedLeg *leg = b->legs[(size_t)(s * b->mm + pin)]) ours = leg->staged_mic > 0 || leg->staged_ref > 0 || leg->inject > 0 || leg->pre_frames > 0;
		}
		if (ours) b->launch_staged();
		for (int s = 0; s < b->hi; ++s)
			if (b->owner[(size_t)s] && h.scope->count(b->owner[(size_t)s])) b->deliver_in_flight(b->owner[(size_t)s], s);
	}
}

// Every conference of the bank has been walked in this tick (its mixer runs behind all of its legs in the ticker's
// depth-first order, msticker.c:261-282, so everything the tick will stage IS staged): the bank's uploads and launches go
// out NOW, at the end of the graph walk, instead of at the start of the next tick -- the device works through the idle
// part of the interval and the next tick's flush finds the results waiting.  Same results, same one tick of latency; the
// launches just leave the tick's critical path.  (A tick in which some mixer did not run falls back to the flush.)
double leg_trace_ms(LegBank *b) { return b->trace_ms; }
uint64_t leg_trace_now() { return LegBank::trace_now(); }
void leg_conf_walked(LegBank *b, int c) {
	if (b->no_early || b->failed || b->early || !b->hub->ticker) return;
	const uint32_t tick = b->hub->ticker->ticks;
	if (b->walk_epoch != tick) b->walk_epoch = tick, b->walked = 0;
	if (b->walk_tick[(size_t)c] == tick) return;
	b->walk_tick[(size_t)c] = tick;
	if (++b->walked < b->in_use) return;
	b->early_any = b->enqueue_at(hub_time(b->hub) + (uint64_t)b->hub->ticker->interval); // the mixers' clock reads what the flush would
	b->early = true;
}

// A bank without mixers has no filter that is walked behind all of a leg's facades; its legs' cancellers are (MSSpeexEC runs
// when both the resampler and the far end have, msticker.c:230-242): once every leg of the bank has taken its far end in this
// tick the bank's work leaves, as above.  (A tick in which some far end is late falls back to the flush.)
void leg_far_walked(LegBank *b, FusedLeg *leg) {
	if (b->no_early || b->failed || b->early || !b->hub->ticker) return;
	const uint32_t tick = b->hub->ticker->ticks;
	if (b->walk_epoch != tick) b->walk_epoch = tick, b->walked = 0;
	if (leg->far_tick == tick) return;
	leg->far_tick = tick;
	if (++b->walked < b->in_use) return;
	b->early_any = b->enqueue_at(hub_time(b->hub) + (uint64_t)b->hub->ticker->interval);
	b->early = true;
}

// ---- the facades' fused halves -----------------------------------------------------------------------------------------
// MSResample: this tick's input, re-framed to 10 ms blocks, straight into the bank's staging rows
void leg_stage_mic(MSFilter *f, ResampleData *d) {
	FusedLeg *leg = d->leg;
	LegBank *b = leg->bank;
	const size_t nbytes = (size_t)b->in_len * 2;
	const int ahead = leg_prefetch_ahead();
	if (leg_prefetch_on() && leg->slot + ahead < b->nlegs) { // the next leg's head: its filter, its state, the block waiting on its queue
		if (const FusedLeg *nx = b->legs[(size_t)leg->slot + (size_t)ahead]) {
			pf2(nx->rs, sizeof(MSFilter));
			pf2(nx->rs_data, sizeof(ResampleData));
			pf2(nx->ec, sizeof(MSFilter));
			pf2(nx->ec_data, 192);
		}
		if (leg->slot + ahead + 1 < b->nlegs) pf2(b->legs[(size_t)leg->slot + (size_t)ahead + 1], sizeof(FusedLeg));
	}
	// the usual case -- nothing held back, one whole 10 ms block on the queue -- goes from the block to its row in one copy
	while (ms_bufferizer_get_avail(d->bz) == 0 && leg->staged_mic < kMaxRounds) {
		mblk_t *m = peekq(&f->inputs[0]->q);
		if (!m || m->b_cont || (size_t)(m->b_wptr - m->b_rptr) != nbytes) break;
		getq(&f->inputs[0]->q);
		memcpy(b->h_mic + ((size_t)leg->staged_mic * b->nlegs + (size_t)leg->slot) * b->in_len, m->b_rptr, nbytes);
		freemsg(m);
		leg->staged_mic++;
	}
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	while (ms_bufferizer_get_avail(d->bz) >= nbytes && leg->staged_mic < kMaxRounds) { // (more than kMaxRounds blocks: the rest next tick)
		ms_bufferizer_read(d->bz, (uint8_t *)(b->h_mic + ((size_t)leg->staged_mic * b->nlegs + (size_t)leg->slot) * b->in_len), nbytes);
		leg->staged_mic++;
	}
	if (leg->staged_mic) {
		b->staged_since = true;
		request_flush(f);
	}
}

// MSSpeexEC as the HEAD of a leg (no MSResample in front of it: the sound card or decoder already runs at the canceller's rate):
// its microphone pin's blocks, re-framed to 10 ms through the filter's own `echo` bufferizer, into the bank's staging rows
void leg_stage_mic_ec(MSFilter *f, SpeexECState *s) {
	FusedLeg *leg = s->leg;
	LegBank *b = leg->bank;
	ms_bufferizer_put_from_queue(&s->echo, f->inputs[1]);
	const size_t nbytes = (size_t)b->in_len * 2;
	while (ms_bufferizer_get_avail(&s->echo) >= nbytes && leg->staged_mic < kMaxRounds) { // (more than kMaxRounds blocks: the rest next tick)
		ms_bufferizer_read(&s->echo, (uint8_t *)(b->h_mic + ((size_t)leg->staged_mic * b->nlegs + (size_t)leg->slot) * b->in_len), nbytes);
		leg->staged_mic++;
	}
	if (leg->staged_mic) {
		leg->pre_frames += b->ec_frames(leg, true); // speexec.c:256-288 for the blocks just staged: their speaker frames leave now
		b->staged_since = true;
		request_flush(f);
	}
}

// MSSpeexEC, far end (speexec.c:239-250): dropped until the microphone has started, then kept twice -- for the canceller
// (the device's delay line, through the staging rows) and for the speaker pin (the flow-controlled bufferizer, host)
void leg_take_far_end(MSFilter *f, SpeexECState *s) {
	FusedLeg *leg = s->leg;
	LegBank *b = leg->bank;
	if (!f->inputs[0]) return;
	if (leg_prefetch_on() && leg->slot + leg_prefetch_ahead() < b->nlegs)
		if (const FusedLeg *nx = b->legs[(size_t)leg->slot + (size_t)leg_prefetch_ahead()]) {
			pf2(static_cast<const char *>(nx->ec_data) + 192, sizeof(SpeexECState) > 192 ? sizeof(SpeexECState) - 192 : 0);
			pf2(nx->vol, sizeof(MSFilter));
		}
	if (!s->echostarted) {
		if (!ms_queue_empty(f->inputs[0])) ms_warning("Getting reference signal but no echo to synchronize on.");
		ms_queue_flush(f->inputs[0]);
		return;
	}
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) {
		for (mblk_t *c = m; c; c = c->b_cont) {
			const int16_t *src = (const int16_t *)c->b_rptr;
			int n = (int)((c->b_wptr - c->b_rptr) / 2);
			if (leg->staged_ref + n > (1 + kLegRefOver) * b->ns || leg->dref_level + n + kMaxRounds * 2 * b->F > b->ref_cap) {
				ms_error("mi355x echo canceller: more far end in one tick than the leg's delay line takes; %d samples dropped", n);
				g_late_events.fetch_add(1, std::memory_order_relaxed);
				continue;
			}
			leg->dref_level += n;
			while (n > 0) {
				const int at = leg->staged_ref;
				int16_t *dst = at < b->ns ? b->h_ref + (size_t)leg->slot * b->ns + at : b->h_refx + (size_t)leg->slot * kLegRefOver * b->ns + (at - b->ns);
				const int k = std::min(n, at < b->ns ? b->ns - at : n);
				memcpy(dst, src, (size_t)k * 2);
				src += k, n -= k, leg->staged_ref += k;
			}
		}
		flowbuf_put(&s->ref, m);
	}
	if (leg->staged_ref) {
		b->staged_since = true;
		request_flush(f);
	}
	if (b->plain && leg->rs) leg_far_walked(b, leg); // (headed by MSSpeexEC the leg is done when its microphone is staged too: leg_head_done)
}
// MSSpeexEC as a leg's head has taken its far end AND staged its microphone blocks in this tick
void leg_head_done(FusedLeg *leg) {
	if (leg->bank->plain) leg_far_walked(leg->bank, leg);
}

// ---- fusing ------------------------------------------------------------------------------------------------------------
struct LegCand {
	MSFilter *rs, *ec, *vol;
	int pin;
	MSFilter *peer = nullptr; // MSVolume's echo-limiter peer, to be metered beside the leg
	MSFilter *eq = nullptr;   // mic_equalizer between MSResample and MSSpeexEC
};

// a mic_equalizer of ours that can move into a leg's bank: on the leg's ticker and rate, nothing of its own in flight
bool leg_equalizer_ok(MSFilter *eqf, MSTicker *ticker, int rate) {
	EqualizerData *ed = (EqualizerData *)eqf->data;
	if (eqf->ticker != ticker || ed->rate != rate || ed->leg || ms_bufferizer_get_avail(ed->spill) || !eqf->inputs[0] || !ms_queue_empty(eqf->inputs[0])) return false;
	if (EqualizerPool *p = ed->pool) {
		if (p->failed || p->staged[(size_t)ed->slot] || p->ready[(size_t)ed->slot]) return false;
		for (const EqualizerPool::Op &o : p->later)
			if (o.slot == ed->slot) return false;
	}
	return true;
}
// the equalizer moves into slot leg->slot of the bank's batch: its gains replayed as equalizer_attach replays them, its FIR memory
// read out of the slot it gives up (or cleared: a filter that never ran) -- and back when the leg leaves (EqualizerData::hist)
bool leg_take_equalizer(LegBank *b, FusedLeg *leg, MSFilter *eqf);
void leg_drop_equalizer(LegBank *b, FusedLeg *leg);

// MSVolume (volsend) names an echo-limiter peer (audio_stream_enable_echo_limiter, audiostream.c:2236-2240: volrecv): the leg can take
// it along if that peer is one of ours on the same ticker and rate, a meter and nothing else, named by nobody else, with nothing of its
// own in flight but blocks staged in THIS walk (they move to the leg's meter rows).  *peer = NULL: no peer.  false: the leg keeps its facades.
bool leg_peer_ok(MSFilter *vol, VolumeData *vd, MSFilter **peer) {
	std::lock_guard<std::mutex> g(g_peer_mu);
	*peer = nullptr;
	if (!vd->peered_by.empty()) return false; // (somebody's limiter reads THIS filter's meter: it stays where they find it)
	if (!vd->peer) return true;
	MSFilter *pf = vd->peer;
	if (pf->desc != &ms_mi355x_volume_desc || pf->ticker != vol->ticker) return false;
	VolumeData *pd = (VolumeData *)pf->data;
	if (pd->sample_rate != vd->sample_rate || pd->leg || pd->sleg || pd->meter_leg || pd->peered_by.size() != 1 || pd->peer != NULL) return false;
	if (ms_bufferizer_get_avail(pd->buffer) || ms_bufferizer_get_avail(pd->spill)) return false;
	if (pd->pool && (pd->pool->failed || pd->pool->ready[(size_t)pd->slot] || pd->pool->params_dirty[(size_t)pd->slot] == 2 || pd->pool->state_dirty[(size_t)pd->slot] == 2)) return false;
	mi_volume_state st = volume_start_state(pd);
	if (pd->pool && pd->slot >= 0 && !(pd->pool->state_dirty[(size_t)pd->slot] && pd->pool->gain_patch[(size_t)pd->slot].whole)) {
		st = pd->pool->state[(size_t)pd->slot];
		const VolumePool::GainPatch &gp = pd->pool->gain_patch[(size_t)pd->slot];
		if (pd->pool->state_dirty[(size_t)pd->slot] && gp.also_gain) st.gain = gp.gain;
		if (pd->pool->state_dirty[(size_t)pd->slot] && gp.also_target) st.target_gain = gp.target;
	}
	if (pd->p.agc_enabled || pd->p.noise_gate_enabled || pd->p.remove_dc || pd->p.static_gain != 1.f || st.gain != 1.f || st.target_gain != 1.f || st.ng_gain != 1.f) return false;
	*peer = pf;
	return true;
}

// MSVolume's bufferizer survives a detach (msvolume.c has no postprocess): with AGC it may hold samples short of a 10 ms chunk
// ... in front of them, whole chunks that waited in the mixer channel when a fused conference was detached (LegBank::take_remainders)
bool leg_remainder_ok(const VolumeData *vd, int max_chunks) {
	const size_t avail = ms_bufferizer_get_avail(vd->buffer);
	return avail == 0 || (volume_chunks(vd) && avail <= (size_t)(vd->sample_rate / 100) * 2 * (size_t)max_chunks && avail % 16 == 0); // (whole groups of 8 samples: mi_fifo_reset_range_at)
}

bool leg_rates_ok(uint32_t in, uint32_t out) { // what the canceller's launch up-samples itself (mi_aec_process_fifos_resampled)
	return (in == 16000 && out == 48000) || (in == 8000 && out == 48000) || (in == 8000 && out == 16000);
}

bool is_ec_desc(const MSFilterDesc *d) { return d == &ms_mi355x_speex_ec_desc || d == &ms_mi355x_webrtc_aec_name_desc; }

// Is pin `pin` of mixer `mx` the end of  MSResample -> MSSpeexEC (pin 1) -> MSVolume, all ours, all on the mixer's ticker,
// all fresh?  Fills the candidate.
// MSAudioConference plumbs every endpoint through a resampler pair: mixer_in -> in_resampler -> mixer pin -> out_resampler ->
// mixer_out (src/voip/audioconference.c:209-257).  For an endpoint that already runs at the conference's rate they forward their
// blocks untouched (msresample.c:126-135): such an MSResample of ours between MSVolume and the pin is looked through -- while the
// leg is fused nothing passes it, after un-fusing MSVolume's chunks pass it again as before.
bool is_pass_resampler(MSFilter *f, MSTicker *ticker) {
	if (!f || f->desc != &ms_mi355x_resample_desc || f->ticker != ticker) return false;
	const ResampleData *rd = (const ResampleData *)f->data;
	return rd->input_rate == rd->output_rate && rd->in_nchannels == rd->out_nchannels && !rd->leg && !rd->pool && ms_bufferizer_get_avail(rd->bz) == 0;
}
// the filter MSVolume's output ends up in, behind an in_resampler of ours if there is one (forwarding, or -- an endpoint at another
// rate than its conference, server_leg.inl -- working)
MSFilter *leg_volume_sink(MSFilter *vol) {
	MSQueue *q = vol->outputs[0];
	MSFilter *g = q ? q->next.filter : NULL;
	if (g && g->desc == &ms_mi355x_resample_desc && g->ticker == vol->ticker && ms_queue_empty(q)) {
		q = g->outputs[0];
		g = q ? q->next.filter : NULL;
	}
	return g;
}

// ... and when such a forwarder is told to resample after all, the conference behind it goes back to its facades (the mixer's next
// process() honours it)
void leg_forwarder_changed(MSFilter *rs) { // (an endpoint's in_resampler in front of a fused conference's pin, or its out_resampler behind one)
	for (MSQueue *q : {rs->outputs[0], rs->inputs[0]}) {
		MSFilter *mx = q ? (q == rs->outputs[0] ? q->next.filter : q->prev.filter) : NULL;
		if (!mx || mx->desc != &ms_mi355x_audio_mixer_desc) continue;
		MixerState *ms = (MixerState *)mx->data;
		if (ms->fbank || ms->sbank) ms->unfuse_wanted = true;
	}
}

bool leg_far_end_in_walk(MSFilter *ec, MSFilter *peer);
bool leg_candidate(MSFilter *mx, MixerState *ms, int pin, LegCand &c) {
	MSQueue *q = mx->inputs[pin];
	MSFilter *vol = q->prev.filter;
	if (is_pass_resampler(vol, mx->ticker)) { // the endpoint's in_resampler, forwarding
		if (!ms_queue_empty(q)) return false;
		q = vol->inputs[0];
		vol = q ? q->prev.filter : NULL;
	}
	if (!vol || vol->desc != &ms_mi355x_volume_desc || vol->ticker != mx->ticker) return false;
	VolumeData *vd = (VolumeData *)vol->data;
	// (an echo-limiter peer keeps a CONFERENCE member on its facades: this bank levels a chunk when the mixer takes it, and a chunk that
	// waited a tick in the channel would meet the peer's NEXT meter reading -- a leg without a mixer levels every chunk as it completes)
	if (!leg_peer_ok(vol, vd, &c.peer) || c.peer || vd->sample_rate != ms->rate || vd->leg) return false; // (with or without AGC: the bank follows, LegBank::light)
	// (MSVolume's bufferizer may hold samples short of a 10 ms chunk from before a detach: they move to the device, leg_give_remainder)
	if (ms_bufferizer_get_avail(vd->spill) || !ms_queue_empty(q)) return false;
	// the mixer channel's own bufferizer (the facades ran one by one before this attach, or a batch without AGC was left): levelled
	// samples -- a batch without AGC takes them into its channel queue, one with AGC queues in front of MSVolume and cannot
	const size_t held = ms_bufferizer_get_avail(&ms->channels[pin].bufferizer);
	if (held && (volume_chunks(vd) || held % 2)) return false; // (how much: below, once the frame size is known)
	MSQueue *qe = vol->inputs[0];
	MSFilter *ec = qe ? qe->prev.filter : NULL;
	if (!ec || !is_ec_desc(ec->desc) || qe->prev.pin != 1 || ec->ticker != mx->ticker || !ms_queue_empty(qe)) return false;
	SpeexECState *es = (SpeexECState *)ec->data;
	if (es->bypass_mode || es->unsupported || !es->configured || es->samplerate != ms->rate || es->echostarted || es->leg) return false;
	if (ms_bufferizer_get_avail(&es->echo) || (int)ms_bufferizer_get_avail(&es->delayed_ref) != es->nominal_ref_samples * 2) return false;
	// (what a batch handed back at the last detach -- the chunks that waited in the channel, the ticks in flight, the samples short of a chunk:
	// take_remainders -- moves into the queue it is rebuilt in as long as the queue keeps the room a running batch needs, kLegHeldChunks.
	// The bound was "less than three chunks" until PLUGIN_BENCH_CHURN found conferences that came back with three, five, seven -- and then
	// stayed on their facades for good)
	const int held_max = (kLegHeldChunks - 4) * (vd->sample_rate / 100);
	if (!leg_remainder_ok(vd, kLegHeldChunks - 4) || held / 2 > (size_t)held_max) return false;
	MSQueue *qr = ec->inputs[1];
	MSFilter *rs = qr ? qr->prev.filter : NULL;
	if (rs && equalizer_passes(rs, mx->ticker) && ms_queue_empty(qr) && rs->inputs[0]) { // a mic_equalizer that is not active (audiostream.c:1801): transparent
		qr = rs->inputs[0];
		rs = qr->prev.filter;
	}
	if (!rs || (is_ours(rs->desc) && rs->ticker != mx->ticker)) return false; // (somebody else's filter may still be waiting for its preprocess: graph_preprocessed looks at this plugin's facades only)
	if (!leg_far_end_in_walk(ec, c.peer)) return false;
	c.eq = nullptr;
	if (rs->desc == &ms_mi355x_equalizer_desc) { // mic_equalizer (audiostream.c:1801): the leg's head is the MSResample in front of it
		if (!ms_queue_empty(qr) || !leg_equalizer_ok(rs, mx->ticker, ms->rate)) return false;
		c.eq = rs;
		qr = rs->inputs[0];
		rs = qr->prev.filter;
		if (!rs || rs->desc != &ms_mi355x_resample_desc || rs->ticker != mx->ticker) return false; // (an equalizer of ours as the head: its blocks come with the flush)
	}
	if (rs->desc != &ms_mi355x_resample_desc) { // anything else feeds the canceller at its own rate: MSSpeexEC is the leg's head
		c.rs = nullptr, c.ec = ec, c.vol = vol, c.pin = pin;
		return true;
	}
	if (!ms_queue_empty(qr)) return false;
	ResampleData *rd = (ResampleData *)rs->data;
	if (rd->in_nchannels != 1 || rd->out_nchannels != 1 || (int)rd->output_rate != ms->rate || !leg_rates_ok(rd->input_rate, rd->output_rate)) return false;
	if (rd->leg || ms_bufferizer_get_avail(rd->bz) || (rd->pool && rd->pool->staged[(size_t)rd->slot])) return false;
	c.rs = rs, c.ec = ec, c.vol = vol, c.pin = pin;
	return true;
}

// A fused leg's canceller is launched at the END of the walk that brought its microphone block: the far end of that walk must be
// there by then -- it must reach MSSpeexEC's pin 0 IN the walk.  What the reference's graph guarantees (synchronous filters) holds here
// when pin 0 is fed by a filter that is not ours (a source, dtmfgen, a tee: it runs in the walk), directly or through MSVolumes that
// are meters only (they hand their blocks on in the walk, volume_passes) or the leg's own metered peer.  A facade of ours that delivers
// with the flush directly upstream -- volrecv with a gain, spk_equalizer, a PLC or decoder without a CPU filter behind it -- keeps the
// leg on its facades, whose queues pair the two streams by count whenever they arrive.
bool volume_meter_config(const VolumeData *d);
bool volume_passes(const VolumeData *d);
bool leg_far_end_in_walk(MSFilter *ec, MSFilter *peer) {
	MSQueue *q = ec->inputs[0];
	for (int hops = 0; q && hops < 12; ++hops) {
		MSFilter *g = q->prev.filter;
		if (!g) return true;
		if (!is_ours(g->desc)) return true; // a source, dtmfgen, a tee ..: it runs in the walk and delivers in it, whatever feeds it (a facade of ours above it hands its blocks over at the start of the tick)
		if (equalizer_passes(g, ec->ticker)) { // spk_equalizer, not active (audiostream.c:1828): it hands on in the walk what it is handed in it
			q = g->inputs[0];
			continue;
		}
		if (g->desc != &ms_mi355x_volume_desc) return false;
		VolumeData *vd = (VolumeData *)g->data;
		if (g != peer && !vd->meter_leg && !(volume_meter_config(vd) && vd->feeds_far_end && (!vd->pool || volume_passes(vd)))) return false; // (its running gain may still be on its way back to 1: then it does not pass yet)
		q = g->inputs[0]; // (a meter only: it hands on in the walk what it is handed in it)
	}
	return true;
}
// ... and the other way round: the fused leg whose far end passes through this MSVolume (hub locked; NULL: none)
FusedLeg *leg_fed_far_end_by(MSFilter *vol) {
	MSQueue *q = vol->outputs[0];
	for (int hops = 0; q && hops < 12; ++hops) {
		MSFilter *g = q->next.filter;
		if (!g) return nullptr;
		if (is_ec_desc(g->desc)) return q->next.pin == 0 ? ((SpeexECState *)g->data)->leg : nullptr;
		if (g->desc == &ms_mi355x_equalizer_desc && !((EqualizerData *)g->data)->leg) { // (a spk_equalizer a leg may have been recognised through)
			q = g->outputs[0];
			continue;
		}
		if (is_ours(g->desc) || g->desc->noutputs != 1) return nullptr;
		q = g->outputs[0];
	}
	return nullptr;
}
// ... and the fused leg whose MICROPHONE passes through this (inactive) mic_equalizer: the canceller right behind it
FusedLeg *leg_fed_mic_by(MSFilter *eqf) {
	MSQueue *q = eqf->outputs[0];
	MSFilter *g = q ? q->next.filter : NULL;
	return (g && is_ec_desc(g->desc) && q->next.pin == 1) ? ((SpeexECState *)g->data)->leg : nullptr;
}

// The leg takes its MSVolume's echo-limiter peer along (hub locked): the peer gives up its bank slot, its running state starts the
// meter's slot; blocks it staged earlier in THIS walk move to the leg's meter rows and are handed on now -- from here on it hands
// its blocks on in the walk (leg_stage_peer), so that the far end still meets the microphone block of the same walk in the canceller
bool leg_take_peer(LegBank *b, FusedLeg *leg, MSFilter *pf) {
	if (!b->want_peers()) return false;
	VolumeData *pd = (VolumeData *)pf->data;
	const size_t s = (size_t)leg->slot, Ln = (size_t)b->nlegs;
	volume_keep_state(pd);
	const mi_volume_state st = volume_start_state(pd);
	mi_volume_params pp = pd->p;
	pp.peer = -1;
	if (mi_volume_set_params(b->vol_peer, (int)s, 1, &pp) != MI_OK || mi_volume_set_state(b->vol_peer, (int)s, 1, &st) != MI_OK || mi_volume_reset_max(b->vol_peer, (int)s, 1) != MI_OK)
		return false;
	b->pstate[s] = st;
	leg->peer = pf;
	leg->peer_staged = 0;
	leg->peer_metered = false;
	b->npeers++;
	pd->meter_leg = leg;
	if (VolumePool *p = pd->pool) {
		const size_t c = (size_t)p->capacity, ps = (size_t)pd->slot;
		for (int r = 0; r < p->staged[ps]; ++r) {
			const int n = p->h_n[(size_t)r * c + ps];
			const int16_t *row = p->h_buf + ((size_t)r * c + ps) * p->cap_samples;
			if (n <= 0) continue;
			if (leg->peer_staged < kMaxRounds && n <= b->pcap) {
				memcpy(b->h_pk + ((size_t)leg->peer_staged * Ln + s) * b->pcap, row, (size_t)n * 2);
				b->h_pn[(size_t)leg->peer_staged * Ln + s] = n;
				leg->peer_staged++;
			}
			mblk_t *m = allocb((size_t)n * 2, 0);
			memcpy(m->b_wptr, row, (size_t)n * 2);
			m->b_wptr += n * 2;
			if (pf->outputs[0]) ms_queue_put(pf->outputs[0], m);
			else freemsg(m);
		}
		p->staged[ps] = 0;
		p->release(pd->slot);
		pd->pool = nullptr, pd->slot = -1;
	}
	if (leg->peer_staged) b->staged_since = true;
	return true;
}
// ... and lets go of it (the leg leaves its bank): MSVolume's running state goes with the filter, which finds a bank slot of its own
// at its next block or attach
void leg_drop_peer(LegBank *b, FusedLeg *leg) {
	if (!leg->peer) return;
	VolumeData *pd = (VolumeData *)leg->peer->data;
	if (!b->failed && b->vol_peer) {
		pd->kept = b->pstate[(size_t)leg->slot];
		pd->kept.gain = pd->gain, pd->kept.target_gain = pd->target_gain; // (a gain method while it was metered took the leg out: the gains follow)
		pd->has_kept = true;
	}
	pd->meter_leg = nullptr;
	leg->peer = nullptr;
	leg->peer_staged = 0;
	b->npeers--;
}
// the peer facade's process(): every block on as it came, a copy in the leg's meter rows (cut like the facade's own rows)
void leg_stage_peer(MSFilter *f, VolumeData *d) {
	FusedLeg *leg = d->meter_leg;
	LegBank *b = leg->bank;
	const size_t Ln = (size_t)b->nlegs, s = (size_t)leg->slot;
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) {
		const int n = (int)(msgdsize(m) / 2);
		if (!b->failed && n > 0) {
			std::vector<int16_t> flat;
			const int16_t *src = (const int16_t *)m->b_rptr;
			if (m->b_cont || n > b->pcap) {
				flat.resize((size_t)n);
				copy_payload(m, (uint8_t *)flat.data());
				src = flat.data();
			}
			for (int at = 0; at < n; at += b->pcap) {
				const int k = std::min(b->pcap, n - at);
				if (leg->peer_staged >= kMaxRounds) { // more blocks than launch rounds in one tick: these go unmetered (counted)
					g_late_events.fetch_add(1, std::memory_order_relaxed);
					break;
				}
				memcpy(b->h_pk + ((size_t)leg->peer_staged * Ln + s) * b->pcap, src + at, (size_t)k * 2);
				b->h_pn[(size_t)leg->peer_staged * Ln + s] = k;
				leg->peer_staged++;
			}
		}
		if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
		else freemsg(m);
	}
	if (leg->peer_staged) {
		b->staged_since = true;
		request_flush(f);
	}
}
bool leg_take_equalizer(LegBank *b, FusedLeg *leg, MSFilter *eqf) {
	EqualizerData *ed = (EqualizerData *)eqf->data;
	if (!b->eq) return false;
	const int n = mi_equalizer_fir_len(b->eq), s = leg->slot;
	if (ed->pool) { // (on this hub, which is held)
		ed->hist->assign((size_t)n, 0);
		ed->has_hist = mi_equalizer_get_history(ed->pool->e, ed->slot, ed->hist->data(), n) == MI_OK;
		ed->pool->release(ed->slot);
		ed->pool = nullptr, ed->slot = -1;
	}
	bool ok = mi_equalizer_flatten(b->eq, s) == MI_OK && mi_equalizer_set_active(b->eq, s, ed->active) == MI_OK;
	for (const MSEqualizerGain &g : *ed->pending) ok = ok && mi_equalizer_set_gain(b->eq, s, g.frequency, g.gain, g.width) == MI_OK;
	if (b->eq_used.size() != (size_t)b->nlegs) b->eq_used.assign((size_t)b->nlegs, 0);
	if (ed->has_hist || b->eq_used[(size_t)s]) ok = ok && mi_equalizer_set_history(b->eq, s, ed->has_hist ? ed->hist->data() : nullptr, n) == MI_OK;
	b->eq_used[(size_t)s] = 1;
	ed->has_hist = false;
	ed->leg = leg;
	leg->eq = eqf;
	return ok;
}
void leg_drop_equalizer(LegBank *b, FusedLeg *leg) {
	if (!leg->eq) return;
	EqualizerData *ed = (EqualizerData *)leg->eq->data;
	if (b->eq && !b->failed) {
		const int n = mi_equalizer_fir_len(b->eq);
		ed->hist->assign((size_t)n, 0);
		ed->has_hist = mi_equalizer_get_history(b->eq, leg->slot, ed->hist->data(), n) == MI_OK;
	}
	b->eq_later.erase(std::remove_if(b->eq_later.begin(), b->eq_later.end(), [&](const EqualizerPool::Op &o) { return o.slot == leg->slot; }), b->eq_later.end()); // (they are in the filter's `pending` list: its next slot replays them)
	ed->leg = nullptr;
	leg->eq = nullptr;
}
void leg_eq_op(FusedLeg *leg, const EqualizerPool::Op &op0) { // MS_EQUALIZER_SET_GAIN / SET_ACTIVE on a fused leg's equalizer (hub locked)
	LegBank *b = leg->bank;
	if (!b->eq || b->failed) return;
	EqualizerPool::Op op = op0;
	op.slot = leg->slot;
	if (b->work_waiting()) b->eq_later.push_back(op);
	else EqualizerPool::apply(b->eq, op);
}
mi_equalizer *leg_eq(FusedLeg *leg, int *slot) {
	*slot = leg->slot;
	return leg->bank->eq;
}
mi_volume_state *leg_pstate(FusedLeg *leg) { return leg->bank->vol_peer ? &leg->bank->pstate[(size_t)leg->slot] : nullptr; }
bool leg_frames_chunks(FusedLeg *leg) { return leg && !leg->bank->light; }

// Called (hub locked, ticker thread) by the first facade of a conference's graph to run after an attach.  true = fused:
// as a conference of sending legs (below), else as one of a server's remote members (server_leg.inl)
bool conf_try_fuse_sending(MSFilter *mx);
bool server_try_fuse(MSFilter *mx);
bool conf_try_fuse(MSFilter *mx) {
	MixerState *ms = (MixerState *)mx->data;
	if (ms->fuse_state != 0) return ms->fuse_state == 1;
	ms->fuse_state = 2; // refused, unless one of the two shapes holds
	if (conf_try_fuse_sending(mx) || server_try_fuse(mx)) ms->fuse_state = 1;
	return ms->fuse_state == 1;
}
bool conf_try_fuse_sending(MSFilter *mx) {
	MixerState *ms = (MixerState *)mx->data;
	const bool off = getenv("MSMI355X_NO_FUSE") != nullptr; // (read per attach: an A/B switch, and what the tests compare against)
	if (off || !ms->prepared || ms->conf_mode == 0 || ms->nchannels != 1 || !mx->ticker || mx->ticker->interval != 10 || ms->rate % 100) return false;
	std::vector<LegCand> cand;
	int maxpin = -1;
	for (int pin = 0; pin < mx->desc->ninputs; ++pin) {
		if (!mx->inputs[pin]) continue;
		LegCand c;
		if (!leg_candidate(mx, ms, pin, c)) return false; // a pin fed by anything else: the conference stays on the facades' own banks
		cand.push_back(c);
		maxpin = pin;
	}
	if (cand.empty()) return false;
	// an output-only pin above the last input (a listener's or recorder's tap, served at audiomixer.c:336-343) gets its mix too:
	// the conference's width follows the highest linked pin of either kind
	for (int pin = 0; pin < mx->desc->noutputs; ++pin)
		if (mx->outputs[pin]) maxpin = std::max(maxpin, pin);
	const SpeexECState *e0 = (const SpeexECState *)cand[0].ec->data;
	const uint32_t ir0 = cand[0].rs ? ((const ResampleData *)cand[0].rs->data)->input_rate : (uint32_t)ms->rate; // (no MSResample: the leg comes in at the mixer's rate)
	for (const LegCand &c : cand) { // one shape per conference (a bank is one shape)
		const SpeexECState *e = (const SpeexECState *)c.ec->data;
		const uint32_t ir_c = c.rs ? ((const ResampleData *)c.rs->data)->input_rate : (uint32_t)ms->rate;
		if (e->framesize != e0->framesize || e->filterlength != e0->filterlength || e->nominal_ref_samples != e0->nominal_ref_samples ||
		    ir_c != ir0 || (c.rs == nullptr) != (cand[0].rs == nullptr) || (c.eq == nullptr) != (cand[0].eq == nullptr) ||
		    volume_chunks((VolumeData *)c.vol->data) != volume_chunks((VolumeData *)cand[0].vol->data))
			return false;
	}
	const bool no_agc = !volume_chunks((VolumeData *)cand[0].vol->data); // (10 ms chunks with AGC or an echo-limiter peer, msvolume.c:480)
	int mm = MIXER_MAX_CHANNELS;
	for (int m : {4, 8, 16, 32})
		if (maxpin < m) {
			mm = m;
			break;
		}
	const uint32_t ir = ir0, rate = (uint32_t)ms->rate;
	const int F = e0->framesize, flen = e0->filterlength, delay = e0->nominal_ref_samples;
	const bool with_eq = cand[0].eq != nullptr;
	LegBank *b = bank<LegBank>("leg:" + std::to_string(ir) + ":" + std::to_string(rate) + ":" + std::to_string(F) + ":" + std::to_string(flen) + ":" +
	                               std::to_string(delay) + ":" + std::to_string(mm) + (no_agc ? ":light" : "") + (with_eq ? ":eq" : ""),
	                           1, [&](int cap) { return new LegBank(std::max(1, cap * 4 / mm), ir, rate, F, flen, delay, mm, false, no_agc, with_eq); }); // 64, 256, 1024, .. legs
	const int c = b ? b->acquire(mx) : -1;
	if (c < 0) return false;
	note_slot(mx);
	// ---- the legs: fresh per-leg state at their slots (what the filters' own banks hold at this point), then the facades let
	// go of their own slots
	const int s0 = c * mm;
	bool was_used = false;
	for (int pin = 0; pin < mm; ++pin) was_used |= b->used[(size_t)(s0 + pin)] != 0;
	bool ok = b->start_slots(s0, mm);
	{ // (a pin without a leg keeps the batch's defaults)
		mi_volume_state st0 = b->vstate0;
		for (int pin = 0; pin < mm && was_used; ++pin) b->start_volume((size_t)(s0 + pin), b->vparams0, st0, true);
	}
	for (const LegCand &cd : cand) {
		const size_t s = (size_t)(s0 + cd.pin);
		VolumeData *vd = (VolumeData *)cd.vol->data;
		SpeexECState *es = (SpeexECState *)cd.ec->data;
		volume_keep_state(vd); // (its bank is on this hub, which is held)
		mi_volume_params vp = vd->p;
		vp.peer = cd.peer ? MI_VOLUME_PEER_EXTERNAL : -1;
		b->start_volume(s, vp, volume_start_state(vd), was_used); // as volume_attach_slot starts a slot: MSVolume's running state, if it has one already
		if (es->state_str) { // a saved canceller state goes to the leg's slot (speexec.c:209-211)
			std::vector<uint8_t> blob;
			if (b64_decode(es->state_str, blob) && mi_aec_import_state(b->aec, (int)s, blob.data(), blob.size()) == MI_OK) ms_message("mi355x echo state restored.");
			else ms_error("Could not apply mi355x echo blob: %s", mi_last_error());
		}
	}
	if (!ok) {
		mi_failed("fusing a conference's legs");
		b->release(c);
		return false;
	}
	for (int pin = 0; pin < mm; ++pin) b->flags[(size_t)(s0 + pin)] = 0, b->gains[(size_t)(s0 + pin)] = 1.0f;
	b->give_begin(s0, mm);
	// MSResample's position and history follow the filters from slot to slot (msresample.c:117-120: the handle outlives a detach): the
	// conference's members in one round trip (a slot without a state to restore keeps a fresh stream's: zeros, as start_slots left it)
	const size_t rs_each = b->rs ? (size_t)mi_resampler_state_bytes(b->rs) : 0;
	std::vector<uint8_t> rs_buf(rs_each * (size_t)mm, 0);
	bool rs_any = false;
	for (const LegCand &cd : cand) {
		FusedLeg *leg = new FusedLeg();
		leg->lv_from = b->lv_seq + 1;
		leg->bank = b, leg->slot = s0 + cd.pin, leg->pin = cd.pin;
		leg->rs = cd.rs, leg->ec = cd.ec, leg->vol = cd.vol, leg->mixer = mx;
		leg->rs_data = cd.rs ? cd.rs->data : nullptr, leg->ec_data = cd.ec->data, leg->vol_data = cd.vol->data;
		leg->dref_level = leg->inject = delay; // zeroes for the time of the delay (speexec.c:205-208): pushed by the first enqueue
		b->legs[(size_t)leg->slot] = leg;
		SpeexECState *es = (SpeexECState *)cd.ec->data;
		VolumeData *vd = (VolumeData *)cd.vol->data;
		if (cd.rs) {
			ResampleData *rd = (ResampleData *)cd.rs->data;
			if (rd->pool) {
				if (rd->slots->size() == 1) resample_keep_from(rd, rd->pool->r, rd->slot, rd->input_rate, rd->output_rate);
				resample_release(rd);
			}
			if (rs_each) rs_any |= resample_restore_bytes(rd, rs_buf.data() + (size_t)cd.pin * rs_each, rs_each, b->in_rate, b->rate, true);
			rd->leg = leg;
		}
		ms_bufferizer_flush(&es->delayed_ref); // the delay line lives on the device now
		if (es->pool) { // (a filter that had run on its facade before: normally preprocess opens no slot, ec_acquire)
			es->pool->staged[(size_t)es->slot] = es->pool->ready[(size_t)es->slot] = 0;
			es->pool->release(es->slot); // (the last release of a bank destroys it)
			es->pool = nullptr, es->slot = -1;
		}
		es->leg = leg;
		if (vd->pool) {
			vd->pool->release(vd->slot);
			vd->pool = nullptr, vd->slot = -1;
		}
		vd->leg = leg;
		if (!b->give_remainder(leg->slot, vd, leg)) mi_failed("moving MSVolume's queued samples to the device");
		if (cd.peer && !leg_take_peer(b, leg, cd.peer)) mi_failed("taking the echo limiter's peer into the batch");
		if (cd.eq && !leg_take_equalizer(b, leg, cd.eq)) mi_failed("taking the leg's equalizer into the batch");
	}
	if (!b->give_end()) mi_failed("moving MSVolume's queued samples to the device");
	if (with_eq && b->eq && mi_equalizer_prepare(b->eq) != MI_OK) mi_failed("designing the legs' equalizers"); // (every leg's taps at once, on the attaching thread)
	if (rs_any && mi_resampler_set_states(b->rs, s0, mm, rs_buf.data(), rs_buf.size()) != MI_OK) mi_failed("moving the resamplers' states to the device");
	if (ms->pool) { // (a conference that had mixed on its facade before: normally preprocess opens no slot, mixer_acquire)
		ms->pool->staged[(size_t)ms->slot] = ms->pool->ready[(size_t)ms->slot] = 0;
		ms->pool->release(ms->slot);
		ms->pool = nullptr, ms->slot = -1;
	}
	ms->fbank = b, ms->fconf = c;
	b->conf_time[(size_t)c] = (uint64_t)-1;
	b->staged_since = true;
	ms->unfuse_wanted = false;
	mixer_push_controls(mx, ms);
	ms_message("mi355x: conference %p fused: %d legs %u -> %u Hz, frame %d, tail %d, one device-resident batch (bank of %d x %d)", (void *)mx,
	           (int)cand.size(), ir, rate, F, flen, b->capacity, mm);
	return true;
}

// the head of a leg (MSResample) looks for the mixer at the end of its chain
// (through a mic_equalizer of ours between the two, audiostream.c:1801)
MSQueue *leg_past_equalizer(MSQueue *q) {
	MSFilter *g = q ? q->next.filter : NULL;
	return (g && g->desc == &ms_mi355x_equalizer_desc && q->next.pin == 0) ? g->outputs[0] : q;
}
MSFilter *leg_find_mixer(MSFilter *rs) {
	MSQueue *q = leg_past_equalizer(rs->outputs[0]);
	MSFilter *ec = q ? q->next.filter : NULL;
	if (!ec || !is_ec_desc(ec->desc) || q->next.pin != 1) return NULL;
	q = ec->outputs[1];
	MSFilter *vol = q ? q->next.filter : NULL;
	if (!vol || vol->desc != &ms_mi355x_volume_desc) return NULL;
	MSFilter *mx = leg_volume_sink(vol);
	// (a mixer that is not a conference -- an AudioStream's outbound_mixer, audiostream.c:1585-1588,1807 -- is "anything else": the leg fuses without a mixer)
	return (mx && mx->desc == &ms_mi355x_audio_mixer_desc && ((MixerState *)mx->data)->conf_mode != 0) ? mx : NULL;
}

// ... and MSSpeexEC, when it is the head itself
MSFilter *leg_find_mixer_ec(MSFilter *ec) {
	MSQueue *q = ec->outputs[1];
	MSFilter *vol = q ? q->next.filter : NULL;
	if (!vol || vol->desc != &ms_mi355x_volume_desc) return NULL;
	MSFilter *mx = leg_volume_sink(vol);
	// (a mixer that is not a conference -- an AudioStream's outbound_mixer, audiostream.c:1585-1588,1807 -- is "anything else": the leg fuses without a mixer)
	return (mx && mx->desc == &ms_mi355x_audio_mixer_desc && ((MixerState *)mx->data)->conf_mode != 0) ? mx : NULL;
}

void ec_prepare(MSFilter *f);    // echo_canceller.inl: the body of ec_preprocess
void ec_acquire(MSFilter *f);    // ... and a bank slot of its own
void mixer_prepare(MSFilter *f, bool running); // mixer.inl

// MSVolume's running state goes with the filter, not with the bank slot (volume.inl: VolumeData::kept)
void leg_keep_volume(FusedLeg *leg) {
	VolumeData *vd = (VolumeData *)leg->vol->data;
	if (leg->bank->failed) return;
	LegBank *b = leg->bank;
	const size_t s = (size_t)leg->slot;
	vd->kept = b->vstate[s];
	if (b->vs_dirty[s]) { // a gain method since the last launch: not on the device yet
		vd->kept.gain = b->vpatch[s].gain;
		if (b->vpatch[s].also_target) vd->kept.target_gain = b->vpatch[s].target;
	}
	vd->has_kept = true;
}

// A leg leaves its batch WHILE ATTACHED (a member stopped qualifying: a method, nothing the reference's canceller would notice): its
// canceller goes with the filter -- the adapted state (mi_aec_export_state / import_state: a restored stream continues bit for bit) into
// the bank slot ec_prepare has just given the facade, the microphone samples short of a frame and the far end's delay line back into
// the facade's own bufferizers (speexec.c:60-62: `echo`, `delayed_ref`).  Before round 5's end the canceller simply started over.
void leg_return_canceller(LegBank *b, FusedLeg *leg, bool started) {
	SpeexECState *es = (SpeexECState *)leg->ec->data;
	if (b->failed || !es->pool || es->slot < 0 || es->pool->failed) return;
	std::vector<uint8_t> blob(mi_aec_blob_bytes(b->aec));
	if (mi_aec_export_state(b->aec, leg->slot, blob.data(), blob.size()) != MI_OK || mi_aec_import_state(es->pool->a, es->slot, blob.data(), blob.size()) != MI_OK) {
		ms_warning("mi355x: a leg's canceller could not follow it out of its batch (%s): it starts over", mi_last_error());
		return;
	}
	ms_bufferizer_flush(&es->delayed_ref);
	ms_bufferizer_flush(&es->echo);
	// (a leg that leaves before its first enqueue: the delay line's zeroes are still a count, LegBank::used)
	const int pending = std::min(leg->inject, leg->dref_level);
	if (pending > 0) {
		std::vector<int16_t> z((size_t)pending, 0);
		bufferizer_put_samples(&es->delayed_ref, z.data(), pending);
		leg->inject -= pending, leg->dref_level -= pending;
	}
	const bool ok = fifo_take(b->hub->ctx, b->f_ref, b->nlegs, leg->slot, b->ns, b->d_scratch, b->d_dgate_any(), {leg->dref_level},
	                          [&](int, const int16_t *x, int n) { bufferizer_put_samples(&es->delayed_ref, x, n); }) &&
	                fifo_take(b->hub->ctx, b->f_mic, b->nlegs, leg->slot, b->ns, b->d_scratch, b->d_dgate_any(), {leg->echo_level},
	                          [&](int, const int16_t *x, int n) { bufferizer_put_samples(&es->echo, x, n); });
	if (!ok) mi_failed("taking a leg's canceller queues back");
	leg->dref_level = leg->echo_level = 0;
	es->echostarted = started ? TRUE : FALSE;
}

// The conference leaves its LegBank: at detach (every facade's postprocess ends up here, the first one does the work) or,
// keep_running, because a member stopped qualifying while attached -- the facades then go on with banks of their own and
// the canceller's queues on the device are dropped (its postprocess flushes them, speexec.c:305-319).
void conf_unfuse(MSFilter *mx, bool keep_running) {
	MixerState *ms = (MixerState *)mx->data;
	LegBank *b = ms->fbank;
	if (!b) {
		server_unfuse(mx, keep_running);
		return;
	}
	HubLock lk(b->hub);
	const int c = ms->fconf, mm = b->mm;
	std::vector<FusedLeg *> gone;
	b->deliver_in_flight(mx, c);
	b->settle_meters();
	b->take_remainders(c * mm, mm, keep_running);
	const size_t rs_each = (b->rs && !b->failed) ? (size_t)mi_resampler_state_bytes(b->rs) : 0;
	std::vector<uint8_t> rs_buf(rs_each * (size_t)mm);
	const bool rs_ok = rs_each && mi_resampler_get_states(b->rs, c * mm, mm, rs_buf.data(), rs_buf.size()) == MI_OK; // (the members' in one round trip)
	for (int pin = 0; pin < mm; ++pin) {
		FusedLeg *leg = b->legs[(size_t)(c * mm + pin)];
		if (!leg) continue;
		leg_keep_volume(leg);
		leg_drop_peer(b, leg);
		leg_drop_equalizer(b, leg);
		if (leg->rs && rs_ok) resample_keep_bytes((ResampleData *)leg->rs->data, rs_buf.data() + (size_t)pin * rs_each, rs_each, b->in_rate, b->rate);
		b->legs[(size_t)(c * mm + pin)] = nullptr;
		if (leg->rs) ((ResampleData *)leg->rs->data)->leg = nullptr;
		((SpeexECState *)leg->ec->data)->leg = nullptr;
		((VolumeData *)leg->vol->data)->leg = nullptr;
		gone.push_back(leg);
	}
	b->conf_ready[(size_t)c] = 0;
	ms->fbank = nullptr, ms->fconf = -1;
	ms->fuse_state = keep_running ? 2 : 0; // a new attach looks again
	ms->unfuse_wanted = false;
	for (int pin = 0; pin < mm; ++pin) b->flags[(size_t)(c * mm + pin)] = 0;
	b->ctl_dirty = true;
	if (keep_running) { // banks of their own again, while the hub is still held by this conference's slot
		for (FusedLeg *leg : gone) {
			const bool started = ((SpeexECState *)leg->ec->data)->echostarted != FALSE;
			ec_prepare(leg->ec);
			ec_acquire(leg->ec);
			leg_return_canceller(b, leg, started);
		}
		mixer_prepare(mx, true);
		ms_warning("mi355x: conference %p left its fused batch (a member's configuration changed); the facades carry on one by one", (void *)mx);
	}
	b->release(c); // (may destroy the bank)
	for (FusedLeg *leg : gone) delete leg;
}

// ---- a leg WITHOUT a mixer:  MSResample -> MSSpeexEC pin 1 -> MSVolume (AGC) -> any other filter, all ours on one ticker
bool leg_fuse_plain_at(MSFilter *rs, MSFilter *ec);
bool leg_try_fuse_plain(MSFilter *rs) {
	if (!ms_queue_empty(rs->outputs[0])) return false;
	MSQueue *q = leg_past_equalizer(rs->outputs[0]);
	MSFilter *ec = q ? q->next.filter : NULL;
	if (!ec || !is_ec_desc(ec->desc) || q->next.pin != 1 || !ms_queue_empty(q)) return false;
	return leg_fuse_plain_at(rs, ec);
}
// the same with MSSpeexEC as the head (its microphone pin fed by anything but our MSResample)
bool leg_try_fuse_plain_ec(MSFilter *ec) { return leg_fuse_plain_at(nullptr, ec); }
bool leg_fuse_plain_at(MSFilter *rs, MSFilter *ec) {
	MSFilter *head = rs ? rs : ec;
	MSFilter *eqf = ec->inputs[1] ? ec->inputs[1]->prev.filter : NULL; // a mic_equalizer of ours between the two (audiostream.c:1801)?
	if (eqf && equalizer_passes(eqf, head->ticker) && ms_queue_empty(ec->inputs[1])) eqf = nullptr; // (not active: transparent -- it forwards in the walk, or, behind a fused MSResample, sees nothing)
	if (eqf && eqf->desc == &ms_mi355x_equalizer_desc) {
		if (!rs || !ms_queue_empty(ec->inputs[1]) || !leg_equalizer_ok(eqf, head->ticker, ((SpeexECState *)ec->data)->samplerate)) return false;
	} else eqf = nullptr;
	if (getenv("MSMI355X_NO_FUSE") != nullptr || !head->ticker || head->ticker->interval != 10 || ec->ticker != head->ticker) return false;
	ResampleData *rd = rs ? (ResampleData *)rs->data : nullptr;
	SpeexECState *es = (SpeexECState *)ec->data;
	if (es->bypass_mode || es->unsupported || !es->configured || es->echostarted || es->leg || (rd && (uint32_t)es->samplerate != rd->output_rate) || es->samplerate % 100) return false;
	if (ms_bufferizer_get_avail(&es->echo) || (int)ms_bufferizer_get_avail(&es->delayed_ref) != es->nominal_ref_samples * 2) return false;
	MSQueue *qv = ec->outputs[1];
	MSFilter *vol = qv ? qv->next.filter : NULL;
	if (!vol || vol->desc != &ms_mi355x_volume_desc || vol->ticker != head->ticker || !ms_queue_empty(qv) || !vol->outputs[0]) return false;
	VolumeData *vd = (VolumeData *)vol->data;
	MSFilter *peer = nullptr;
	if (!leg_peer_ok(vol, vd, &peer) || vd->sample_rate != es->samplerate || vd->leg) return false;
	if (!leg_remainder_ok(vd, 1) || ms_bufferizer_get_avail(vd->buffer) >= (size_t)(vd->sample_rate / 100) * 2 || ms_bufferizer_get_avail(vd->spill)) return false; // (MSVolume holds less than a chunk between blocks)
	const bool no_agc = !volume_chunks(vd); // (10 ms chunks with AGC or an echo-limiter peer, msvolume.c:480)
	if (!leg_far_end_in_walk(ec, peer)) return false;
	if (rd && (rd->in_nchannels != 1 || rd->out_nchannels != 1 || !leg_rates_ok(rd->input_rate, rd->output_rate) || rd->leg || ms_bufferizer_get_avail(rd->bz))) return false;
	const uint32_t rate = (uint32_t)es->samplerate, ir = rd ? rd->input_rate : rate;
	const int F = es->framesize, flen = es->filterlength, delay = es->nominal_ref_samples;
	// volsend -> [outbound_mixer that can only forward] -> MSAlawEnc / MSUlawEnc of ours (audiostream.c:1803-1809 without dtmfgen_rtp: a
	// telephone-event payload is negotiated, :1396-1404): the chunks are encoded in the batch
	MSFilter *omix = nullptr, *encf = vol->outputs[0]->next.filter;
	if (is_forwarding_mixer(encf, head->ticker) && ms_queue_empty(encf->outputs[0])) omix = encf, encf = encf->outputs[0]->next.filter;
	if (rate == 8000 && encf && is_g711_enc(encf->desc) && encf->ticker == head->ticker && encf->inputs[0] && ms_queue_empty(encf->inputs[0])) {
		MapFilter *ed = (MapFilter *)encf->data;
		if (ed->sleg || ed->fleg || ms_bufferizer_get_avail(ed->bz) || (ed->pool && (!ed->pool->staged[(size_t)ed->slot].empty() || !ed->pool->ready[(size_t)ed->slot].empty()))) encf = nullptr;
	} else encf = nullptr;
	if (!encf) omix = nullptr;
	const int law = encf ? (((MapFilter *)encf->data)->law ? MI_LAW_PCMU : MI_LAW_PCMA) : -1;
	LegBank *b = bank<LegBank>("legp:" + std::to_string(ir) + ":" + std::to_string(rate) + ":" + std::to_string(F) + ":" + std::to_string(flen) + ":" +
	                               std::to_string(delay) + (no_agc ? ":light" : "") + (eqf ? ":eq" : "") + (encf ? ":enc" + std::to_string(law) : ""),
	                           1, [&](int cap) { return new LegBank(cap * 4, ir, rate, F, flen, delay, 1, true, no_agc, eqf != nullptr, law); }); // 64, 256, 1024, .. legs
	const int s = b ? b->acquire(vol) : -1;
	if (s < 0) return false;
	note_slot(vol);
	const bool was_used = b->used[(size_t)s] != 0;
	bool ok = b->start_slots(s, 1);
	volume_keep_state(vd); // (its bank is on this hub, which is held)
	{
		mi_volume_params vp = vd->p;
		vp.peer = peer ? MI_VOLUME_PEER_EXTERNAL : -1;
		b->start_volume((size_t)s, vp, volume_start_state(vd), was_used);
	}
	if (es->state_str) {
		std::vector<uint8_t> blob;
		if (b64_decode(es->state_str, blob) && mi_aec_import_state(b->aec, s, blob.data(), blob.size()) == MI_OK) ms_message("mi355x echo state restored.");
		else ms_error("Could not apply mi355x echo blob: %s", mi_last_error());
	}
	if (!ok) {
		mi_failed("fusing a call leg");
		b->release(s);
		return false;
	}
	FusedLeg *leg = new FusedLeg();
	leg->lv_from = b->lv_seq + 1;
	leg->bank = b, leg->slot = s, leg->pin = 0;
	leg->rs = rs, leg->ec = ec, leg->vol = vol, leg->mixer = nullptr;
	leg->rs_data = rs ? rs->data : nullptr, leg->ec_data = ec->data, leg->vol_data = vol->data;
	leg->dref_level = leg->inject = delay; // zeroes for the time of the delay (speexec.c:205-208): pushed by the first enqueue
	b->legs[(size_t)s] = leg;
	b->nout[(size_t)s] = b->nready[(size_t)s] = 0;
	if (rd) {
		if (rd->pool) {
			if (rd->slots->size() == 1) resample_keep_from(rd, rd->pool->r, rd->slot, rd->input_rate, rd->output_rate);
			resample_release(rd);
		}
		resample_restore_to(rd, b->rs, s, b->in_rate, b->rate, true);
		rd->leg = leg;
	}
	ms_bufferizer_flush(&es->delayed_ref);
	if (es->pool) {
		es->pool->staged[(size_t)es->slot] = es->pool->ready[(size_t)es->slot] = 0;
		es->pool->release(es->slot);
		es->pool = nullptr, es->slot = -1;
	}
	es->leg = leg;
	if (vd->pool) {
		vd->pool->release(vd->slot);
		vd->pool = nullptr, vd->slot = -1;
	}
	vd->leg = leg;
	b->give_begin(s, 1);
	if (!b->give_remainder(s, vd, leg) || !b->give_end()) mi_failed("moving MSVolume's queued samples to the device");
	if (peer && !leg_take_peer(b, leg, peer)) mi_failed("taking the echo limiter's peer into the batch");
	if (eqf && !leg_take_equalizer(b, leg, eqf)) mi_failed("taking the leg's equalizer into the batch");
	if (eqf && b->eq && mi_equalizer_prepare(b->eq) != MI_OK) mi_failed("designing the leg's equalizer"); // (the taps' design -- a host FFT per stream -- here, on the attaching thread, not under the first tick's launch)
	if (encf) {
		MapFilter *ed = (MapFilter *)encf->data;
		map_release(ed); // (its own bank's slot, if it ever had one)
		ed->fleg = leg;
		leg->enc = encf, leg->omix = omix;
	}
	b->staged_since = true;
	ms_message("mi355x: call leg %p fused: %u -> %u Hz, frame %d, tail %d (%sMSSpeexEC -> MSVolume%s as one device-resident batch)", (void *)vol, ir, rate, F, flen,
	           rs ? "MSResample -> " : "", encf ? (law == MI_LAW_PCMU ? " -> MSUlawEnc" : " -> MSAlawEnc") : "");
	return true;
}

void leg_unfuse_plain(FusedLeg *leg, bool keep_running) {
	LegBank *b = leg->bank;
	HubLock lk(b->hub);
	const int s = leg->slot;
	b->deliver_in_flight(leg->vol, s);
	b->settle_meters();
	b->take_remainders(s, 1, keep_running);
	leg_keep_volume(leg);
	leg_drop_peer(b, leg);
	leg_drop_equalizer(b, leg);
	if (leg->rs && b->rs && !b->failed) resample_keep_from((ResampleData *)leg->rs->data, b->rs, s, b->in_rate, b->rate);
	b->legs[(size_t)s] = nullptr;
	b->nout[(size_t)s] = b->nready[(size_t)s] = 0;
	if (leg->rs) ((ResampleData *)leg->rs->data)->leg = nullptr;
	((SpeexECState *)leg->ec->data)->leg = nullptr;
	((VolumeData *)leg->vol->data)->leg = nullptr;
	if (leg->enc) ((MapFilter *)leg->enc->data)->fleg = nullptr; // (the packet it is filling stays with the facade: MapFilter::pending)
	if (keep_running) {
		const bool started = ((SpeexECState *)leg->ec->data)->echostarted != FALSE;
		ec_prepare(leg->ec); // a bank slot of its own again, while the hub is still held by this leg's slot
		ec_acquire(leg->ec);
		leg_return_canceller(b, leg, started);
		ms_warning("mi355x: call leg %p left its fused batch (a member's configuration changed); the facades carry on one by one", (void *)leg->vol);
	}
	b->release(s); // (may destroy the bank)
	delete leg;
}

void leg_release(FusedLeg *leg, bool keep_running) {
	if (!leg) return;
	if (leg->mixer) conf_unfuse(leg->mixer, keep_running);
	else leg_unfuse_plain(leg, keep_running);
}
// (a conference's flag lives with its mixer: the FIRST of its members to be walked takes the conference out, before anything of that
// walk is staged -- the mixer itself runs behind all of them and would find a tick's rows staged in a bank it is about to leave)
bool leg_wants_out(FusedLeg *leg) {
	if (!leg) return false;
	return leg->mixer ? ((MixerState *)leg->mixer->data)->unfuse_wanted.load() : leg->unfuse_wanted.load();
}
bool leg_has_resampler(FusedLeg *leg) { return leg && leg->rs != nullptr; }

Pool *leg_pool(FusedLeg *leg) { return leg->bank; }
Pool *leg_pool_of(LegBank *b) { return b; }
// MS_AUDIO_MIXER_SET_INPUT_GAIN / SET_ACTIVE / ENABLE_OUTPUT on a fused conference (hub locked): the bank's control rows
void leg_push_mixer_controls(MSFilter *f, MixerState *s, bool from_method) {
	LegBank *b = s->fbank;
	const bool later = from_method && b->work_waiting();
	std::vector<uint8_t> &fl_row = later ? b->next_flags : b->flags;
	std::vector<float> &g_row = later ? b->next_gains : b->gains;
	for (int pin = 0; pin < b->mm; ++pin) {
		const size_t at = (size_t)(s->fconf * b->mm + pin);
		uint8_t fl = 0;
		if (f->inputs[pin] && b->legs[at]) fl |= MI_MIX_LINKED;
		if (s->channels[pin].active) fl |= MI_MIX_ACTIVE;
		if (f->outputs[pin] && s->channels[pin].output_enabled) fl |= MI_MIX_OUTPUT;
		fl_row[at] = fl;
		g_row[at] = s->channels[pin].gain;
	}
	if (later) b->next_conf[(size_t)s->fconf] = 1, b->next_any = true;
	else b->next_conf[(size_t)s->fconf] = 0, b->ctl_dirty = true;
}
mi_volume_state *leg_vstate(FusedLeg *leg) { return &leg->bank->vstate[(size_t)leg->slot]; }
// MS_VOLUME_* methods on a fused leg's MSVolume (hub locked): parameters / running state for the next flush
void leg_push_volume(FusedLeg *leg, const mi_volume_params *p, const float *gain, const float *target) {
	LegBank *b = leg->bank;
	const size_t s = (size_t)leg->slot;
	const uint8_t when = b->work_waiting() ? 2 : 1; // (2: behind the coming flush, LegBank::flushed)
	if (when == 1) b->v_delay[s] = b->chunks_waiting(s);
	b->vparams[s] = *p;
	b->vparams[s].peer = leg->peer ? MI_VOLUME_PEER_EXTERNAL : -1;
	b->vp_dirty[s] = when;
	if (gain) {
		b->vpatch[s] = {*gain, target ? *target : 0.f, target != nullptr};
		b->vs_dirty[s] = when;
	}
	b->v_dirty = true;
}
mi_aec *leg_canceller(FusedLeg *leg, int *slot) {
	*slot = leg->slot;
	return leg->bank->aec;
}

// a facade of a fused leg stopped qualifying (a method call on the application's thread): the conference leaves the batch
// at the start of the next flush -- on the ticker thread, where the facades' state may be touched
void leg_disqualify(FusedLeg *leg) {
	if (!leg) return;
	if (leg->mixer) ((MixerState *)leg->mixer->data)->unfuse_wanted = true;
	else leg->unfuse_wanted = true; // (honoured by the leg's MSResample at its next block)
}
