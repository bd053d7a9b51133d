// filters.cpp -- the MI355X filter plugin: MSFilterDesc facades registered under the
// reference's MS_*_ID so that ms_factory_create_filter(f, MS_RESAMPLE_ID) etc. hand out
// these instead of the CPU filters (registration prepends, lookup is first-match:
// src/base/msfactory.c:281,:440-450; plugins load after the built-ins: src/voip/msvoip.c:369-374).
//
// What stays on the host, exactly as in the reference: queues, bufferizers and the
// per-stream framing state machines (mixer bypass/flow control, EC zero injection,
// volume re-framing), method tables, locking.  What moves to the GPU: the sample
// loops, through the C ABI of include/msmi355x.h.
//
// Batching: the reference runs one process() per filter per tick (src/base/msticker.c:244-259).
// Here process() STAGES its 10 ms block into a slot of a per-type pool and emits the result
// of the PREVIOUS tick; one postponed ticker task (src/base/msfilter.c:289-300, run before the
// graphs of the next tick, msticker.c:301-312) launches every staged pool: one kernel per
// filter type for all streams.  Cost: one tick (10 ms) of added latency per GPU filter; this
// is the only scheduling-compatible option without touching the ticker.
#include "../../include/ms2_plugin_abi.h"
#include "../../include/msmi355x.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

constexpr int kMaxRounds = 4; // blocks one stream may hand over within a single tick

[[noreturn]] void die(const char *what) {
	ms_error("msmi355x plugin: %s: %s", what, mi_last_error());
	fprintf(stderr, "msmi355x plugin: %s: %s (there is no CPU fallback)\n", what, mi_last_error());
	abort();
}
#define MI_MUST(expr)                     \
	do {                                  \
		if ((expr) != MI_OK) die(#expr);  \
	} while (0)

struct Pool {
	virtual ~Pool() {}
	virtual void flush() = 0;               // launch the staged blocks, fetch the results
	virtual void emit(MSFilter *f, int slot) = 0; // hand a slot's results to its filter's output queues
	std::vector<uint8_t> used;
	std::vector<MSFilter *> owner;
	MSTicker *ticker = nullptr; // a pool serves the filters of ONE ticker thread
	int capacity = 0;
	void init_slots(int cap) {
		capacity = cap;
		used.assign((size_t)cap, 0);
		owner.assign((size_t)cap, nullptr);
	}
	int acquire(MSFilter *f) {
		for (int i = 0; i < capacity; ++i)
			if (!used[(size_t)i]) {
				used[(size_t)i] = 1;
				owner[(size_t)i] = f;
				return i;
			}
		ms_error("msmi355x plugin: pool exhausted (%d slots; raise MSMI355X_SLOTS)", capacity);
		return -1;
	}
	void release(int slot) {
		used[(size_t)slot] = 0;
		owner[(size_t)slot] = nullptr;
	}
	void emit_all() {
		for (int i = 0; i < capacity; ++i)
			if (used[(size_t)i] && owner[(size_t)i]) emit(owner[(size_t)i], i);
	}
};

struct Hub {
	std::recursive_mutex mu;
	mi_ctx *ctx = nullptr;
	int capacity = 256;
	std::map<MSTicker *, bool> flush_pending;
	std::vector<Pool *> pools;
	mi_ctx *context() {
		if (!ctx) {
			const char *cap = getenv("MSMI355X_SLOTS");
			if (cap && atoi(cap) > 0) capacity = atoi(cap);
			const char *dev = getenv("MSMI355X_DEVICE");
			if (mi_ctx_create(dev ? atoi(dev) : 0, nullptr, &ctx) != MI_OK) die("mi_ctx_create");
		}
		return ctx;
	}
};
Hub g_hub;

template <typename T>
T *pinned(size_t n) {
	void *p = mi_host_alloc(g_hub.context(), n * sizeof(T));
	if (!p) die("mi_host_alloc");
	memset(p, 0, n * sizeof(T));
	return (T *)p;
}
template <typename T>
T *devmem(size_t n) {
	void *p = mi_dev_alloc(g_hub.context(), n * sizeof(T));
	if (!p) die("mi_dev_alloc");
	return (T *)p;
}

void flush_ticker(MSTicker *t) {
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (t == nullptr) g_hub.flush_pending.clear(); // the requesting filter was detached meanwhile: flush everything
	else g_hub.flush_pending[t] = false;
	for (Pool *p : g_hub.pools)
		if (t == nullptr || p->ticker == t) {
			p->flush();
			p->emit_all();
		}
}

// Runs on the ticker thread at the start of the next tick, before any process() (msticker.c:301-312):
// one launch per staged pool, then the results go straight into the owners' output queues, so the
// downstream filters see them in this tick's graph run even if the owner itself gets no new input.
void flush_task(MSFilter *f) { flush_ticker(f->ticker); }

// called by a filter that staged work this tick
void request_flush(MSFilter *f) {
	if (!g_hub.flush_pending[f->ticker]) {
		g_hub.flush_pending[f->ticker] = true;
		ms_filter_postpone_task(f, flush_task);
	}
}

// =================================================================== resampler
struct ResamplePool : Pool {
	uint32_t in_rate, out_rate;
	int in_len, ostride;
	mi_resampler *r = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	int32_t *h_olen, *d_olen;
	uint8_t *h_run, *d_run;
	std::vector<int> staged, ready;
	ResamplePool(uint32_t ir, uint32_t orate) : in_rate(ir), out_rate(orate) {
		init_slots(g_hub.capacity);
		MI_MUST(mi_resampler_create(g_hub.context(), capacity, ir, orate, 3 /* SPEEX_RESAMPLER_QUALITY_VOIP */, &r));
		in_len = (int)(ir / 100);
		ostride = (mi_resampler_out_capacity(r, in_len) + 7) & ~7;
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
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity;
		int maxr = 0;
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r_ = 0; r_ < maxr; ++r_) {
			for (int s = 0; s < capacity; ++s) h_run[r_ * c + s] = staged[(size_t)s] > r_;
			MI_MUST(mi_copy_h2d(ctx, d_in, h_in + r_ * c * in_len, c * in_len * 2));
			MI_MUST(mi_copy_h2d(ctx, d_run, h_run + r_ * c, c));
			MI_MUST(mi_resampler_process_masked(r, d_in, in_len, in_len, d_out, ostride, d_olen, d_run));
			MI_MUST(mi_copy_d2h(ctx, h_out + r_ * c * ostride, d_out, c * ostride * 2));
			MI_MUST(mi_copy_d2h(ctx, h_olen + r_ * c, d_olen, c * 4));
		}
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
	void emit(MSFilter *f, int slot) override;
};
std::map<std::tuple<MSTicker *, uint32_t, uint32_t>, ResamplePool *> g_resample_pools;

struct ResampleData { // ResampleData msresample.c:33-42
	MSBufferizer *bz;
	uint32_t ts;
	uint32_t input_rate, output_rate;
	int in_nchannels, out_nchannels;
	ResamplePool *pool;
	int slot;                 // first channel's slot (the one that emits)
	std::vector<int> *slots;  // one batch slot per input channel (speex keeps one state per channel too)
};

void resample_init(MSFilter *f) { // msresample.c:44-54,:62-80
	ResampleData *d = (ResampleData *)ms_malloc0(sizeof(*d));
	d->bz = ms_bufferizer_new();
	d->input_rate = 8000;
	d->output_rate = 16000;
	d->in_nchannels = d->out_nchannels = 1;
	d->slot = -1;
	d->slots = new std::vector<int>();
	f->data = d;
}

void resample_release(ResampleData *d) {
	if (d->pool) {
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		for (int sl : *d->slots) {
			d->pool->release(sl);
			d->pool->staged[(size_t)sl] = d->pool->ready[(size_t)sl] = 0;
			MI_MUST(mi_resampler_reset(d->pool->r, sl, 1));
		}
	}
	d->slots->clear();
	d->pool = nullptr;
	d->slot = -1;
}

void resample_uninit(MSFilter *f) {
	ResampleData *d = (ResampleData *)f->data;
	resample_release(d);
	ms_bufferizer_destroy(d->bz);
	delete d->slots;
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
	ms_filter_lock(f);
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	const int nch = d->in_nchannels < 1 ? 1 : d->in_nchannels;
	if (d->pool && (d->pool->in_rate != d->input_rate || d->pool->out_rate != d->output_rate || (int)d->slots->size() != nch))
		resample_release(d); // rates / channels changed: the handle is re-created, history lost (:138-148)
	if (!d->pool) {
		auto key = std::make_tuple(f->ticker, d->input_rate, d->output_rate);
		auto it = g_resample_pools.find(key);
		if (it == g_resample_pools.end()) {
			ResamplePool *p = new ResamplePool(d->input_rate, d->output_rate);
			p->ticker = f->ticker;
			g_hub.pools.push_back(p);
			it = g_resample_pools.emplace(key, p).first;
		}
		d->pool = it->second;
		for (int ch = 0; ch < nch; ++ch) { // interleaved input: one state per channel, like speex_resampler_init(nb_channels)
			const int sl = d->pool->acquire(f);
			if (sl < 0) break;
			d->slots->push_back(sl);
		}
		if ((int)d->slots->size() != nch) {
			resample_release(d);
			ms_queue_flush(f->inputs[0]);
			ms_filter_unlock(f);
			return;
		}
		d->slot = (*d->slots)[0];
	}
	ResamplePool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	// this tick's input, re-framed to 10 ms blocks (a streaming filter: the sample sequence is
	// independent of the blocking); the results are emitted by the flush task (ResamplePool::emit)
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	const size_t nbytes = (size_t)p->in_len * 2 * (size_t)nch;
	std::vector<int16_t> frame;
	while (p->staged[s] < kMaxRounds && ms_bufferizer_get_avail(d->bz) >= nbytes) {
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
	ms_filter_lock(f);
	d->input_rate = *(unsigned int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_output_sr(MSFilter *f, void *arg) { // :194-205
	ResampleData *d = (ResampleData *)f->data;
	ms_filter_lock(f);
	d->output_rate = *(unsigned int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_in_nch(MSFilter *f, void *arg) {
	ResampleData *d = (ResampleData *)f->data;
	ms_filter_lock(f);
	d->in_nchannels = *(int *)arg;
	ms_filter_unlock(f);
	return 0;
}
int resample_set_out_nch(MSFilter *f, void *arg) {
	ResampleData *d = (ResampleData *)f->data;
	ms_filter_lock(f);
	d->out_nchannels = *(int *)arg;
	ms_filter_unlock(f);
	return 0;
}
MSFilterMethod resample_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, resample_set_sr},
                                     {MS_FILTER_SET_OUTPUT_SAMPLE_RATE, resample_set_output_sr},
                                     {MS_FILTER_SET_NCHANNELS, resample_set_in_nch},
                                     {MS_FILTER_SET_OUTPUT_NCHANNELS, resample_set_out_nch},
                                     {0, NULL}};

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
	std::vector<int> staged, ready;
	std::vector<mi_volume_params> params;
	std::vector<mi_volume_state> state;
	std::vector<uint8_t> params_dirty, state_dirty;
	VolumePool(int r) : rate(r) {
		init_slots(g_hub.capacity);
		MI_MUST(mi_volume_create(g_hub.context(), capacity, rate, &v));
		cap_samples = std::max(960, rate / 100 * 2);
		cap_samples = (cap_samples + 7) & ~7;
		const size_t c = (size_t)capacity;
		h_buf = pinned<int16_t>(kMaxRounds * c * cap_samples);
		h_n = pinned<int32_t>(kMaxRounds * c);
		d_buf = devmem<int16_t>(c * cap_samples);
		d_n = devmem<int32_t>(c);
		staged.assign(c, 0);
		ready.assign(c, 0);
		mi_volume_params p;
		mi_volume_default_params(&p);
		params.assign(c, p);
		state.resize(c);
		MI_MUST(mi_volume_get_state(v, 0, capacity, state.data()));
		params_dirty.assign(c, 0);
		state_dirty.assign(c, 0);
	}
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity;
		for (int s = 0; s < capacity; ++s) {
			if (params_dirty[(size_t)s]) MI_MUST(mi_volume_set_params(v, s, 1, &params[(size_t)s]));
			if (state_dirty[(size_t)s]) MI_MUST(mi_volume_set_state(v, s, 1, &state[(size_t)s]));
			params_dirty[(size_t)s] = state_dirty[(size_t)s] = 0;
		}
		int maxr = 0;
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			for (int s = 0; s < capacity; ++s)
				if (staged[(size_t)s] <= r) h_n[r * c + s] = 0;
			MI_MUST(mi_copy_h2d(ctx, d_buf, h_buf + r * c * cap_samples, c * cap_samples * 2));
			MI_MUST(mi_copy_h2d(ctx, d_n, h_n + r * c, c * 4));
			MI_MUST(mi_volume_process(v, d_buf, cap_samples, cap_samples, d_n));
			MI_MUST(mi_copy_d2h(ctx, h_buf + r * c * cap_samples, d_buf, c * cap_samples * 2));
		}
		if (maxr) {
			MI_MUST(mi_ctx_sync(ctx));
			MI_MUST(mi_volume_get_state(v, 0, capacity, state.data())); // meters for the app thread
		}
		for (int s = 0; s < capacity; ++s) {
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
	void emit(MSFilter *f, int slot) override;
};
std::map<std::pair<MSTicker *, int>, VolumePool *> g_volume_pools;

struct VolumeData { // struct Volume msvolume.c:48-86, host-side part
	mi_volume_params p;
	float gain, target_gain; // pending values for a slot not yet acquired
	int sample_rate, nsamples;
	MSFilter *peer;
	MSBufferizer *buffer;
	MSBufferizer *spill; // light path: the part of an over-long block that did not fit this tick's rounds
	Extremum min, max;
	VolumePool *pool;
	int slot;
	bool ng_soft_start;
};

void volume_init(MSFilter *f) { // msvolume.c:88-118
	VolumeData *d = new VolumeData();
	mi_volume_default_params(&d->p);
	d->gain = d->target_gain = 1;
	d->sample_rate = 8000;
	d->nsamples = 80;
	d->peer = NULL;
	d->buffer = ms_bufferizer_new();
	d->spill = ms_bufferizer_new();
	d->max.period = 1000;
	d->min.period = 30000;
	d->pool = nullptr;
	d->slot = -1;
	f->data = d;
}

void volume_uninit(MSFilter *f) {
	VolumeData *d = (VolumeData *)f->data;
	if (d->pool && d->slot >= 0) {
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		d->pool->release(d->slot);
	}
	ms_bufferizer_destroy(d->buffer);
	ms_bufferizer_destroy(d->spill);
	delete d;
}

mi_volume_state *vstate(VolumeData *d) { return (d->pool && d->slot >= 0) ? &d->pool->state[(size_t)d->slot] : nullptr; }

void volume_push_params(VolumeData *d) {
	if (!d->pool || d->slot < 0) return;
	d->pool->params[(size_t)d->slot] = d->p;
	d->pool->params_dirty[(size_t)d->slot] = 1;
}

void volume_attach_slot(MSFilter *f) {
	VolumeData *d = (VolumeData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (d->pool && (d->pool->rate != d->sample_rate || d->pool->ticker != f->ticker)) {
		d->pool->release(d->slot);
		d->pool = nullptr;
		d->slot = -1;
	}
	if (!d->pool) {
		auto key = std::make_pair(f->ticker, d->sample_rate);
		auto it = g_volume_pools.find(key);
		if (it == g_volume_pools.end()) {
			VolumePool *p = new VolumePool(d->sample_rate);
			p->ticker = f->ticker;
			g_hub.pools.push_back(p);
			it = g_volume_pools.emplace(key, p).first;
		}
		d->pool = it->second;
		d->slot = d->pool->acquire(f);
		if (d->slot < 0) {
			d->pool = nullptr;
			return;
		}
		// fresh slot: volume_init state, then whatever the methods set before attach
		mi_volume_state st;
		memset(&st, 0, sizeof(st));
		st.gain = d->gain;
		st.target_gain = d->target_gain;
		st.ng_gain = 1;
		d->pool->state[(size_t)d->slot] = st;
		d->pool->state_dirty[(size_t)d->slot] = 1;
	}
	// the peer is addressed by its slot in the same pool
	d->p.peer = -1;
	if (d->peer) {
		VolumeData *pd = (VolumeData *)d->peer->data;
		if (pd->pool == d->pool && pd->slot >= 0) d->p.peer = pd->slot;
		else ms_warning("MSVolume[mi355x]: peer not in the same batch yet (different rate or not attached)");
	}
	volume_push_params(d);
}

void volume_preprocess(MSFilter *f) { // msvolume.c:447-469
	VolumeData *d = (VolumeData *)f->data;
	d->nsamples = (int)(0.01 * (float)d->sample_rate);
	d->min.reset();
	d->max.reset();
	volume_attach_slot(f);
}

void volume_process(MSFilter *f) { // msvolume.c:471-514
	VolumeData *d = (VolumeData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (!d->pool) volume_attach_slot(f);
	if (!d->pool) {
		ms_queue_flush(f->inputs[0]);
		return;
	}
	if (d->peer && d->p.peer < 0) volume_attach_slot(f);
	VolumePool *p = d->pool;
	const size_t c = (size_t)p->capacity, s = (size_t)d->slot;
	mblk_t *m;
	if (d->p.agc_enabled || d->peer != NULL) { // :480-503 re-framed to 10 ms chunks
		const size_t nbytes = (size_t)d->nsamples * 2;
		ms_bufferizer_put_from_queue(d->buffer, f->inputs[0]);
		while (p->staged[s] < kMaxRounds && ms_bufferizer_get_avail(d->buffer) >= nbytes) {
			ms_bufferizer_read(d->buffer, (uint8_t *)(p->h_buf + (p->staged[s] * c + s) * p->cap_samples), nbytes);
			p->h_n[p->staged[s] * c + s] = d->nsamples;
			p->staged[s]++;
		}
	} else { // :505-512 light path: one chunk per mblk.  A block longer than a batch row (20 ms and more than 960 samples)
		// is cut into row-sized chunks -- no sample is dropped; the meter then sees those chunks, not the whole block.
		for (;;) {
			if (p->staged[s] >= kMaxRounds) break;
			int16_t *row = p->h_buf + (p->staged[s] * c + s) * p->cap_samples;
			int n = 0;
			const size_t spilled = ms_bufferizer_get_avail(d->spill);
			if (spilled) {
				n = (int)std::min(spilled / 2, (size_t)p->cap_samples);
				ms_bufferizer_read(d->spill, (uint8_t *)row, (size_t)n * 2);
			} else if ((m = ms_queue_get(f->inputs[0])) != NULL) {
				n = (int)(msgdsize(m) / 2);
				if (n > p->cap_samples) {
					ms_bufferizer_put(d->spill, m); // served chunk by chunk from the top of the loop
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
	}
	if (p->staged[s]) request_flush(f);
}

void VolumePool::emit(MSFilter *f, int slot) {
	VolumeData *d = (VolumeData *)f->data;
	const size_t c = (size_t)capacity, s = (size_t)slot;
	for (int r = 0; r < ready[s]; ++r) {
		const int n = h_n[r * c + s];
		mblk_t *om = allocb((size_t)n * 2, 0);
		memcpy(om->b_wptr, h_buf + (r * c + s) * cap_samples, (size_t)n * 2);
		om->b_wptr += n * 2;
		if (f->outputs[0]) ms_queue_put(f->outputs[0], om);
		else freemsg(om);
	}
	if (ready[s] && f->ticker) { // meters (update_energy msvolume.c:405-406)
		d->max.record_max(f->ticker->time, state[s].energy);
		d->min.record_min(f->ticker->time, state[s].energy);
	}
	ready[s] = 0;
}

float linear_to_dbm0(float linear) { // ms_volume_linear_to_dbm0 msvolume.c:565-568
	if (linear == 0) return MS_VOLUME_DB_LOWEST;
	return 10 * log10f(linear);
}

int volume_get(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	mi_volume_state *st = vstate(d);
	*(float *)arg = linear_to_dbm0(st ? st->energy : 0.f);
	return 0;
}
int volume_get_linear(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
void volume_set_gains(VolumeData *d, bool also_target) {
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	mi_volume_state *st = vstate(d);
	if (st) {
		st->gain = d->gain;
		if (also_target) st->target_gain = d->target_gain;
		d->pool->state_dirty[(size_t)d->slot] = 1;
	}
	volume_push_params(d);
}
int volume_set_gain(MSFilter *f, void *arg) { // :270-276
	VolumeData *d = (VolumeData *)f->data;
	d->gain = d->target_gain = d->p.static_gain = *(float *)arg;
	volume_set_gains(d, true);
	return 0;
}
int volume_set_db_gain(MSFilter *f, void *arg) { // :262-268 (power ratio)
	VolumeData *d = (VolumeData *)f->data;
	d->gain = d->p.static_gain = (float)pow(10, (*(float *)arg) / 10);
	volume_set_gains(d, false);
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
	d->peer = (MSFilter *)arg;
	if (d->pool) volume_attach_slot(f);
	return 0;
}
int volume_set_rate(MSFilter *f, void *arg) {
	((VolumeData *)f->data)->sample_rate = *(int *)arg;
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
		d->p.field = val;                                          \
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);        \
		volume_push_params(d);                                     \
		return 0;                                                  \
	}
VOL_FLOAT_SETTER(volume_set_ea_threshold, ea_thres, val >= 0 && val <= 1) // :305-314
VOL_FLOAT_SETTER(volume_set_ea_speed, vol_upramp, val >= 0 && val <= .5)  // :324-333
VOL_FLOAT_SETTER(volume_set_ea_force, force, true)
VOL_FLOAT_SETTER(volume_set_ea_transmit, ea_transmit_thres, true)
VOL_FLOAT_SETTER(volume_set_ng_threshold, ng_threshold, true)
int volume_set_ea_sustain(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	d->p.sustain_time = *(int *)arg;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	volume_push_params(d);
	return 0;
}
int volume_set_agc(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	d->p.agc_enabled = *(int *)arg;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	volume_push_params(d);
	return 0;
}
int volume_enable_noise_gate(MSFilter *f, void *arg) { // :352-359
	VolumeData *d = (VolumeData *)f->data;
	d->p.noise_gate_enabled = *(bool_t *)arg;
	if (d->p.noise_gate_enabled) d->gain = d->target_gain = d->p.ng_floorgain;
	volume_set_gains(d, d->p.noise_gate_enabled != 0);
	return 0;
}
int volume_set_ng_floorgain(MSFilter *f, void *arg) { // :367-378
	VolumeData *d = (VolumeData *)f->data;
	d->p.ng_floorgain = *(float *)arg;
	if (d->p.ng_floorgain < 0.005f) d->p.ng_floorgain = 0.005f;
	if (d->p.noise_gate_enabled) d->gain = d->target_gain = d->p.ng_floorgain;
	volume_set_gains(d, d->p.noise_gate_enabled != 0);
	return 0;
}
int volume_remove_dc(MSFilter *f, void *arg) {
	VolumeData *d = (VolumeData *)f->data;
	d->p.remove_dc = *(int *)arg;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	volume_push_params(d);
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

// =================================================================== equalizer
struct EqualizerPool : Pool {
	int rate, cap_samples;
	mi_equalizer *e = nullptr;
	int16_t *h_buf, *d_buf;
	int32_t *h_n, *d_n;
	std::vector<int> staged, ready;
	EqualizerPool(int r) : rate(r) {
		init_slots(g_hub.capacity);
		MI_MUST(mi_equalizer_create(g_hub.context(), capacity, rate, &e));
		cap_samples = (std::max(960, rate / 100 * 2) + 7) & ~7;
		const size_t c = (size_t)capacity;
		h_buf = pinned<int16_t>(kMaxRounds * c * cap_samples);
		h_n = pinned<int32_t>(kMaxRounds * c);
		d_buf = devmem<int16_t>(c * cap_samples);
		d_n = devmem<int32_t>(c);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity;
		int maxr = 0;
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			for (int s = 0; s < capacity; ++s)
				if (staged[(size_t)s] <= r) h_n[r * c + s] = 0;
			MI_MUST(mi_copy_h2d(ctx, d_buf, h_buf + r * c * cap_samples, c * cap_samples * 2));
			MI_MUST(mi_copy_h2d(ctx, d_n, h_n + r * c, c * 4));
			MI_MUST(mi_equalizer_process_masked(e, d_buf, cap_samples, cap_samples, d_n));
			MI_MUST(mi_copy_d2h(ctx, h_buf + r * c * cap_samples, d_buf, c * cap_samples * 2));
		}
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
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
std::map<std::pair<MSTicker *, int>, EqualizerPool *> g_equalizer_pools;

struct EqualizerData {
	int rate;
	bool active;
	EqualizerPool *pool;
	int slot;
	std::vector<MSEqualizerGain> *pending; // gains since the last rate change, in call order
	MSBufferizer *spill;                   // the part of an over-long block that did not fit this tick's rounds
};

// Gains set before the filter is attached to a ticker are kept in `pending` and replayed, in
// order, when the slot is acquired (the reference keeps them in its own fft_cpx array).
void equalizer_attach(MSFilter *f) {
	EqualizerData *d = (EqualizerData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (d->pool && d->pool->rate == d->rate && d->pool->ticker == f->ticker) return;
	if (d->pool) d->pool->release(d->slot);
	d->pool = nullptr;
	d->slot = -1;
	if (!f->ticker) return;
	auto key = std::make_pair(f->ticker, d->rate);
	auto it = g_equalizer_pools.find(key);
	if (it == g_equalizer_pools.end()) {
		EqualizerPool *p = new EqualizerPool(d->rate);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_equalizer_pools.emplace(key, p).first;
	}
	d->pool = it->second;
	d->slot = d->pool->acquire(f);
	if (d->slot < 0) {
		d->pool = nullptr;
		return;
	}
	MI_MUST(mi_equalizer_flatten(d->pool->e, d->slot)); // equalizer_rate_update flattens
	MI_MUST(mi_equalizer_set_active(d->pool->e, d->slot, d->active));
	for (const MSEqualizerGain &g : *d->pending)
		MI_MUST(mi_equalizer_set_gain(d->pool->e, d->slot, g.frequency, g.gain, g.width));
}

void equalizer_init(MSFilter *f) { // equalizer.c:271-273: default rate 8000
	EqualizerData *d = (EqualizerData *)ms_malloc0(sizeof(*d));
	d->rate = 8000;
	d->active = true;
	d->slot = -1;
	d->pending = new std::vector<MSEqualizerGain>();
	d->spill = ms_bufferizer_new();
	f->data = d;
}
void equalizer_preprocess(MSFilter *f) { equalizer_attach(f); }
void equalizer_uninit(MSFilter *f) {
	EqualizerData *d = (EqualizerData *)f->data;
	if (d->pool) {
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		d->pool->release(d->slot);
	}
	delete d->pending;
	ms_bufferizer_destroy(d->spill);
	ms_free(d);
}
void equalizer_process(MSFilter *f) { // equalizer.c:279-288
	EqualizerData *d = (EqualizerData *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	mblk_t *m;
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
		if (p->staged[s] >= kMaxRounds) break;
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
	d->pending->push_back(*g);
	if (!d->pool) return 0;
	return mi_equalizer_set_gain(d->pool->e, d->slot, g->frequency, g->gain, g->width) == MI_OK ? 0 : -1;
}
int equalizer_get_gain(MSFilter *f, void *arg) { // equalizer.c:297-303 incl. its slot-indexing quirk
	EqualizerData *d = (EqualizerData *)f->data;
	MSEqualizerGain *g = (MSEqualizerGain *)arg;
	g->width = 0;
	g->gain = 0;
	if (!d->pool) return -1;
	const int nfft = mi_equalizer_fir_len(d->pool->e);
	std::vector<float> dump((size_t)nfft / 2);
	if (mi_equalizer_dump(d->pool->e, d->slot, dump.data(), nfft / 2) != MI_OK) return -1;
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
	d->rate = *(int *)arg;
	d->pending->clear(); // equalizer_rate_update re-allocates a flat spectrum
	if (d->pool && d->pool->rate == d->rate) MI_MUST(mi_equalizer_flatten(d->pool->e, d->slot));
	else equalizer_attach(f);
	return 0;
}
int equalizer_set_active(MSFilter *f, void *arg) { // equalizer.c:311-315: arg read as bool_t
	EqualizerData *d = (EqualizerData *)f->data;
	d->active = *(bool_t *)arg != 0;
	if (d->pool) MI_MUST(mi_equalizer_set_active(d->pool->e, d->slot, d->active));
	return 0;
}
int equalizer_dump(MSFilter *f, void *arg) {
	EqualizerData *d = (EqualizerData *)f->data;
	if (!d->pool) return -1;
	return mi_equalizer_dump(d->pool->e, d->slot, (float *)arg, mi_equalizer_fir_len(d->pool->e) / 2) == MI_OK ? 0 : -1;
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

// ======================================================================= mixer
constexpr int MIXER_MAX_CHANNELS = MI_MIXER_MAX_CHANNELS; // audiomixer.c:29
constexpr uint64_t BYPASS_MODE_TIMEOUT = 1000;            // audiomixer.c:31

struct MixerPool : Pool {
	int ns; // samples per tick (all channels interleaved)
	mi_mixer *m = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	uint8_t *h_has, *d_has, *h_run, *d_run, *h_mode, *d_mode;
	std::vector<uint8_t> flags;
	std::vector<float> gain;
	bool ctl_dirty = true;
	std::vector<uint8_t> staged, ready;
	MixerPool(int nsamples) : ns(nsamples) {
		init_slots(std::max(1, g_hub.capacity / 8));
		MI_MUST(mi_mixer_create(g_hub.context(), capacity, MIXER_MAX_CHANNELS, ns, &m));
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
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity, n = c * MIXER_MAX_CHANNELS;
		bool any = false;
		for (size_t s = 0; s < c; ++s) {
			h_run[s] = staged[s];
			any |= staged[s] != 0;
		}
		if (ctl_dirty) {
			MI_MUST(mi_mixer_set_controls(m, flags.data(), gain.data()));
			ctl_dirty = false;
		}
		if (any) {
			MI_MUST(mi_copy_h2d(ctx, d_in, h_in, n * ns * 2));
			MI_MUST(mi_copy_h2d(ctx, d_has, h_has, n));
			MI_MUST(mi_copy_h2d(ctx, d_run, h_run, c));
			MI_MUST(mi_copy_h2d(ctx, d_mode, h_mode, c));
			MI_MUST(mi_mixer_process_masked(m, d_in, d_has, 1, d_mode, d_out, d_run));
			MI_MUST(mi_copy_d2h(ctx, h_out, d_out, n * ns * 2));
			MI_MUST(mi_ctx_sync(ctx));
		}
		for (size_t s = 0; s < c; ++s) {
			ready[s] = staged[s];
			staged[s] = 0;
		}
	}
	void emit(MSFilter *f, int slot) override;
};
std::map<std::pair<MSTicker *, int>, MixerPool *> g_mixer_pools;

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
};

void mixer_init(MSFilter *f) { // audiomixer.c:145-156
	MixerState *s = (MixerState *)ms_malloc0(sizeof(*s));
	s->conf_mode = FALSE;
	s->nchannels = 1;
	s->rate = 44100;
	s->master_channel = -1;
	s->slot = -1;
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
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) ms_bufferizer_uninit(&s->channels[i].bufferizer);
	ms_free(s);
}
bool_t has_single_output(MSFilter *f, MixerState *s) { // audiomixer.c:167-176
	int count = 0;
	for (int i = 0; i < f->desc->noutputs; ++i)
		if (f->outputs[i] && s->channels[i].output_enabled) count++;
	return count == 1;
}
void mixer_push_controls(MSFilter *f, MixerState *s) {
	if (!s->pool) return;
	MixerPool *p = s->pool;
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
		uint8_t fl = 0;
		if (f->inputs[i]) fl |= MI_MIX_LINKED;
		if (s->channels[i].active) fl |= MI_MIX_ACTIVE;
		if (f->outputs[i] && s->channels[i].output_enabled) fl |= MI_MIX_OUTPUT;
		p->flags[(size_t)s->slot * MIXER_MAX_CHANNELS + i] = fl;
		p->gain[(size_t)s->slot * MIXER_MAX_CHANNELS + i] = s->channels[i].gain;
	}
	p->ctl_dirty = true;
}
void mixer_preprocess(MSFilter *f) { // audiomixer.c:178-200
	MixerState *s = (MixerState *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->bytespertick = (2 * s->nchannels * s->rate * f->ticker->interval) / 1000;
	for (int i = 0; i < MIXER_MAX_CHANNELS; ++i) {
		s->channels[i].last_flow_control = (uint64_t)-1;
		s->channels[i].last_activity = (uint64_t)-1;
	}
	s->skip_threshold = s->bytespertick * 2;
	s->bypass_mode = FALSE;
	s->single_output = has_single_output(f, s);
	const int ns = s->bytespertick / 2;
	auto key = std::make_pair(f->ticker, ns);
	auto it = g_mixer_pools.find(key);
	if (it == g_mixer_pools.end()) {
		MixerPool *p = new MixerPool(ns);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_mixer_pools.emplace(key, p).first;
	}
	s->pool = it->second;
	s->slot = s->pool->acquire(f);
	if (s->slot < 0) s->pool = nullptr;
	mixer_push_controls(f, s);
}
void mixer_postprocess(MSFilter *f) { // audiomixer.c:202-208 (slot released at every detach)
	MixerState *s = (MixerState *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (s->pool) {
		s->pool->release(s->slot);
		s->pool->staged[(size_t)s->slot] = s->pool->ready[(size_t)s->slot] = 0;
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
	const uint64_t now = f->ticker->time;
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
			for (mblk_t *m; (m = ms_queue_get(src)) != NULL;) ms_queue_put(dst, m);
			break;
		}
		for (mblk_t *m = peekq(&src->q); m != NULL && m != &src->q._q_stopper; m = m->b_next) ms_queue_put(dst, dupmsg(m));
	}
	ms_queue_flush(src);
}

// true = this tick is already dealt with (forwarded, or nobody contributes)
bool_t mixer_check_bypass(MSFilter *f, MixerState *s) {
	const Contributors c = mixer_census(f, s);
	if (c.count > 1) {
		if (s->bypass_mode) ms_message("MSAudioMixer [%p] is leaving bypass mode.", (void *)f);
		s->bypass_mode = FALSE;
		return FALSE;
	}
	if (c.count == 1) {
		if (!s->bypass_mode) ms_message("MSAudioMixer [%p] is entering bypass mode.", (void *)f);
		s->bypass_mode = TRUE;
		mixer_forward(f, s, c.pin);
	}
	return TRUE;
}

// ---- per-channel flow control (behaviour of audiomixer.c:92-111): every 5 s, if the bufferizer never dropped below
// `threshold` bytes in that window, discard the standing excess down to half the threshold.  Returns the bytes dropped.
int channel_flow_control(Channel *chan, int threshold, uint64_t now) {
	const bool first_call = chan->last_flow_control == (uint64_t)-1;
	int dropped = 0;
	if (!first_call) {
		const int level = (int)ms_bufferizer_get_avail(&chan->bufferizer);
		if (chan->min_fullness == -1 || level < chan->min_fullness) chan->min_fullness = level;
		if (now - chan->last_flow_control < 5000) return 0;
		if (chan->min_fullness >= threshold) {
			dropped = chan->min_fullness - threshold / 2;
			ms_bufferizer_skip_bytes(&chan->bufferizer, dropped);
		}
	}
	chan->last_flow_control = now; // a new observation window starts
	chan->min_fullness = -1;
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
	ms_filter_lock(f);
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (!s->pool) {
		ms_filter_unlock(f);
		return;
	}
	if (mixer_check_bypass(f, s)) {
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
		const int skip = channel_flow_control(chan, s->skip_threshold, f->ticker->time);
		if (skip > 0)
			ms_warning("Too much data in channel %i, %i ms in excess dropped", i, (skip * 1000) / (2 * s->nchannels * s->rate));
	}
	p->h_mode[(size_t)s->slot] = (uint8_t)(s->conf_mode != 0);
	p->staged[(size_t)s->slot] = 1; // ALWAYS_STREAMOUT :315-317
	request_flush(f);
	ms_filter_unlock(f);
}

int mixer_set_rate(MSFilter *f, void *data) {
	((MixerState *)f->data)->rate = *(int *)data;
	return 0;
}
int mixer_get_rate(MSFilter *f, void *data) {
	*(int *)data = ((MixerState *)f->data)->rate;
	return 0;
}
int mixer_set_nchannels(MSFilter *f, void *data) {
	((MixerState *)f->data)->nchannels = *(int *)data;
	return 0;
}
int mixer_get_nchannels(MSFilter *f, void *data) {
	*(int *)data = ((MixerState *)f->data)->nchannels;
	return 0;
}
bool mixer_pin_ok(const char *who, int pin) {
	if (pin < 0 || pin >= MIXER_MAX_CHANNELS) {
		ms_warning("%s: invalid pin number %i", who, pin);
		return false;
	}
	return true;
}
int mixer_set_input_gain(MSFilter *f, void *data) { // audiomixer.c:372-382
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_set_input_gain", ctl->pin)) return -1;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->channels[ctl->pin].gain = ctl->param.gain;
	mixer_push_controls(f, s);
	return 0;
}
int mixer_set_active(MSFilter *f, void *data) { // :384-393
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_set_active_gain", ctl->pin)) return -1;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->channels[ctl->pin].active = (bool_t)ctl->param.active;
	mixer_push_controls(f, s);
	return 0;
}
int mixer_enable_output(MSFilter *f, void *data) { // :395-408
	MixerState *s = (MixerState *)f->data;
	MSAudioMixerCtl *ctl = (MSAudioMixerCtl *)data;
	if (!mixer_pin_ok("mixer_enable_output", ctl->pin)) return -1;
	ms_filter_lock(f);
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->channels[ctl->pin].output_enabled = (bool_t)ctl->param.enabled;
	s->single_output = has_single_output(f, s);
	mixer_push_controls(f, s);
	ms_filter_unlock(f);
	return 0;
}
int mixer_set_conference_mode(MSFilter *f, void *data) {
	((MixerState *)f->data)->conf_mode = *(int *)data;
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

// ============================================================== echo canceller
struct EcPool : Pool {
	int rate, F, flen;
	mi_aec *a = nullptr;
	int16_t *h_mic, *h_ref, *h_out, *d_mic, *d_ref, *d_out;
	uint8_t *h_run, *d_run;
	std::vector<int> staged, ready;
	EcPool(int r, int frame, int filter_length) : rate(r), F(frame), flen(filter_length) {
		init_slots(g_hub.capacity);
		MI_MUST(mi_aec_create(g_hub.context(), capacity, rate, F, flen, &a));
		const size_t c = (size_t)capacity;
		h_mic = pinned<int16_t>(kMaxRounds * c * F);
		h_ref = pinned<int16_t>(kMaxRounds * c * F);
		h_out = pinned<int16_t>(kMaxRounds * c * F);
		h_run = pinned<uint8_t>(kMaxRounds * c);
		d_mic = devmem<int16_t>(c * F);
		d_ref = devmem<int16_t>(c * F);
		d_out = devmem<int16_t>(c * F);
		d_run = devmem<uint8_t>(c);
		staged.assign(c, 0);
		ready.assign(c, 0);
	}
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity;
		int maxr = 0;
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			for (int s = 0; s < capacity; ++s) h_run[r * c + s] = staged[(size_t)s] > r;
			MI_MUST(mi_copy_h2d(ctx, d_mic, h_mic + r * c * F, c * F * 2));
			MI_MUST(mi_copy_h2d(ctx, d_ref, h_ref + r * c * F, c * F * 2));
			MI_MUST(mi_copy_h2d(ctx, d_run, h_run + r * c, c));
			MI_MUST(mi_aec_process(a, d_mic, d_ref, d_out, F, d_run, MI_AEC_POSTFILTER));
			MI_MUST(mi_copy_d2h(ctx, h_out + r * c * F, d_out, c * F * 2));
		}
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
		}
	}
	void emit(MSFilter *f, int slot) override {
		const size_t c = (size_t)capacity, sl = (size_t)slot;
		for (int r = 0; r < ready[sl]; ++r) { // cleaned frames -> outputs[1] (speexec.c:303)
			mblk_t *oecho = allocb((size_t)F * 2, 0);
			memcpy(oecho->b_wptr, h_out + (r * c + sl) * F, (size_t)F * 2);
			oecho->b_wptr += F * 2;
			if (f->outputs[1]) ms_queue_put(f->outputs[1], oecho);
			else freemsg(oecho);
		}
		ready[sl] = 0;
	}
};
std::map<std::tuple<MSTicker *, int, int, int>, EcPool *> g_ec_pools;

// MSFlowControlledBufferizer, src/base/msqueue.c:127-256 (SendEvent drop method)
struct FlowBuf {
	MSBufferizer base;
	MSFilter *filter;
	uint64_t flow_control_time;
	uint32_t interval_ms, max_size_ms, granularity_ms, min_size_ms_during_interval;
	int samplerate, nchannels;
	bool immediate_drop; // MSFlowControlledBufferizerImmediateDrop instead of SendEvent (msqueue.c:213-218)
};
void flowbuf_init(FlowBuf *o, MSFilter *f, int rate) {
	ms_bufferizer_init(&o->base);
	o->filter = f;
	o->interval_ms = 5000;
	o->max_size_ms = 100;
	o->granularity_ms = 0;
	o->flow_control_time = 0;
	o->min_size_ms_during_interval = UINT32_MAX;
	o->samplerate = rate;
	o->nchannels = 1;
	o->immediate_drop = false;
}
void flowbuf_put(FlowBuf *o, mblk_t *m, MSQueue *q = nullptr) { // msqueue.c:193-256 (m, or everything queued on q)
	const uint32_t accumulated_ms = (uint32_t)((o->base.size * 1000) / (size_t)o->samplerate / 2) / (uint32_t)o->nchannels;
	if (accumulated_ms < o->min_size_ms_during_interval) o->min_size_ms_during_interval = accumulated_ms;
	if (q) ms_bufferizer_put_from_queue(&o->base, q);
	else ms_bufferizer_put(&o->base, m);
	const uint64_t now = o->filter->ticker->time;
	const uint32_t since = (uint32_t)(now - o->flow_control_time);
	if (o->flow_control_time == 0) o->flow_control_time = now;
	if (since >= o->interval_ms) {
		uint32_t diff_ms = 0;
		bool trig = false;
		if (o->min_size_ms_during_interval != UINT32_MAX && o->min_size_ms_during_interval > o->max_size_ms) {
			diff_ms = o->min_size_ms_during_interval - o->max_size_ms;
			trig = true;
		} else if (accumulated_ms > o->max_size_ms * 4) {
			diff_ms = (accumulated_ms - o->max_size_ms) / 2;
			trig = true;
		}
		if (trig && diff_ms > o->granularity_ms / 2) {
			MSAudioFlowControlDropEvent ev;
			ev.flow_control_interval_ms = o->interval_ms;
			ev.drop_ms = diff_ms - o->granularity_ms / 2;
			if (ev.drop_ms > 0) {
				if (o->immediate_drop) ms_bufferizer_skip_bytes(&o->base, (int)((ev.drop_ms * 2 * (uint32_t)o->nchannels * (uint32_t)o->samplerate) / 1000));
				else ms_filter_notify(o->filter, MS_AUDIO_FLOW_CONTROL_DROP_EVENT, &ev);
			}
		}
		o->flow_control_time = now;
		o->min_size_ms_during_interval = UINT32_MAX;
	}
}

struct SpeexECState { // speexec.c:49-72
	MSBufferizer delayed_ref;
	FlowBuf ref;
	MSBufferizer echo;
	int framesize, framesize_at_8000, filterlength, samplerate, delay_ms, tail_length_ms, nominal_ref_samples;
	char *state_str;
	bool_t echostarted, bypass_mode, using_zeroes;
	EcPool *pool;
	int slot;
};

void ec_init(MSFilter *f) { // speexec.c:74-109
	SpeexECState *s = (SpeexECState *)ms_malloc0(sizeof(*s));
	s->samplerate = 8000;
	ms_bufferizer_init(&s->delayed_ref);
	ms_bufferizer_init(&s->echo);
	flowbuf_init(&s->ref, f, s->samplerate);
	s->tail_length_ms = 250;
	s->framesize_at_8000 = 64;
	s->slot = -1;
	f->data = s;
}
void ec_uninit(MSFilter *f) {
	SpeexECState *s = (SpeexECState *)f->data;
	if (s->state_str) ms_free(s->state_str);
	ms_bufferizer_uninit(&s->delayed_ref);
	ms_free(s);
}
void ec_configure_flow(SpeexECState *s) { // speexec.c:182-186
	s->ref.samplerate = s->samplerate;
	s->ref.max_size_ms = (uint32_t)s->delay_ms;
	s->ref.granularity_ms = (uint32_t)((s->framesize * 1000) / s->samplerate);
}
// ---- the canceller's state as a string: fetch_config / apply_config, speexec.c:119-167 (there a SpeexEchoStateBlob of the
// speex fork through bctbx_base64_*; here the blob of mi_aec_export_state through a local RFC 4648 codec)
const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
char *b64_encode(const uint8_t *p, size_t n) {
	char *out = (char *)ms_malloc0(4 * ((n + 2) / 3) + 1), *o = out;
	for (size_t i = 0; i < n; i += 3) {
		const uint32_t v = ((uint32_t)p[i] << 16) | ((i + 1 < n ? (uint32_t)p[i + 1] : 0u) << 8) | (i + 2 < n ? (uint32_t)p[i + 2] : 0u);
		*o++ = kB64[(v >> 18) & 63];
		*o++ = kB64[(v >> 12) & 63];
		*o++ = i + 1 < n ? kB64[(v >> 6) & 63] : '=';
		*o++ = i + 2 < n ? kB64[v & 63] : '=';
	}
	*o = 0;
	return out;
}
bool b64_decode(const char *txt, std::vector<uint8_t> &out) {
	int8_t rev[256];
	memset(rev, -1, sizeof(rev));
	for (int i = 0; i < 64; ++i) rev[(uint8_t)kB64[i]] = (int8_t)i;
	out.clear();
	uint32_t acc = 0;
	int bits = 0;
	for (const char *c = txt; *c && *c != '='; ++c) {
		if (*c == '\n' || *c == '\r' || *c == ' ') continue;
		const int v = rev[(uint8_t)*c];
		if (v < 0) return false;
		acc = (acc << 6) | (uint32_t)v;
		bits += 6;
		if (bits >= 8) {
			bits -= 8;
			out.push_back((uint8_t)(acc >> bits));
		}
	}
	return true;
}
struct SpeexECState;
void ec_apply_config(SpeexECState *s);
void ec_fetch_config(SpeexECState *s);

void ec_preprocess(MSFilter *f) { // speexec.c:188-216
	SpeexECState *s = (SpeexECState *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->echostarted = FALSE;
	s->filterlength = (s->tail_length_ms * s->samplerate) / 1000;
	s->framesize = mi_aec_framesize(s->framesize_at_8000, s->samplerate);
	if (s->framesize != 64 && s->framesize != 128 && s->framesize != 256) {
		// e.g. 96 kHz would need 512-sample frames: audio keeps flowing uncancelled rather than the process dying
		ms_error("mi355x echo canceller: frame size %d (rate %d) is not built; the filter forwards both pins untouched",
		         s->framesize, s->samplerate);
		s->bypass_mode = TRUE;
		return;
	}
	if (s->filterlength > 64 * s->framesize) { // the kernels hold at most 64 filter blocks (341 ms at 48 kHz, 512 ms at 8/16 kHz)
		ms_warning("mi355x echo canceller: tail of %d ms shortened to %d ms (64 blocks of %d samples)", s->tail_length_ms,
		           64 * s->framesize * 1000 / s->samplerate, s->framesize);
		s->filterlength = 64 * s->framesize;
	}
	const int delay_samples = s->delay_ms * s->samplerate / 1000;
	ms_message("Initializing mi355x echo canceler with framesize=%i, filterlength=%i, delay_samples=%i", s->framesize,
	           s->filterlength, delay_samples);
	auto key = std::make_tuple(f->ticker, s->samplerate, s->framesize, s->filterlength);
	auto it = g_ec_pools.find(key);
	if (it == g_ec_pools.end()) {
		EcPool *p = new EcPool(s->samplerate, s->framesize, s->filterlength);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_ec_pools.emplace(key, p).first;
	}
	s->pool = it->second;
	s->slot = s->pool->acquire(f);
	if (s->slot < 0) s->pool = nullptr;
	mblk_t *m = allocb((size_t)delay_samples * 2, 0); // zeroes for the time of the delay
	memset(m->b_wptr, 0, (size_t)delay_samples * 2);
	m->b_wptr += delay_samples * 2;
	ms_bufferizer_put(&s->delayed_ref, m);
	s->nominal_ref_samples = delay_samples;
	ec_apply_config(s); // :209-211
}
void ec_apply_config(SpeexECState *s) { // :121-143
	if (s->state_str == NULL || s->pool == nullptr) return;
	std::vector<uint8_t> blob;
	if (!b64_decode(s->state_str, blob)) {
		ms_error("Could not decode base64 %.32s...", s->state_str);
		return;
	}
	if (mi_aec_import_state(s->pool->a, s->slot, blob.data(), blob.size()) != MI_OK) {
		ms_error("Could not apply mi355x echo blob: %s", mi_last_error()); // e.g. saved at another rate or tail length
		return;
	}
	ms_message("mi355x echo state restored.");
}
void ec_fetch_config(SpeexECState *s) { // :145-167
	if (s->pool == nullptr) return;
	std::vector<uint8_t> blob(mi_aec_blob_bytes(s->pool->a));
	if (mi_aec_export_state(s->pool->a, s->slot, blob.data(), blob.size()) != MI_OK) {
		ms_error("Could not retrieve mi355x echo blob: %s", mi_last_error());
		return;
	}
	if (s->state_str) ms_free(s->state_str);
	s->state_str = b64_encode(blob.data(), blob.size());
}
void ec_postprocess(MSFilter *f) { // speexec.c:307-321: state destroyed at detach
	SpeexECState *s = (SpeexECState *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	ms_bufferizer_flush(&s->delayed_ref);
	ms_bufferizer_flush(&s->echo);
	ms_bufferizer_flush(&s->ref.base);
	if (s->pool) {
		MI_MUST(mi_aec_reset(s->pool->a, s->slot, 1));
		s->pool->release(s->slot);
		s->pool->staged[(size_t)s->slot] = s->pool->ready[(size_t)s->slot] = 0;
	}
	s->pool = nullptr;
	s->slot = -1;
}

// inputs[0] far-end reference, inputs[1] mic; outputs[0] reference copy, outputs[1] cleaned mic (speexec.c:218-222)
// ---- the framing of speexec.c:223-305, in three steps ---------------------------------------------------------------
// (1) far-end blocks: kept twice once the microphone has started -- in `delayed_ref` (what the canceller will be fed,
//     behind the configured delay) and in the flow-controlled `ref` (what goes on to the speaker, frame by frame).
void ec_take_far_end(MSFilter *f, SpeexECState *s) {
	if (!f->inputs[0]) return;
	if (!s->echostarted) {
		ms_warning("Getting reference signal but no echo to synchronize on.");
		ms_queue_flush(f->inputs[0]);
		return;
	}
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) {
		ms_bufferizer_put(&s->delayed_ref, dupmsg(m));
		flowbuf_put(&s->ref, m);
	}
}

mblk_t *ec_block(size_t nbytes) {
	mblk_t *m = allocb(nbytes, 0);
	memset(m->b_wptr, 0, nbytes);
	m->b_wptr += nbytes;
	return m;
}

// (2) one speaker frame per microphone frame: from `ref` when the delay line holds more than the nominal delay plus a
//     frame, otherwise a frame of silence that is ALSO appended to the delay line (the canceller then sees zeros too).
void ec_emit_speaker_frame(MSFilter *f, SpeexECState *s, size_t nbytes) {
	const size_t needed = (size_t)s->nominal_ref_samples * 2 + nbytes;
	if (ms_bufferizer_get_avail(&s->delayed_ref) < needed) {
		mblk_t *silence = ec_block(nbytes);
		ms_bufferizer_put(&s->delayed_ref, silence);
		ms_queue_put(f->outputs[0], dupmsg(silence));
		if (!s->using_zeroes) ms_warning("Not enough ref samples, using zeroes");
		s->using_zeroes = TRUE;
		return;
	}
	if (s->using_zeroes) ms_message("Samples are back.");
	s->using_zeroes = FALSE;
	mblk_t *m = ec_block(nbytes);
	if (ms_bufferizer_read(&s->ref.base, m->b_rptr, nbytes) == 0) {
		ms_error("Should never happen");
		abort();
	}
	ms_queue_put(f->outputs[0], m);
}

// (3) every complete microphone frame is staged with its reference frame; the batch cancels them at the next flush
void ec_process(MSFilter *f) {
	SpeexECState *s = (SpeexECState *)f->data;
	if (s->bypass_mode) { // both pins straight through
		for (int pin = 0; pin < 2; ++pin)
			for (mblk_t *m; (m = ms_queue_get(f->inputs[pin])) != NULL;) ms_queue_put(f->outputs[pin], m);
		return;
	}
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	EcPool *p = s->pool;
	if (!p) {
		ms_queue_flush(f->inputs[0]);
		ms_queue_flush(f->inputs[1]);
		return;
	}
	const size_t nbytes = (size_t)s->framesize * 2, cap = (size_t)p->capacity, slot = (size_t)s->slot;
	ec_take_far_end(f, s);
	ms_bufferizer_put_from_queue(&s->echo, f->inputs[1]);
	while (p->staged[slot] < kMaxRounds && ms_bufferizer_get_avail(&s->echo) >= nbytes) {
		const size_t row = ((size_t)p->staged[slot] * cap + slot) * (size_t)p->F;
		ms_bufferizer_read(&s->echo, (uint8_t *)(p->h_mic + row), nbytes);
		s->echostarted = TRUE;
		ec_emit_speaker_frame(f, s, nbytes);
		if (ms_bufferizer_read(&s->delayed_ref, (uint8_t *)(p->h_ref + row), nbytes) == 0) {
			ms_error("Should never happen");
			abort();
		}
		p->staged[slot]++;
	}
	if (p->staged[slot]) request_flush(f);
}

int ec_set_sr(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->samplerate = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_framesize(MSFilter *f, void *arg) {
	((SpeexECState *)f->data)->framesize_at_8000 = *(int *)arg;
	return 0;
}
int ec_set_delay(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->delay_ms = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_tail_length(MSFilter *f, void *arg) {
	SpeexECState *s = (SpeexECState *)f->data;
	s->tail_length_ms = *(int *)arg;
	ec_configure_flow(s);
	return 0;
}
int ec_set_bypass_mode(MSFilter *f, void *arg) {
	((SpeexECState *)f->data)->bypass_mode = *(bool_t *)arg;
	return 0;
}
int ec_get_bypass_mode(MSFilter *f, void *arg) {
	*(bool_t *)arg = ((SpeexECState *)f->data)->bypass_mode;
	return 0;
}
int ec_set_state(MSFilter *f, void *arg) { // :361-365 (the previous string leaks there; freed here)
	SpeexECState *s = (SpeexECState *)f->data;
	const size_t n = strlen((const char *)arg) + 1;
	if (s->state_str) ms_free(s->state_str);
	s->state_str = (char *)ms_malloc0(n);
	memcpy(s->state_str, arg, n);
	return 0;
}
int ec_get_state(MSFilter *f, void *arg) { // :367-374: the CURRENT state while attached, the stored string otherwise
	SpeexECState *s = (SpeexECState *)f->data;
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		ec_fetch_config(s);
	}
	*(char **)arg = s->state_str;
	return 0;
}
int ec_get_delay(MSFilter *f, void *arg) {
	*(int *)arg = ((SpeexECState *)f->data)->delay_ms;
	return 0;
}
MSFilterMethod ec_methods[] = {{MS_FILTER_SET_SAMPLE_RATE, ec_set_sr},
                               {MS_ECHO_CANCELLER_SET_TAIL_LENGTH, ec_set_tail_length},
                               {MS_ECHO_CANCELLER_SET_DELAY, ec_set_delay},
                               {MS_ECHO_CANCELLER_SET_FRAMESIZE, ec_set_framesize},
                               {MS_ECHO_CANCELLER_SET_BYPASS_MODE, ec_set_bypass_mode},
                               {MS_ECHO_CANCELLER_GET_BYPASS_MODE, ec_get_bypass_mode},
                               {MS_ECHO_CANCELLER_GET_STATE_STRING, ec_get_state},
                               {MS_ECHO_CANCELLER_SET_STATE_STRING, ec_set_state},
                               {MS_ECHO_CANCELLER_GET_DELAY, ec_get_delay},
                               {0, NULL}};

// ====================================================================== video
// ---- MSScalerDesc (msvideo.h:473-478): the reference's synchronous, one-frame interface -------------
// Dispatch on the SOURCE format like yuv_scale (src/voip/msvideo.c:542-581): I420 is scaled (dst RGB24
// is honoured like the swscale implementation :672-681 does; libyuv's ignores it), packed formats are
// converted to I420 at the same size.
int pix_to_mi(MSPixFmt f) {
	switch (f) {
		case MS_YUY2:
		case MS_YUYV: return MI_PIX_YUY2;
		case MS_UYVY: return MI_PIX_UYVY;
		case MS_RGB24: return MI_PIX_BGR24;
		case MS_RGB24_REV: return MI_PIX_RGB24_RAW;
		case MS_RGBA32_REV: return MI_PIX_BGRA32;
		default: return -1;
	}
}
int pix_bpp(MSPixFmt f) {
	switch (f) {
		case MS_YUY2:
		case MS_YUYV:
		case MS_UYVY: return 2;
		case MS_RGB24:
		case MS_RGB24_REV: return 3;
		default: return 4;
	}
}

struct ScalerCtx { // what MSScalerContext* points to
	int sw, sh, dw, dh;
	MSPixFmt sf, df;
	mi_scaler *sc = nullptr;
	mi_pixconv *pc[2] = {nullptr, nullptr}; // [flip]
	std::vector<uint8_t> packed_in, packed_out;
};

MSScalerContext *sd_create(int sw, int sh, MSPixFmt sf, int dw, int dh, MSPixFmt df, int flags) {
	(void)flags; // bilinear either way, like yuv_create_scale_context msvideo.c:526-540
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	ScalerCtx *c = new ScalerCtx();
	c->sw = sw, c->sh = sh, c->dw = dw, c->dh = dh, c->sf = sf, c->df = df;
	if (sf == MS_YUV420P) {
		const int fmt = (df == MS_RGB24) ? MI_PIX_RGB24 : MI_PIX_I420;
		if ((df != MS_RGB24 && df != MS_YUV420P) || mi_scaler_create(g_hub.context(), sw, sh, dw, dh, fmt, &c->sc) != MI_OK) {
			ms_error("msmi355x scaler: %dx%d fmt %d -> %dx%d fmt %d unsupported: %s", sw, sh, (int)sf, dw, dh, (int)df, mi_last_error());
			delete c;
			return NULL;
		}
	} else if (pix_to_mi(sf) < 0 || sw != dw || sh != dh) {
		ms_warning("msmi355x scaler: unsupported format %d or size change on a packed source", (int)sf); // msvideo.c:574-576
		delete c;
		return NULL;
	}
	return (MSScalerContext *)c;
}

int sd_process(MSScalerContext *ctx, uint8_t *src[], int src_strides[], uint8_t *dst[], int dst_strides[]) {
	ScalerCtx *c = (ScalerCtx *)ctx;
	if (!c) return -1;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (c->sc) {
		const uint8_t *sp[3] = {src[0], src[1], src[2]};
		uint8_t *dp[3] = {dst[0], dst[1], dst[2]};
		return mi_scaler_process_planes_host(c->sc, sp, src_strides, dp, dst_strides) == MI_OK ? 0 : -1;
	}
	// packed -> I420.  A negative stride means the caller walks the bitmap bottom-up (pixconv.c:78-81).
	const int bpp = pix_bpp(c->sf), rowb = c->sw * bpp;
	const int flip = src_strides[0] < 0;
	const int stride = flip ? -src_strides[0] : src_strides[0];
	if (stride < rowb) return -1;
	if (!c->pc[flip] && mi_pixconv_create(g_hub.context(), c->sw, c->sh, pix_to_mi(c->sf), flip, &c->pc[flip]) != MI_OK) {
		ms_error("msmi355x scaler: %s", mi_last_error());
		return -1;
	}
	const uint8_t *lowest = flip ? src[0] - (size_t)stride * (c->sh - 1) : src[0];
	const uint8_t *in = lowest;
	if (stride != rowb) { // pack the rows
		c->packed_in.resize((size_t)rowb * c->sh);
		for (int y = 0; y < c->sh; ++y) memcpy(c->packed_in.data() + (size_t)y * rowb, lowest + (size_t)y * stride, (size_t)rowb);
		in = c->packed_in.data();
	}
	const size_t ob = mi_pixconv_dst_bytes(c->pc[flip]);
	c->packed_out.resize(ob);
	if (mi_pixconv_process_host(c->pc[flip], 1, in, mi_pixconv_src_bytes(c->pc[flip]), c->packed_out.data(), ob) != MI_OK) {
		ms_error("msmi355x scaler: %s", mi_last_error());
		return -1;
	}
	const int w = c->sw, h = c->sh, h2 = h + (h & 1), cw = w / 2, ch = (h + 1) / 2;
	const uint8_t *o = c->packed_out.data();
	for (int y = 0; y < h; ++y) memcpy(dst[0] + (size_t)y * dst_strides[0], o + (size_t)y * w, (size_t)w);
	for (int y = 0; y < ch; ++y) {
		memcpy(dst[1] + (size_t)y * dst_strides[1], o + (size_t)w * h2 + (size_t)y * cw, (size_t)cw);
		memcpy(dst[2] + (size_t)y * dst_strides[2], o + (size_t)w * h2 + (size_t)cw * (h2 / 2) + (size_t)y * cw, (size_t)cw);
	}
	return 0;
}

void sd_free(MSScalerContext *ctx) {
	ScalerCtx *c = (ScalerCtx *)ctx;
	if (!c) return;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (c->sc) mi_scaler_destroy(c->sc);
	for (int i = 0; i < 2; ++i)
		if (c->pc[i]) mi_pixconv_destroy(c->pc[i]);
	delete c;
}

// ---- frame pools: every MSSizeConv (or MSPixConv) of one geometry on one ticker shares a batch --------
struct FramePool : Pool {
	struct Staged {
		MSFilter *f;
		uint32_t ts;
	};
	size_t src_bytes = 0, dst_bytes = 0, src_pitch = 0, dst_pitch = 0;
	int out_w = 0, out_h = 0;
	int frame_cap = 0; // frames per tick the staging holds (filters that may attach: `capacity`)
	uint8_t *h_src = nullptr, *h_dst = nullptr, *d_src = nullptr, *d_dst = nullptr;
	std::vector<Staged> staged, ready;
	virtual int launch(int nframes) = 0;
	void alloc_buffers() {
		src_pitch = (src_bytes + 31) & ~(size_t)15; // slack for the kernels' 16-byte row loads
		dst_pitch = (dst_bytes + 15) & ~(size_t)15;
		const size_t c = (size_t)frame_cap;
		h_src = pinned<uint8_t>(c * src_pitch);
		h_dst = pinned<uint8_t>(c * dst_pitch);
		d_src = devmem<uint8_t>(c * src_pitch + 32);
		d_dst = devmem<uint8_t>(c * dst_pitch + 32);
	}
	// next staging buffer, or NULL when `capacity` frames are already waiting for this tick's flush
	uint8_t *stage(MSFilter *f, uint32_t ts) {
		if ((int)staged.size() >= frame_cap) {
			ms_error("msmi355x plugin: frame pool full (%d frames per tick; raise MSMI355X_FRAME_SLOTS)", frame_cap);
			return nullptr;
		}
		staged.push_back({f, ts});
		return h_src + (staged.size() - 1) * src_pitch;
	}
	void flush() override {
		ready.clear();
		const int n = (int)staged.size();
		if (!n) return;
		mi_ctx *ctx = g_hub.context();
		MI_MUST(mi_copy_h2d(ctx, d_src, h_src, (size_t)n * src_pitch));
		MI_MUST(launch(n));
		MI_MUST(mi_copy_d2h(ctx, h_dst, d_dst, (size_t)n * dst_pitch));
		MI_MUST(mi_ctx_sync(ctx));
		ready.swap(staged);
	}
	void emit(MSFilter *f, int slot) override;
	void forget(MSFilter *f) { // the filter left the pool: its frames in flight are dropped
		for (Staged &s : staged)
			if (s.f == f) s.f = nullptr;
		for (Staged &s : ready)
			if (s.f == f) s.f = nullptr;
	}
};

struct VideoOut { // what a frame-pool client exposes for result delivery
	MSYuvBufAllocator *allocator;
};

void FramePool::emit(MSFilter *f, int slot) {
	(void)slot;
	for (size_t k = 0; k < ready.size(); ++k) {
		if (ready[k].f != f) continue;
		ready[k].f = nullptr;
		VideoOut *vo = (VideoOut *)f->data; // first member of both filter states
		YuvBuf ob;
		mblk_t *om = ms_yuv_buf_allocator_get(vo->allocator, &ob, out_w, out_h);
		if (om == NULL) continue;
		// device layout == ms_yuv_buf_init layout (stride w, contiguous planes)
		memcpy(ob.planes[0], h_dst + k * dst_pitch, dst_bytes);
		mblk_set_timestamp_info(om, ready[k].ts);
		if (f->outputs[0]) ms_queue_put(f->outputs[0], om);
		else freemsg(om);
	}
}

// frames per geometry and tick (MSMI355X_FRAME_SLOTS, default 32): a 1080p row is 3 MB of pinned memory
int frame_slots() {
	const char *e = getenv("MSMI355X_FRAME_SLOTS");
	const int v = e ? atoi(e) : 0;
	return v > 0 ? v : 32;
}

struct ScalerPool : FramePool {
	mi_scaler *sc = nullptr;
	ScalerPool(mi_scaler *created, int dw, int dh) : sc(created) {
		init_slots(g_hub.capacity);
		frame_cap = frame_slots();
		src_bytes = mi_scaler_src_bytes(sc);
		dst_bytes = mi_scaler_dst_bytes(sc);
		out_w = dw, out_h = dh;
		alloc_buffers();
	}
	int launch(int n) override { return mi_scaler_process(sc, n, d_src, src_pitch, d_dst, dst_pitch); }
};
std::map<std::tuple<MSTicker *, int, int, int, int>, ScalerPool *> g_scaler_pools;

struct PixPool : FramePool {
	mi_pixconv *pc = nullptr;
	PixPool(mi_pixconv *created, int w, int h) : pc(created) {
		init_slots(g_hub.capacity);
		frame_cap = frame_slots();
		src_bytes = mi_pixconv_src_bytes(pc);
		dst_bytes = mi_pixconv_dst_bytes(pc);
		out_w = w, out_h = h;
		alloc_buffers();
	}
	int launch(int n) override { return mi_pixconv_process(pc, n, d_src, src_pitch, d_dst, dst_pitch); }
};
std::map<std::tuple<MSTicker *, int, int, int>, PixPool *> g_pix_pools;

// ---- MSSizeConv (src/videofilters/sizeconv.c) ----------------------------------------------------------
struct SizeConvState { // SizeConvState sizeconv.c:29-40
	MSYuvBufAllocator *allocator; // first: VideoOut
	MSVideoSize target_vsize;
	MSVideoSize in_vsize;
	float fps;
	float start_time;
	int frame_count;
	queue_t rq;
	bool_t needRefresh;
	ScalerPool *pool;
	int slot;
};

void size_conv_leave_pool(SizeConvState *s, MSFilter *f) {
	if (s->pool) {
		s->pool->forget(f);
		s->pool->release(s->slot);
	}
	s->pool = nullptr;
	s->slot = -1;
}

void size_conv_init(MSFilter *f) { // sizeconv.c:46-60
	SizeConvState *s = (SizeConvState *)ms_malloc0(sizeof(SizeConvState));
	s->target_vsize.width = MS_VIDEO_SIZE_CIF_W;
	s->target_vsize.height = MS_VIDEO_SIZE_CIF_H;
	s->allocator = ms_yuv_buf_allocator_new();
	s->start_time = 0;
	s->frame_count = -1;
	s->needRefresh = FALSE;
	s->fps = -1; /* default to process ALL frames */
	s->slot = -1;
	qinit(&s->rq);
	f->data = s;
}
void size_conv_uninit(MSFilter *f) { // :62-66
	SizeConvState *s = (SizeConvState *)f->data;
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		size_conv_leave_pool(s, f);
	}
	ms_yuv_buf_allocator_free(s->allocator);
	ms_free(s);
}
void size_conv_postprocess(MSFilter *f) { // :68-76 (the scaler context there == our pool membership)
	SizeConvState *s = (SizeConvState *)f->data;
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		size_conv_leave_pool(s, f);
	}
	flushq(&s->rq, 0);
	s->frame_count = -1;
}

// get_resampler sizeconv.c:82-95: (re)join the pool of this geometry
ScalerPool *size_conv_pool(MSFilter *f, SizeConvState *s, int w, int h) {
	if (s->pool && s->in_vsize.width == w && s->in_vsize.height == h && s->pool->ticker == f->ticker &&
	    s->pool->out_w == s->target_vsize.width && s->pool->out_h == s->target_vsize.height)
		return s->pool;
	size_conv_leave_pool(s, f);
	auto key = std::make_tuple(f->ticker, w, h, s->target_vsize.width, s->target_vsize.height);
	auto it = g_scaler_pools.find(key);
	if (it == g_scaler_pools.end()) {
		// a geometry the kernels cannot take is not fatal: the frame is dropped with an error, as a failing
		// ms_scaler_process is in the reference (sizeconv.c:162-166)
		mi_scaler *sc = nullptr;
		if (mi_scaler_create(g_hub.context(), w, h, s->target_vsize.width, s->target_vsize.height, MI_PIX_I420, &sc) != MI_OK) {
			ms_error("MSSizeConv: %dx%d -> %dx%d: %s", w, h, s->target_vsize.width, s->target_vsize.height, mi_last_error());
			return nullptr;
		}
		ScalerPool *p = new ScalerPool(sc, s->target_vsize.width, s->target_vsize.height);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_scaler_pools.emplace(key, p).first;
	}
	s->pool = it->second;
	s->slot = s->pool->acquire(f);
	if (s->slot < 0) s->pool = nullptr;
	s->in_vsize.width = w;
	s->in_vsize.height = h;
	ms_message("MSSizeConv: create new scaler context with w %d, h %d", w, h);
	return s->pool;
}

// -- the three decisions of sizeconv.c:97-184, one helper each ------------------------------------------------
// (1) frame-rate limiter, :107-132: which queued frames survive this tick.  Returns false when the tick must not
//     emit at all (the frame period has not elapsed); in both throttled cases only the newest frame is kept.
bool size_conv_rate_gate(MSFilter *f, SizeConvState *s) {
	if (s->frame_count == -1) { // first tick after a (re)start
		s->start_time = (float)f->ticker->time;
		s->frame_count = 0;
	}
	if (s->fps < 0) return true; // unlimited: every frame goes through
	const int due = (int)((f->ticker->time - s->start_time) * s->fps / 1000.0);
	while (s->rq.q_mcount > 1) { // older captures are dropped, the most recent one stays
		ms_message("MSSizeConv: extra frame removed.");
		freemsg(getq(&s->rq));
	}
	return due > s->frame_count;
}

// (2) geometry fix-up, :139-157: same orientation as the input, same aspect ratio.  Returns true when the
//     target had to change (the application is told and must re-negotiate before frames flow again).
bool size_conv_adapt_target(SizeConvState *s, int in_w, int in_h) {
	const MSVideoSize before = s->target_vsize, in_sz = {in_w, in_h};
	MSVideoSize &t = s->target_vsize;
	if (ms_video_size_get_orientation(in_sz) != ms_video_size_get_orientation(t)) std::swap(t.width, t.height);
	if (in_w * t.height / t.width != in_h) {
		if (in_w > in_h) t.height = in_h * t.width / in_w;
		else t.width = in_w * t.height / in_h;
	}
	return t.width != before.width || t.height != before.height;
}

// (3) hand one frame to the batch (the ms_scaler_process call of :161): planes gathered into the packed layout
bool size_conv_stage(MSFilter *f, SizeConvState *s, const YuvBuf &in, uint32_t ts) {
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	ScalerPool *p = size_conv_pool(f, s, in.w, in.h);
	uint8_t *dst = p ? p->stage(f, ts) : nullptr;
	if (!dst) return false;
	const int h2 = in.h + (in.h & 1), cw = in.w / 2, crows = (in.h + 1) / 2;
	uint8_t *du = dst + (size_t)in.w * h2, *dv = du + (size_t)cw * (h2 / 2);
	for (int y = 0; y < in.h; ++y) memcpy(dst + (size_t)y * in.w, in.planes[0] + (size_t)y * in.strides[0], (size_t)in.w);
	for (int y = 0; y < crows; ++y) {
		memcpy(du + (size_t)y * cw, in.planes[1] + (size_t)y * in.strides[1], (size_t)cw);
		memcpy(dv + (size_t)y * cw, in.planes[2] + (size_t)y * in.strides[2], (size_t)cw);
	}
	return true;
}

void size_conv_process(MSFilter *f) { // sizeconv.c:97-184
	SizeConvState *s = (SizeConvState *)f->data;
	bool staged = false;
	ms_filter_lock(f);
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) putq(&s->rq, m);
	if (!size_conv_rate_gate(f, s)) {
		ms_filter_unlock(f);
		return;
	}
	for (mblk_t *im; (im = getq(&s->rq)) != NULL;) {
		YuvBuf in;
		if (ms_yuv_buf_init_from_mblk(&in, im) != 0) {
			ms_warning("size_conv_process(): bad buffer.");
			freemsg(im);
			continue;
		}
		s->frame_count++;
		if (in.w == s->target_vsize.width && in.h == s->target_vsize.height) {
			ms_queue_put(f->outputs[0], im); // already the right size: forwarded as is, this tick
			continue;
		}
		if (size_conv_adapt_target(s, in.w, in.h)) {
			s->needRefresh = TRUE;
			ms_filter_notify_no_arg(f, MS_FILTER_OUTPUT_FMT_CHANGED);
		} else if (s->needRefresh) {
			ms_warning("MSSizeConv: output fmt changed, waiting.");
		} else if (size_conv_stage(f, s, in, mblk_get_timestamp_info(im))) {
			staged = true;
		} else {
			ms_error("MSSizeConv: error in ms_scaler_process().");
		}
		freemsg(im);
	}
	ms_filter_unlock(f);
	if (staged) {
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		request_flush(f);
	}
}

int sizeconv_set_vsize(MSFilter *f, void *arg) { // sizeconv.c:186-197
	SizeConvState *s = (SizeConvState *)f->data;
	ms_filter_lock(f);
	s->target_vsize = *(MSVideoSize *)arg;
	ms_message("sizeconv_set_vsize(): set target size w %d, h %d", s->target_vsize.width, s->target_vsize.height);
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		size_conv_leave_pool(s, f);
	}
	ms_filter_unlock(f);
	return 0;
}
int sizeconv_set_fps(MSFilter *f, void *arg) { // :199-204
	SizeConvState *s = (SizeConvState *)f->data;
	s->fps = *((float *)arg);
	s->frame_count = -1; /* reset counter used for fps */
	return 0;
}
int sizeconv_get_vsize(MSFilter *f, void *data) { // :206-212
	SizeConvState *s = (SizeConvState *)f->data;
	MSVideoSize *vsize = (MSVideoSize *)data;
	vsize->width = s->target_vsize.width;
	vsize->height = s->target_vsize.height;
	return 0;
}
MSFilterMethod sizeconv_methods[] = {{MS_FILTER_SET_FPS, sizeconv_set_fps}, // sizeconv.c:214-217
                                     {MS_FILTER_SET_VIDEO_SIZE, sizeconv_set_vsize},
                                     {MS_FILTER_GET_VIDEO_SIZE, sizeconv_get_vsize},
                                     {0, NULL}};

// ---- MSPixConv (src/videofilters/pixconv.c) --------------------------------------------------------------
struct PixConvState { // PixConvState pixconv.c:27-34
	MSYuvBufAllocator *allocator; // first: VideoOut
	MSVideoSize size;
	MSPixFmt in_fmt;
	MSPixFmt out_fmt;
	PixPool *pool;
	int slot;
};

void pixconv_leave_pool(PixConvState *s, MSFilter *f) {
	if (s->pool) {
		s->pool->forget(f);
		s->pool->release(s->slot);
	}
	s->pool = nullptr;
	s->slot = -1;
}
void pixconv_init(MSFilter *f) { // pixconv.c:36-45
	PixConvState *s = (PixConvState *)ms_malloc0(sizeof(PixConvState));
	s->allocator = ms_yuv_buf_allocator_new();
	s->size.width = MS_VIDEO_SIZE_CIF_W;
	s->size.height = MS_VIDEO_SIZE_CIF_H;
	s->in_fmt = MS_YUV420P;
	s->out_fmt = MS_YUV420P;
	s->slot = -1;
	f->data = s;
}
void pixconv_uninit(MSFilter *f) { // :47-55
	PixConvState *s = (PixConvState *)f->data;
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		pixconv_leave_pool(s, f);
	}
	ms_yuv_buf_allocator_free(s->allocator);
	ms_free(s);
}
void pixconv_process(MSFilter *f) { // pixconv.c:62-94
	PixConvState *s = (PixConvState *)f->data;
	mblk_t *im;
	bool staged_any = false;
	while ((im = ms_queue_get(f->inputs[0])) != NULL) {
		const uint32_t frame_ts = mblk_get_timestamp_info(im);
		if (s->in_fmt == s->out_fmt) {
			mblk_set_timestamp_info(im, frame_ts);
			ms_queue_put(f->outputs[0], im);
			continue;
		}
		MSPicture inbuf;
		if (ms_picture_init_from_mblk_with_size(&inbuf, im, s->in_fmt, s->size.width, s->size.height) == 0) {
			std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
			const int fmt = pix_to_mi(s->in_fmt);
			const int flip = s->in_fmt == MS_RGB24_REV; // :78-81
			if (fmt < 0 || (inbuf.w & 1)) {
				ms_error("MSPixConv: Error in ms_sws_scale()."); // what a failing ms_scaler_process logs, :84
			} else {
				if (!s->pool || s->pool->ticker != f->ticker || s->pool->out_w != inbuf.w || s->pool->out_h != inbuf.h) {
					pixconv_leave_pool(s, f);
					auto key = std::make_tuple(f->ticker, inbuf.w, inbuf.h, (int)s->in_fmt);
					auto it = g_pix_pools.find(key);
					if (it == g_pix_pools.end()) {
						mi_pixconv *pc = nullptr;
						if (mi_pixconv_create(g_hub.context(), inbuf.w, inbuf.h, fmt, flip, &pc) != MI_OK) {
							ms_error("MSPixConv: %dx%d format %d: %s", inbuf.w, inbuf.h, (int)s->in_fmt, mi_last_error());
							freemsg(im);
							continue;
						}
						PixPool *p = new PixPool(pc, inbuf.w, inbuf.h);
						p->ticker = f->ticker;
						g_hub.pools.push_back(p);
						it = g_pix_pools.emplace(key, p).first;
					}
					s->pool = it->second;
					s->slot = s->pool->acquire(f);
					if (s->slot < 0) s->pool = nullptr;
				}
				uint8_t *dst = s->pool ? s->pool->stage(f, frame_ts) : nullptr;
				if (dst) {
					memcpy(dst, inbuf.planes[0], s->pool->src_bytes);
					staged_any = true;
				}
			}
		}
		freemsg(im);
	}
	if (staged_any) {
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
		request_flush(f);
	}
}
int pixconv_set_vsize(MSFilter *f, void *arg) { // :96-100
	((PixConvState *)f->data)->size = *(MSVideoSize *)arg;
	return 0;
}
int pixconv_set_pixfmt(MSFilter *f, void *arg) { // :102-107
	PixConvState *s = (PixConvState *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	s->in_fmt = *(MSPixFmt *)arg;
	pixconv_leave_pool(s, f);
	return 0;
}
MSFilterMethod pixconv_methods[] = {{MS_FILTER_SET_VIDEO_SIZE, pixconv_set_vsize}, // pixconv.c:109-110
                                    {MS_FILTER_SET_PIX_FMT, pixconv_set_pixfmt},
                                    {0, NULL}};

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
};

struct MapPool : Pool {
	MapOp op;
	size_t cap, used = 0; // elements
	uint8_t *h_in, *h_in2 = nullptr, *h_out, *d_in, *d_in2 = nullptr, *d_out;
	std::vector<std::vector<MapBlock>> staged, ready;
	explicit MapPool(MapOp o) : op(o) {
		init_slots(g_hub.capacity);
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
	void flush() override {
		if (used) {
			mi_ctx *ctx = g_hub.context();
			const MapOpInfo &k = kMapOps[op];
			MI_MUST(mi_copy_h2d(ctx, d_in, h_in, used * k.in_bpe));
			if (d_in2) MI_MUST(mi_copy_h2d(ctx, d_in2, h_in2, used * k.in_bpe));
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
			MI_MUST(mi_copy_d2h(ctx, h_out, d_out, used * k.out_bpe));
			MI_MUST(mi_ctx_sync(ctx));
		}
		for (int s = 0; s < capacity; ++s) {
			auto &st = staged[(size_t)s], &rd = ready[(size_t)s];
			rd.insert(rd.end(), st.begin(), st.end());
			st.clear();
		}
		used = 0;
	}
	void emit(MSFilter *f, int slot) override {
		const size_t bpe = kMapOps[op].out_bpe;
		for (const MapBlock &b : ready[(size_t)slot]) {
			mblk_t *o = allocb(b.n * bpe, 0);
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
			for (MapBlock &b : *v)
				if (b.meta) freemsg(b.meta);
			v->clear();
		}
	}
};
std::map<std::pair<MSTicker *, int>, MapPool *> g_map_pools;

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
};

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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	d->pool->drop_slot(d->slot);
	d->pool->release(d->slot);
	d->pool = nullptr;
	d->slot = -1;
}

// the pool of (this ticker, op) and a slot in it; false when the pool is exhausted
bool map_attach(MSFilter *f, MapFilter *d, MapOp op) {
	if (d->pool && d->pool->op == op && d->pool->ticker == f->ticker) return true;
	map_release(d);
	auto key = std::make_pair(f->ticker, (int)op);
	auto it = g_map_pools.find(key);
	if (it == g_map_pools.end()) {
		MapPool *p = new MapPool(op);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_map_pools.emplace(key, p).first;
	}
	const int sl = it->second->acquire(f);
	if (sl < 0) return false;
	d->pool = it->second;
	d->slot = sl;
	return true;
}

void map_uninit(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
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
void g711_dec_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	ms_bufferizer_put_from_queue(d->bz, f->inputs[0]);
	if (ms_bufferizer_get_avail(d->bz) < size_of_pcm) return;
	if (!map_attach(f, d, d->law ? OP_ULAW_ENC : OP_ALAW_ENC)) {
		ms_bufferizer_flush(d->bz);
		return;
	}
	while (ms_bufferizer_get_avail(d->bz) >= size_of_pcm) {
		uint8_t *dst = d->pool->reserve(d->slot, size_of_pcm / 2, nullptr, true, d->ts);
		if (!dst) break;
		ms_bufferizer_read(d->bz, dst, size_of_pcm);
		d->ts += (uint32_t)(size_of_pcm / 2);
	}
	request_flush(f);
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
void l16_enc_preprocess(MSFilter *f) { l16_enc_update((MapFilter *)f->data); }
void l16_enc_process(MSFilter *f) {
	MapFilter *d = (MapFilter *)f->data;
	ms_filter_lock(f);
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
}
void adapter_postprocess(MSFilter *f) { // :125-135
	MapFilter *d = (MapFilter *)f->data;
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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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

// ---- MSAudioFlowControl flowcontrol.c:154-279
constexpr int kFlowBlock = 2048; // samples per staged block (mi_flowctl's limit); longer blocks are split
struct FlowPool : Pool {
	mi_flowctl *fc = nullptr;
	int16_t *h_in, *h_out, *d_in, *d_out;
	int32_t *h_len, *h_olen, *d_len, *d_olen;
	std::vector<uint32_t> req_drop, req_total; // pending MS_AUDIO_FLOW_CONTROL_DROP requests ...
	std::vector<int> req_round;                // ... and how many staged blocks of the stream precede each
	std::vector<uint32_t> arm_drop, arm_total;
	bool have_req = false;
	std::vector<int> staged, ready;
	std::vector<std::vector<mblk_t *>> held, done; // the blocks themselves: the dropper edits them in place
	FlowPool() {
		init_slots(g_hub.capacity);
		MI_MUST(mi_flowctl_create(g_hub.context(), capacity, kFlowBlock, &fc));
		const size_t c = (size_t)capacity;
		h_in = pinned<int16_t>(kMaxRounds * c * kFlowBlock);
		h_out = pinned<int16_t>(kMaxRounds * c * kFlowBlock);
		h_len = pinned<int32_t>(kMaxRounds * c);
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
		staged.assign(c, 0);
		ready.assign(c, 0);
		held.resize(c);
		done.resize(c);
	}
	void flush() override {
		mi_ctx *ctx = g_hub.context();
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
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			arm(r, false);
			for (int s = 0; s < capacity; ++s)
				if (staged[(size_t)s] <= r) h_len[r * c + s] = 0;
			MI_MUST(mi_copy_h2d(ctx, d_in, h_in + r * c * kFlowBlock, c * kFlowBlock * 2));
			MI_MUST(mi_copy_h2d(ctx, d_len, h_len + r * c, c * 4));
			MI_MUST(mi_flowctl_process(fc, d_in, kFlowBlock, d_len, kFlowBlock, d_out, kFlowBlock, d_olen));
			MI_MUST(mi_copy_d2h(ctx, h_out + r * c * kFlowBlock, d_out, c * kFlowBlock * 2));
			MI_MUST(mi_copy_d2h(ctx, h_olen + r * c, d_olen, c * 4));
		}
		arm(maxr, true);
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			ready[(size_t)s] = staged[(size_t)s];
			staged[(size_t)s] = 0;
			done[(size_t)s].swap(held[(size_t)s]);
			held[(size_t)s].clear();
		}
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
std::map<MSTicker *, FlowPool *> g_flow_pools;

struct FlowFilter { // MSAudioFlowControlState :154-158
	FlowPool *pool;
	int slot;
	int samplerate, nchannels;
	MSAudioFlowControlConfig config;
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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	const size_t s = (size_t)d->slot;
	for (auto *v : {&d->pool->held[s], &d->pool->done[s]}) {
		for (mblk_t *m : *v) freemsg(m);
		v->clear();
	}
	d->pool->staged[s] = d->pool->ready[s] = 0;
	d->pool->req_drop[s] = d->pool->req_total[s] = 0;
	d->pool->release(d->slot);
	d->pool = nullptr;
	d->slot = -1;
}
bool flowctl_attach(MSFilter *f, FlowFilter *d) {
	if (d->pool && d->pool->ticker == f->ticker) return true;
	flowctl_release(d);
	auto it = g_flow_pools.find(f->ticker);
	if (it == g_flow_pools.end()) {
		FlowPool *p = new FlowPool();
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_flow_pools.emplace(f->ticker, p).first;
	}
	const int sl = it->second->acquire(f);
	if (sl < 0) return false;
	d->pool = it->second;
	d->slot = sl;
	MI_MUST(mi_flowctl_reset(d->pool->fc, sl, 1));
	MI_MUST(mi_flowctl_set_config(d->pool->fc, sl, 1, d->config.strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT,
	                              d->config.silent_threshold));
	return true;
}
void flowctl_preprocess(MSFilter *f) { // :166-169 ms_audio_flow_controller_reset
	FlowFilter *d = (FlowFilter *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	if (flowctl_attach(f, d)) MI_MUST(mi_flowctl_reset(d->pool->fc, d->slot, 1));
}
void flowctl_process(MSFilter *f) { // :171-183
	FlowFilter *d = (FlowFilter *)f->data;
	ms_filter_lock(f);
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
void flowctl_postprocess(MSFilter *f) { flowctl_release((FlowFilter *)f->data); }
void flowctl_uninit(MSFilter *f) { // :188-191
	flowctl_release((FlowFilter *)f->data);
	ms_free(f->data);
}
int flowctl_set_config(MSFilter *f, void *arg) { // :193-197
	FlowFilter *d = (FlowFilter *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	d->config = *(MSAudioFlowControlConfig *)arg;
	if (d->pool)
		MI_MUST(mi_flowctl_set_config(d->pool->fc, d->slot, 1, d->config.strategy == MSAudioFlowControlBasic ? MI_FLOWCTL_BASIC : MI_FLOWCTL_SOFT,
		                              d->config.silent_threshold));
	return 0;
}
int flowctl_drop(MSFilter *f, void *arg) { // :199-211; applied by the next launch at this point of the block sequence
	FlowFilter *d = (FlowFilter *)f->data;
	const MSAudioFlowControlDropEvent *ev = (const MSAudioFlowControlDropEvent *)arg;
	ms_filter_lock(f);
	{
		std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
	int32_t *h_len, *d_len;
	uint8_t *h_mode, *d_mode;
	std::vector<int> staged;
	std::vector<std::vector<PlcEntry>> pending, done;
	explicit PlcPool(int r) : rate(r) {
		init_slots(g_hub.capacity);
		MI_MUST(mi_plc_create(g_hub.context(), capacity, rate, kPlcBlock, &plc));
		const size_t c = (size_t)capacity;
		h_rows = pinned<int16_t>(kMaxRounds * c * kPlcBlock);
		h_len = pinned<int32_t>(kMaxRounds * c);
		h_mode = pinned<uint8_t>(kMaxRounds * c);
		d_rows = devmem<int16_t>(c * kPlcBlock);
		d_len = devmem<int32_t>(c);
		d_mode = devmem<uint8_t>(c);
		staged.assign(c, 0);
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
	void flush() override {
		mi_ctx *ctx = g_hub.context();
		const size_t c = (size_t)capacity;
		int maxr = 0;
		for (int s = 0; s < capacity; ++s) maxr = std::max(maxr, staged[(size_t)s]);
		for (int r = 0; r < maxr; ++r) {
			for (int s = 0; s < capacity; ++s)
				if (staged[(size_t)s] <= r) h_mode[r * c + s] = MI_PLC_NONE, h_len[r * c + s] = 0;
			MI_MUST(mi_copy_h2d(ctx, d_rows, h_rows + r * c * kPlcBlock, c * kPlcBlock * 2));
			MI_MUST(mi_copy_h2d(ctx, d_len, h_len + r * c, c * 4));
			MI_MUST(mi_copy_h2d(ctx, d_mode, h_mode + r * c, c));
			MI_MUST(mi_plc_process(plc, d_rows, kPlcBlock, d_len, d_mode));
			MI_MUST(mi_copy_d2h(ctx, h_rows + r * c * kPlcBlock, d_rows, c * kPlcBlock * 2));
		}
		if (maxr) MI_MUST(mi_ctx_sync(ctx));
		for (int s = 0; s < capacity; ++s) {
			auto &p = pending[(size_t)s], &d = done[(size_t)s];
			d.insert(d.end(), p.begin(), p.end());
			p.clear();
			staged[(size_t)s] = 0;
		}
	}
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
std::map<std::pair<MSTicker *, int>, PlcPool *> g_plc_pools;

struct PlcFilter { // generic_plc_struct msgenericplc.c:30-41
	PlcPool *pool;
	int slot;
	Concealer *concealer;
	int rate, nchannels;
	bool cng_set, cng_running;
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
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
	if (d->pool && d->pool->ticker == f->ticker && d->pool->rate == d->rate) return true;
	plc_release(d);
	auto key = std::make_pair(f->ticker, d->rate);
	auto it = g_plc_pools.find(key);
	if (it == g_plc_pools.end()) {
		mi_plc *probe = nullptr; // a rate the kernel does not take (44.1 kHz family) must not abort the process: pass-through
		if (mi_plc_create(g_hub.context(), 1, d->rate, kPlcBlock, &probe) != MI_OK) {
			ms_error("msmi355x plugin: MSGenericPLC at %d Hz: %s; audio is forwarded without concealment", d->rate, mi_last_error());
			return false;
		}
		mi_plc_destroy(probe);
		PlcPool *p = new PlcPool(d->rate);
		p->ticker = f->ticker;
		g_hub.pools.push_back(p);
		it = g_plc_pools.emplace(key, p).first;
	}
	const int sl = it->second->acquire(f);
	if (sl < 0) return false;
	d->pool = it->second;
	d->slot = sl;
	MI_MUST(mi_plc_reset(d->pool->plc, sl, 1));
	return true;
}
void plc_preprocess(MSFilter *f) {
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	plc_attach(f, (PlcFilter *)f->data);
}
void plc_process(MSFilter *f) { // generic_plc_process :59-167
	PlcFilter *d = (PlcFilter *)f->data;
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
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
void plc_postprocess(MSFilter *f) { plc_release((PlcFilter *)f->data); }
void plc_uninit(MSFilter *f) { // :169-178
	PlcFilter *d = (PlcFilter *)f->data;
	plc_release(d);
	delete d->concealer;
	ms_free(d);
}
int plc_get_sr(MSFilter *f, void *arg) {
	*(int *)arg = ((PlcFilter *)f->data)->rate;
	return 0;
}
int plc_set_sr(MSFilter *f, void *arg) {
	((PlcFilter *)f->data)->rate = *(int *)arg;
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

} // namespace

extern "C" {

// Descriptors: same ids, names, pin counts and flags as the reference's, plus
// MS_FILTER_IS_HW_ACCELERATED (msfilter.h:142).  Writable statics: the factory mutates flags.
MSFilterDesc ms_mi355x_resample_desc = {MS_RESAMPLE_ID, "MSResample", "Audio resampler (MI355X batch)", MS_FILTER_OTHER,
                                        NULL, 1, 1, resample_init, NULL, resample_process, NULL, resample_uninit,
                                        resample_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_audio_mixer_desc = {MS_AUDIO_MIXER_ID, "MSAudioMixer",
                                           "A filter that mixes down 16 bit sample audio streams (MI355X batch)",
                                           MS_FILTER_OTHER, NULL, MIXER_MAX_CHANNELS, MIXER_MAX_CHANNELS, mixer_init,
                                           mixer_preprocess, mixer_process, mixer_postprocess, mixer_uninit,
                                           mixer_methods, MS_FILTER_IS_PUMP | MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_volume_desc = {MS_VOLUME_ID, "MSVolume", "A filter that controls and measure sound volume (MI355X batch)",
                                      MS_FILTER_OTHER, NULL, 1, 1, volume_init, volume_preprocess, volume_process, NULL,
                                      volume_uninit, volume_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_equalizer_desc = {MS_EQUALIZER_ID, "MSEqualizer", "Parametric sound equalizer (MI355X batch)",
                                         MS_FILTER_OTHER, NULL, 1, 1, equalizer_init, equalizer_preprocess, equalizer_process, NULL,
                                         equalizer_uninit, equalizer_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_speex_ec_desc = {MS_SPEEX_EC_ID, "MSSpeexEC", "Echo canceller, MDF + post-filter (MI355X batch)",
                                        MS_FILTER_OTHER, NULL, 2, 2, ec_init, ec_preprocess, ec_process, ec_postprocess,
                                        ec_uninit, ec_methods, MS_FILTER_IS_HW_ACCELERATED};

MSFilterDesc ms_mi355x_size_conv_desc = {MS_SIZE_CONV_ID, "MSSizeConv", "A video size converter (MI355X batch)", MS_FILTER_OTHER,
                                         NULL, 1, 1, size_conv_init, NULL, size_conv_process, size_conv_postprocess,
                                         size_conv_uninit, sizeconv_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_pix_conv_desc = {MS_PIX_CONV_ID, "MSPixConv", "A pixel format converter (MI355X batch)", MS_FILTER_OTHER,
                                        NULL, 1, 1, pixconv_init, NULL, pixconv_process, NULL, pixconv_uninit,
                                        pixconv_methods, MS_FILTER_IS_HW_ACCELERATED};
MSScalerDesc ms_mi355x_scaler_desc = {sd_create, sd_process, sd_free};

// the stages either side of the path
MSFilterDesc ms_mi355x_alaw_dec_desc = {MS_ALAW_DEC_ID, "MSAlawDec", "ITU-G.711 alaw decoder (MI355X batch)", MS_FILTER_DECODER, "pcma", 1, 1,
                                        g711_dec_init_a, NULL, g711_dec_process, NULL, map_uninit, g711_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_ulaw_dec_desc = {MS_ULAW_DEC_ID, "MSUlawDec", "ITU-G.711 ulaw decoder (MI355X batch)", MS_FILTER_DECODER, "pcmu", 1, 1,
                                        g711_dec_init_u, NULL, g711_dec_process, NULL, map_uninit, g711_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_alaw_enc_desc = {MS_ALAW_ENC_ID, "MSAlawEnc", "ITU-G.711 alaw encoder (MI355X batch)", MS_FILTER_ENCODER, "pcma", 1, 1,
                                        g711_enc_init_a, NULL, g711_enc_process, NULL, map_uninit, g711_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_ulaw_enc_desc = {MS_ULAW_ENC_ID, "MSUlawEnc", "ITU-G.711 ulaw encoder (MI355X batch)", MS_FILTER_ENCODER, "pcmu", 1, 1,
                                        g711_enc_init_u, NULL, g711_enc_process, NULL, map_uninit, g711_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_l16_enc_desc = {MS_L16_ENC_ID, "MSL16Enc", "L16 dummy encoder (MI355X batch)", MS_FILTER_ENCODER, "L16", 1, 1,
                                       l16_enc_init, l16_enc_preprocess, l16_enc_process, NULL, map_uninit, l16_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_l16_dec_desc = {MS_L16_DEC_ID, "MSL16Dec", "L16 dummy decoder (MI355X batch)", MS_FILTER_DECODER, "L16", 1, 1,
                                       l16_dec_init, NULL, l16_dec_process, NULL, map_uninit, l16_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_channel_adapter_desc = {MS_CHANNEL_ADAPTER_ID, "MSChannelAdapter",
                                               "A filter that converts from mono to stereo and vice versa (MI355X batch)", MS_FILTER_OTHER, NULL, 2, 1,
                                               adapter_init, adapter_preprocess, adapter_process, adapter_postprocess, adapter_uninit,
                                               adapter_methods, MS_FILTER_IS_PUMP | MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_generic_plc_desc = {MS_GENERIC_PLC_ID, "MSGenericPLC", "Generic PLC (MI355X batch)", MS_FILTER_OTHER, NULL, 1, 1,
                                           plc_init, plc_preprocess, plc_process, plc_postprocess, plc_uninit, plc_methods,
                                           MS_FILTER_IS_PUMP | MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_audio_flow_control_desc = {MS_AUDIO_FLOW_CONTROL_ID, "MSAudioFlowControl",
                                                  "Flow control filter to drop sample in the audio graph if too many samples are queued (MI355X batch)",
                                                  MS_FILTER_OTHER, NULL, 1, 1, flowctl_init, flowctl_preprocess, flowctl_process, flowctl_postprocess,
                                                  flowctl_uninit, flowctl_methods, MS_FILTER_IS_HW_ACCELERATED};

void libmsmi355xfilters_init(MSFactory *factory) {
	ms_factory_register_filter(factory, &ms_mi355x_resample_desc);
	ms_factory_register_filter(factory, &ms_mi355x_audio_mixer_desc);
	ms_factory_register_filter(factory, &ms_mi355x_volume_desc);
	ms_factory_register_filter(factory, &ms_mi355x_equalizer_desc);
	ms_factory_register_filter(factory, &ms_mi355x_speex_ec_desc);
	ms_factory_register_filter(factory, &ms_mi355x_size_conv_desc);
	ms_factory_register_filter(factory, &ms_mi355x_pix_conv_desc);
	for (MSFilterDesc *d : {&ms_mi355x_alaw_dec_desc, &ms_mi355x_ulaw_dec_desc, &ms_mi355x_alaw_enc_desc, &ms_mi355x_ulaw_enc_desc,
	                        &ms_mi355x_l16_enc_desc, &ms_mi355x_l16_dec_desc, &ms_mi355x_channel_adapter_desc, &ms_mi355x_audio_flow_control_desc,
	                        &ms_mi355x_generic_plc_desc})
		ms_factory_register_filter(factory, d);
	ms_video_set_scaler_impl(&ms_mi355x_scaler_desc); // msvideo.c:719-721: the reference's own video filters follow
	ms_message("libmsmi355xfilters: MI355X batched filters registered (ABI %d)", mi_abi_version());
}

void ms_mi355x_flush(void) { flush_ticker(nullptr); }

void ms_mi355x_shutdown(void) {
	std::lock_guard<std::recursive_mutex> lk(g_hub.mu);
	// pools keep their device objects for the life of the process (like the reference's plugins,
	// there is no unload hook: src/base/msfactory.c:761-771); only the context is synchronised here
	if (g_hub.ctx) mi_ctx_sync(g_hub.ctx);
}

} // extern "C"
