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
//
// Runtime: everything is per TICKER.  A TickerHub owns the pools of the filters one MSTicker
// thread runs, its own mi_ctx (HIP stream) and its own mutex: process() of filters on different tickers never contend.
// Pools are banks of geometrically growing capacity (16, 64, 256, ... slots): a ticker with one call uploads 16 rows, a
// ticker with ten thousand legs gets there in six banks; a bank is destroyed (device objects, pinned buffers) when its
// last slot is released and the hub with its last bank.  No runtime error aborts the host process: a failing kernel
// library call marks the bank failed, its filters drop their blocks and count a late event, and when no HIP device
// can be opened at load time the plugin registers nothing and the reference's own filters stay in charge.
#include "../../include/ms2_plugin_abi.h"
#include "../../include/msmi355x.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

extern "C" { // the descriptors (defined at the end of this file): the fused chain recognises its facades by them
extern MSFilterDesc ms_mi355x_resample_desc, ms_mi355x_audio_mixer_desc, ms_mi355x_volume_desc, ms_mi355x_speex_ec_desc, ms_mi355x_webrtc_aec_name_desc;
extern MSFilterDesc ms_mi355x_equalizer_desc;
extern MSFilterDesc ms_mi355x_alaw_enc_desc, ms_mi355x_ulaw_enc_desc, ms_mi355x_alaw_dec_desc, ms_mi355x_ulaw_dec_desc;
}

namespace {

constexpr int kMaxRounds = 4; // blocks one stream may hand over within a single tick

struct TickerHub;
struct Pool;
thread_local Pool *tl_building = nullptr; // the bank whose constructor is running (allocations register with it)
std::atomic<uint64_t> g_late_events{0};   // blocks dropped or passed through because the GPU path failed

// A kernel-library call failed: no abort (this plugin sits inside a media server).  The bank under construction / at
// hand is marked failed by the caller; the event is logged and counted.
bool mi_failed(const char *what) {
	ms_error("msmi355x plugin: %s: %s", what, mi_last_error());
	g_late_events.fetch_add(1, std::memory_order_relaxed);
	return true;
}

// a kernel-library call that must not fail: logged and counted, never fatal.  Inside a bank's member functions
// (constructor, flush) name lookup finds Pool::mark_failed and the bank stops being used; elsewhere the free function.
inline void mark_failed() {}
#define MI_MUST(expr)              \
	do {                           \
		if ((expr) != MI_OK) {     \
			mi_failed(#expr);      \
			mark_failed();         \
		}                          \
	} while (0)

struct Pool {
	virtual ~Pool();
	// A bank's flush in two halves, so that a hub's flush costs ONE synchronisation however many banks it holds:
	//   enqueue()  uploads, launches and downloads of everything staged, asynchronous on the hub's stream; true = something
	//              was enqueued (the hub then synchronises once, behind the last bank);
	//   finish()   the bookkeeping that needs the results on the host (staged -> ready); then emit() per slot.
	// A bank that has not been split keeps the defaults: nothing enqueued, its whole flush() -- with a synchronisation of
	// its own -- in the finish phase.
	virtual bool enqueue() { return false; }
	virtual void finish() { flush(); }
	virtual void flush() {                         // launch the staged blocks, fetch the results: enqueue, wait, finish
		if (enqueue()) sync_stream();
		finish();
	}
	void sync_stream();
	virtual void emit(MSFilter *f, int slot) = 0;  // hand a slot's results to its filter's output queues
	virtual void emitted() {}                      // every slot has emitted: what the bank held for them may go
	virtual void flushed() {}                      // a hub flush is through, all of its rounds (flush_hub)
	// A graph that is being detached has its tick in flight delivered before its filters let go (TickerHub::scope): a bank that
	// can flush the slots of that graph alone -- every other slot's staged rows, counts and state left exactly as they are, for
	// the ticker's own flush -- says so here and asks parked(slot) wherever it looks at what a slot staged
	virtual bool scoped() const { return false; }
	bool parked(int slot) const;
	// A method between two walks meets the NEXT walk's blocks in the reference.  Here the last walk's blocks may still be waiting
	// for the coming flush (staged, or on their way down a chain of facades): what a method sets then waits too and goes live
	// when that flush is through (flushed()); with nothing waiting it is live at once.
	bool work_waiting() const;
	TickerHub *hub = nullptr;
	std::string key;
	std::vector<MSFilter *> owner;
	std::vector<int> free_list; // O(1) acquire / release
	std::vector<void *> host_allocs, dev_allocs;
	int capacity = 0, in_use = 0;
	int hi = 0;          // slots [0, hi) have been handed out at some point: what flush() needs to look at
	bool failed = false; // a kernel-library call failed: the bank's filters stop using the GPU path
	void mark_failed() { failed = true; }
	void init_slots(int cap) {
		capacity = cap;
		owner.assign((size_t)cap, nullptr);
		free_list.resize((size_t)cap);
		for (int i = 0; i < cap; ++i) free_list[(size_t)i] = cap - 1 - i; // lowest slot first
	}
	int acquire(MSFilter *f) {
		if (free_list.empty() || failed) return -1;
		const int s = free_list.back();
		free_list.pop_back();
		owner[(size_t)s] = f;
		hi = std::max(hi, s + 1);
		++in_use;
		return s;
	}
	void release(int slot); // may destroy the bank (and the hub): do not touch either afterwards
	void emit_all() {
		for (int i = 0; i < hi; ++i)
			if (owner[(size_t)i] && !parked(i)) emit(owner[(size_t)i], i);
	}
	mi_ctx *ctx() const;
	template <typename T>
	T *pinned(size_t n) {
		void *p = failed ? nullptr : mi_host_alloc(ctx(), n * sizeof(T));
		if (!p) {
			failed = mi_failed("mi_host_alloc");
			p = calloc(n ? n : 1, sizeof(T)); // keeps the constructors' pointer arithmetic valid; never used for a launch
			host_fallback.push_back(p);
			return (T *)p;
		}
		memset(p, 0, n * sizeof(T));
		host_allocs.push_back(p);
		return (T *)p;
	}
	template <typename T>
	T *devmem(size_t n) {
		void *p = failed ? nullptr : mi_dev_alloc(ctx(), n * sizeof(T));
		if (!p) {
			failed = mi_failed("mi_dev_alloc");
			return nullptr;
		}
		dev_allocs.push_back(p);
		return (T *)p;
	}
	std::vector<void *> host_fallback;
};

struct TickerHub {
	std::recursive_mutex mu;
	MSTicker *ticker = nullptr;
	// the hub's device context (HIP stream), opened by the first bank or scaler context on the least-loaded device of
	// MSMI355X_DEVICES -- one mediastreamer2 process runs a ticker per stream / per conference (mediastream.c:237,
	// audioconference.c:72), so its tickers spread over the node's GPUs; a hub that never gets a bank opens none
	mi_ctx *ctx = nullptr;
	int device = -1;
	bool ctx_failed = false;
	mi_ctx *ensure_ctx();
	int pins = 0; // objects outside the banks that use `ctx` (MSScalerDesc contexts): the hub stays while there are any
	std::vector<Pool *> pools;         // flush order = creation order
	MSFilter *flush_owner = nullptr;   // the filter whose postponed task will flush this ticker's pools (NULL: none pending)
	uint32_t flush_posted_tick = 0;    // ... and the tick it was posted in (MSTicker::ticks): it runs at the head of the next one -- or it was dropped (request_flush)
	// More tasks for the same flush, on OTHER filters (the 2nd, 4th, 8th, 16th, 64th, 256th, 1024th and 4096th to ask in the walk: other graphs,
	// as a walk goes graph by graph, whether the graphs are two-filter chains or conferences of 32 legs): a task goes with its filter when that filter's graph is detached (msticker.c:187-190), on the application's thread, between two
	// ticks -- and the hub's flush is every graph's.  (Found by the churn run: re-plumbed in the order they are walked, the graphs took the
	// flush with them tick after tick; the others' results then arrived four ticks at once -- and a conference's 32 members ask one after the
	// other, the sixteenth is no safer than the first.)  Whichever task runs first flushes, the others find the tick done (flush_done_tick).
	static constexpr int kFlushBackups = 8;
	MSFilter *flush_backup[kFlushBackups] = {};
	uint32_t flush_asks = 0, flush_done_tick = 0;
	bool backup_waiting() const {
		for (const MSFilter *bk : flush_backup)
			if (bk) return true;
		return false;
	}
	void drop_backups() {
		for (MSFilter *&bk : flush_backup) bk = nullptr;
	}
	// chain linking: while the flush task runs, a facade that emits into a queue read by ANOTHER facade of this ticker has
	// that one run right away (then its bank is flushed in the same task): a chain of GPU filters costs one tick, not one
	// tick per filter
	bool in_flush = false;
	uint64_t flushes = 0; // rounds of (enqueue, one wait, emit) so far: what ms_mi355x_hub_stats reports
	std::vector<MSFilter *> touched, touched_pumps;
	std::unordered_map<MSFilter *, uint64_t> pumped; // pump facades run early by the flush task, and for which tick
	// The reference's filters are synchronous: ms_ticker_detach finds nothing in flight (msticker.c:197-218).  Here a walk's blocks
	// are staged and come back with the next tick's flush, so the FIRST postprocess of a graph that is being detached flushes that
	// graph -- and only it: `scope` holds its filters while that flush runs (on the application's thread, under `mu`; the ticker
	// thread may be walking the ticker's other graphs, whose staged work, queues and filters are not touched).  drained /
	// drained_seq: the graph flushed last and the staging count it was flushed at (the graph's other postprocess calls find it there)
	const std::unordered_set<MSFilter *> *scope = nullptr;
	std::unordered_set<MSFilter *> drained;
	uint64_t stage_seq = 0, drained_seq = (uint64_t)-1;
	// Lifetime: ONE atomic word = the number of scopes that hold or are about to take `mu` (references: taken under the
	// registry lock -- hub_for, referenced_hubs -- or while a slot / pin of the hub is held, so a hub found in the registry
	// cannot be freed between the look-up and the lock) + the RETIRED bit (no banks left: out of the registry; set under
	// `mu` and the registry lock by a scope that holds a reference).  Whoever drops the last reference of a retired hub
	// deletes it, and learns both facts from the value its decrement returns: after the decrement a scope must not touch
	// the hub any more (another scope may have retired and deleted it in between -- ThreadSanitizer found exactly that
	// read-after-decrement in an earlier form with a separate flag).
	static constexpr unsigned RETIRED = 1u << 30;
	std::atomic<unsigned> life{0};
	void ref() { life.fetch_add(1, std::memory_order_acq_rel); }
	bool retired() const { return (life.load(std::memory_order_acquire) & RETIRED) != 0; }
	unsigned refs() const { return life.load(std::memory_order_acquire) & (RETIRED - 1); }
};

mi_ctx *Pool::ctx() const { return hub->ctx; }
// MSTicker::time as a bank reads it.  The ticker's own thread advances it between ticks (msticker.c:494-495); a bank may be at work on the
// APPLICATION's thread then -- a postprocess delivering a detaching graph's tick in flight -- and reads whichever of the two values: a relaxed load
inline uint64_t hub_time(const TickerHub *h) { return h->ticker ? __atomic_load_n(&h->ticker->time, __ATOMIC_RELAXED) : 0; }
inline uint64_t ticker_now(const MSTicker *t) { return __atomic_load_n(&t->time, __ATOMIC_RELAXED); } // (the same for code a detaching graph's flush may run on the application's thread)
bool Pool::work_waiting() const { return (hub->flush_owner != nullptr || hub->backup_waiting()) && !hub->in_flush; }
bool Pool::parked(int slot) const { return hub->scope && !(owner[(size_t)slot] && hub->scope->count(owner[(size_t)slot])); }
void Pool::sync_stream() {
	if (hub->ctx && mi_ctx_sync(hub->ctx) != MI_OK) failed = mi_failed("mi_ctx_sync");
}

std::mutex g_registry_mu; // (a plain mutex: under sixteen threads attaching at once glibc's reader-preferring rwlock starved the writers -- note_slot -- for 70 % of the attach)
std::unordered_map<MSTicker *, TickerHub *> g_hubs;          // the hub of a ticker
struct FilterRef {
	TickerHub *hub;
	int slots;
};
std::unordered_map<MSFilter *, FilterRef> g_filter_hubs; // the hub a filter holds slots in (it may be detached by now)
thread_local TickerHub *tl_hub = nullptr;
// devices the hubs are spread over: MSMI355X_DEVICES="0,1,.." (default: every visible device), or the one of the older
// MSMI355X_DEVICE; g_device_hubs counts the live contexts per device
std::vector<int> g_devices;
std::atomic<int> g_device_hubs[64];
std::atomic<unsigned> g_device_rr{0};

void parse_devices() {
	g_devices.clear();
	const char *list = getenv("MSMI355X_DEVICES"), *one = getenv("MSMI355X_DEVICE");
	if (list && *list) {
		for (const char *p = list; *p;) {
			char *end = nullptr;
			const long v = strtol(p, &end, 10);
			if (end == p) break;
			if (v >= 0 && v < 64) g_devices.push_back((int)v);
			p = (*end == ',') ? end + 1 : end;
			if (*end != ',' ) break;
		}
	} else if (one && *one) {
		char *end = nullptr;
		const long v = strtol(one, &end, 10);
		if (end != one && v >= 0 && v < 64) g_devices.push_back((int)v); // same bound as the list: g_device_hubs[64]
		else ms_error("msmi355x plugin: MSMI355X_DEVICE=%s is not a device index (0..63); using every visible device", one);
	}
	if (g_devices.empty()) {
		const int n = mi_device_count();
		for (int i = 0; i < n && i < 64; ++i) g_devices.push_back(i);
	}
	if (g_devices.empty()) g_devices.push_back(0); // no device: mi_ctx_create reports it
}

mi_ctx *TickerHub::ensure_ctx() {
	if (ctx || ctx_failed) return ctx;
	if (g_devices.empty()) parse_devices();
	// least-loaded device, ties broken round-robin (tickers come and go with the calls they serve)
	const unsigned start = g_device_rr.fetch_add(1, std::memory_order_relaxed);
	int best = g_devices[start % g_devices.size()];
	for (size_t i = 0; i < g_devices.size(); ++i) {
		const int d = g_devices[(start + i) % g_devices.size()];
		if (g_device_hubs[d].load(std::memory_order_relaxed) < g_device_hubs[best].load(std::memory_order_relaxed)) best = d;
	}
	if (mi_ctx_create(best, nullptr, &ctx) != MI_OK) {
		mi_failed("mi_ctx_create");
		ctx = nullptr; // banks created on this hub fail at their first allocation and their filters fall back
		ctx_failed = true;
		return nullptr;
	}
	device = best;
	g_device_hubs[best].fetch_add(1, std::memory_order_relaxed);
	return ctx;
}

void destroy_hub(TickerHub *hub) { // the last scope of a retired hub
	if (hub->ctx) {
		mi_ctx_destroy(hub->ctx);
		g_device_hubs[hub->device].fetch_sub(1, std::memory_order_relaxed);
	}
	delete hub;
}

TickerHub *hub_for(MSFilter *f, bool create) {
	{
		std::lock_guard<std::mutex> rl(g_registry_mu);
		auto fi = g_filter_hubs.find(f);
		if (fi != g_filter_hubs.end()) return fi->second.hub->ref(), fi->second.hub;
		auto hi = g_hubs.find(f ? f->ticker : nullptr);
		if (hi != g_hubs.end()) return hi->second->ref(), hi->second;
	}
	if (!create) return nullptr;
	std::unique_lock<std::mutex> wl(g_registry_mu);
	MSTicker *t = f ? f->ticker : nullptr;
	auto hi = g_hubs.find(t);
	if (hi != g_hubs.end()) return hi->second->ref(), hi->second;
	TickerHub *h = new TickerHub();
	h->ref();
	h->ticker = t;
	g_hubs[t] = h;
	return h;
}

// Scope of every facade entry point: finds (or creates) the hub the filter belongs to, locks it, makes it the thread's
// current hub.  process() / preprocess() / the flush task run on the ticker thread; methods and uninit on any thread.
struct HubLock {
	TickerHub *h, *prev;
	static void unref(TickerHub *hub) {
		if (hub->life.fetch_sub(1, std::memory_order_acq_rel) == (TickerHub::RETIRED | 1u)) destroy_hub(hub); // last scope of a retired hub
	}
	// A hub that holds no bank when its last scope ends (a filter at an unsupported configuration, a failed bank, a method
	// call on a filter that never ran) must not stay in the registry with its HIP stream: retired here, under the hub's
	// lock.  With the registry locked exclusively nobody can take a new reference (hub_for / referenced_hubs take theirs
	// under it, the slot-based constructors need a bank), so refs == 1 means this scope is the only one.
	static void retire_if_idle(TickerHub *hub) {
		if (hub->retired() || !hub->pools.empty() || hub->pins > 0) return;
		std::unique_lock<std::mutex> wl(g_registry_mu);
		if (hub->refs() != 1) return;
		auto it = g_hubs.find(hub->ticker);
		if (it != g_hubs.end() && it->second == hub) g_hubs.erase(it);
		hub->flush_owner = nullptr, hub->drop_backups();
		hub->life.fetch_or(TickerHub::RETIRED, std::memory_order_acq_rel);
	}
	explicit HubLock(MSFilter *f) : h(nullptr), prev(tl_hub) {
		for (;;) { // hub_for hands the hub over with a reference taken under the registry lock
			h = hub_for(f, true);
			h->mu.lock();
			if (!h->retired()) break;
			// its last bank went between the look-up and the lock: it is out of the registry, look again (a new hub)
			h->mu.unlock();
			unref(h);
		}
		tl_hub = h;
	}
	// hot path: a filter that holds a slot knows its hub through the bank -- no registry lookup (and the slot keeps the hub alive)
	HubLock(MSFilter *f, Pool *p) : h(nullptr), prev(tl_hub) {
		if (p) {
			h = p->hub;
			h->ref();
			h->mu.lock();
		} else {
			for (;;) {
				h = hub_for(f, true);
				h->mu.lock();
				if (!h->retired()) break;
				h->mu.unlock();
				unref(h);
			}
		}
		tl_hub = h;
	}
	explicit HubLock(TickerHub *hub) : h(hub), prev(tl_hub) { // the caller holds a slot of `hub`: it cannot go away
		h->ref();
		h->mu.lock();
		tl_hub = h;
	}
	struct Adopt {};
	HubLock(TickerHub *hub, Adopt) : h(hub), prev(tl_hub) { // the reference was taken under the registry lock (hub_for, referenced_hubs)
		h->mu.lock();
		tl_hub = h;
	}
	bool dead() const { return h->retired(); } // the hub's last bank went before this scope got the lock: nothing to do on it
	~HubLock() {
		tl_hub = prev;
		retire_if_idle(h);
		h->mu.unlock();
		unref(h);
	}
	HubLock(const HubLock &) = delete;
	HubLock &operator=(const HubLock &) = delete;
};
#define g_hub (*tl_hub)

Pool::~Pool() {
	for (void *p : host_allocs) mi_host_free(hub->ctx, p);
	for (void *p : dev_allocs) mi_dev_free(hub->ctx, p);
	for (void *p : host_fallback) free(p);
}

void Pool::release(int slot) {
	if (slot < 0 || slot >= capacity || !owner[(size_t)slot]) return;
	MSFilter *f = owner[(size_t)slot];
	owner[(size_t)slot] = nullptr;
	free_list.push_back(slot);
	--in_use;
	TickerHub *h = hub;
	{
		std::unique_lock<std::mutex> wl(g_registry_mu);
		auto fi = g_filter_hubs.find(f);
		if (fi != g_filter_hubs.end() && --fi->second.slots <= 0) g_filter_hubs.erase(fi);
	}
	if (in_use == 0) { // last slot of the bank: its device objects and pinned buffers go
		if (h->ctx) mi_ctx_sync(h->ctx);
		h->pools.erase(std::find(h->pools.begin(), h->pools.end(), this));
		delete this;
		if (h->pools.empty() && h->pins == 0) { // and with the last bank the hub (its stream): a ticker per call must not leak one
			std::unique_lock<std::mutex> wl(g_registry_mu);
			auto it = g_hubs.find(h->ticker);
			if (it != g_hubs.end() && it->second == h) g_hubs.erase(it);
			h->life.fetch_or(TickerHub::RETIRED, std::memory_order_acq_rel); // deleted by the last HubLock scope to end
			h->flush_owner = nullptr, h->drop_backups();
		}
	}
}

// A bank of pool type P for `key` on the current hub with at least `need` free slots, created on demand with the next
// capacity of the series 16, 64, 256, 1024, 4096, 16384 (MSMI355X_SLOTS caps or fixes the first one).
template <typename P>
P *bank(const std::string &key, int need, const std::function<P *(int cap)> &make) {
	TickerHub &h = g_hub;
	int nbanks = 0;
	for (Pool *p : h.pools)
		if (p->key == key) {
			++nbanks;
			if (!p->failed && (int)p->free_list.size() >= need) return static_cast<P *>(p);
		}
	static const int first = [] {
		const char *e = getenv("MSMI355X_SLOTS");
		const int v = e ? atoi(e) : 0;
		return v > 0 ? v : 16;
	}();
	long long cap = first;
	for (int i = 0; i < nbanks && cap < 16384; ++i) cap *= 4;
	cap = std::max<long long>(cap, need);
	P *p = make((int)cap);
	p->key = key;
	if (p->failed) { // the device refused: the caller falls back (pass-through / drop + late event)
		delete p;
		return nullptr;
	}
	h.pools.push_back(p);
	return p;
}

// Pool constructors call this first: wires the bank to the current hub so that its allocations use the hub's context
struct Building {
	Pool *prev;
	explicit Building(Pool *p, int cap) : prev(tl_building) {
		p->hub = tl_hub;
		p->init_slots(cap);
		tl_building = p;
		if (!p->hub->ensure_ctx()) p->failed = true;
	}
	~Building() { tl_building = prev; }
};

void note_slot(MSFilter *f) { // the filter holds one more slot on the current hub (so a detached filter still finds it)
	std::unique_lock<std::mutex> wl(g_registry_mu);
	FilterRef &r = g_filter_hubs[f];
	r.hub = tl_hub;
	r.slots++;
}

bool is_ours(const MSFilterDesc *d); // one of the descriptors below

// every ms_queue_put of the facades goes through here (macro below): during the flush task it also notes the reader
inline void emit_to(MSQueue *q, mblk_t *m) {
	ms_queue_put(q, m);
	TickerHub *h = tl_hub;
	if (!h || !h->in_flush) return;
	MSFilter *g = q->next.filter;
	// (the reader may belong to a graph that is being attached right now, on the application's thread: its leg joined this hub's batch when the last
	// facade was preprocessed, ms_ticker_attach is still setting the remaining filters' tickers -- msticker.c:163-166)
	if (!g || !is_ours(g->desc) || __atomic_load_n(&g->ticker, __ATOMIC_RELAXED) != h->ticker || !h->ticker) return;
	if (h->scope && !h->scope->count(g)) return; // (a detaching graph's flush runs nobody else's process())
	std::vector<MSFilter *> &v = (g->desc->flags & MS_FILTER_IS_PUMP) ? h->touched_pumps : h->touched;
	if (std::find(v.begin(), v.end(), g) == v.end()) v.push_back(g);
}

bool inputs_waiting(MSFilter *g) {
	for (int i = 0; i < g->desc->ninputs; ++i)
		if (g->inputs[i] && !ms_queue_empty(g->inputs[i])) return true;
	return false;
}

// a pump facade (mixer, PLC, channel adapter) may run ahead of the graph only if everything it reads comes from facades
// of this plugin: a CPU filter's block for this tick has not been produced yet when the tasks run
bool all_inputs_ours(MSFilter *g) {
	for (int i = 0; i < g->desc->ninputs; ++i)
		if (g->inputs[i] && (!g->inputs[i]->prev.filter || !is_ours(g->inputs[i]->prev.filter->desc))) return false;
	return true;
}

// pump facades call this first in process(): true = the flush task already ran them for this tick
bool already_ran_this_tick(MSFilter *f) {
	TickerHub &h = g_hub;
	if (h.in_flush || !f->ticker) return false;
	auto it = h.pumped.find(f);
	return it != h.pumped.end() && it->second == ticker_now(f->ticker);
}

void deliver_fused_in_scope(TickerHub &h);  // leg_chain.inl
void deliver_server_in_scope(TickerHub &h); // server_leg.inl
void deliver_recv_in_scope(TickerHub &h);   // recv_leg.inl
void flush_hub(TickerHub &h) {
	h.in_flush = true;
	h.touched.clear();
	h.touched_pumps.clear();
	if (h.scope) deliver_server_in_scope(h);
	if (h.scope) deliver_recv_in_scope(h);
	if (h.scope) deliver_fused_in_scope(h); // fused conferences / legs of the graph: the launches already out are waited for, their results handed on
	auto takes_part = [&](Pool *p) { return !h.scope || p->scoped(); };
	for (int round = 0; round < 16; ++round) { // chains deeper than this finish on the next tick
		// banks flushed in creation order; a facade may stage into any bank while another one emits
		// every bank enqueues on the hub's stream, ONE wait, then every bank hands its results on
		bool any = false;
		const size_t npools = h.pools.size(); // (a bank created while the results are emitted is flushed in the next round)
		for (size_t i = 0; i < npools; ++i)
			if (!h.pools[i]->failed && takes_part(h.pools[i])) any |= h.pools[i]->enqueue();
		if (any && h.ctx && mi_ctx_sync(h.ctx) != MI_OK) {
			mi_failed("mi_ctx_sync");
			for (size_t i = 0; i < npools; ++i)
				if (takes_part(h.pools[i])) h.pools[i]->failed = true; // nothing of this flush can be trusted
		}
		++h.flushes;
		for (size_t i = 0; i < npools && i < h.pools.size(); ++i) {
			Pool *p = h.pools[i];
			if (!takes_part(p)) continue;
			p->finish(); // (a failed bank still settles its bookkeeping: its filters pass their blocks on or drop them)
			p->emit_all();
			p->emitted();
		}
		std::vector<MSFilter *> run;
		run.swap(h.touched);
		if (run.empty()) { // nothing but pumps left: they go last, once everything that feeds them has arrived
			for (MSFilter *g : h.touched_pumps)
				if (all_inputs_ours(g) && g->ticker) {
					h.pumped[g] = ticker_now(g->ticker);
					run.push_back(g);
				}
			h.touched_pumps.clear();
			if (run.empty()) break;
			for (MSFilter *g : run) g->desc->process(g);
			continue;
		}
		for (MSFilter *g : run) // call_process msticker.c:244-259: while there is input (a filter that staged stops)
			for (int n = 0; n < 64 && inputs_waiting(g); ++n) g->desc->process(g);
	}
	for (Pool *p : h.pools)
		if (takes_part(p)) p->flushed();
	h.in_flush = false;
}

// Runs on the ticker thread at the start of the next tick, before any process() (msticker.c:301-312):
// one launch per staged pool, then the results go straight into the owners' output queues, so the
// downstream filters see them in this tick's graph run even if the owner itself gets no new input.
void flush_task(MSFilter *f) {
	HubLock lk(f);
	g_hub.flush_owner = nullptr, g_hub.drop_backups();
	const uint32_t tick = g_hub.ticker ? g_hub.ticker->ticks : 0;
	if (g_hub.ticker && g_hub.flush_done_tick == tick) return; // (the other of the two tasks ran at the head of this tick)
	g_hub.flush_done_tick = tick;
	flush_hub(g_hub);
}

// called by a filter that staged work this tick (hub locked)
void request_flush(MSFilter *f) {
	++g_hub.stage_seq;
	if (g_hub.in_flush) return; // staged from inside the flush task (chain linking): the task's loop gets to it
	// A task posted in an earlier tick runs at the head of this one (msticker.c:484-485: run_tasks, then run_graphs).  If it is still owed, its
	// filter left the ticker in between: ms_ticker_detach drops a detaching filter's tasks (msticker.c:187-190,:314-324) on the APPLICATION's
	// thread, and a walk may run before that filter's postprocess tells this hub (facade_detached) -- post a new one
	if ((g_hub.flush_owner || g_hub.backup_waiting()) && f->ticker && g_hub.flush_posted_tick != f->ticker->ticks) g_hub.flush_owner = nullptr, g_hub.drop_backups();
	if (!g_hub.flush_owner && !g_hub.backup_waiting()) {
		g_hub.flush_owner = f;
		g_hub.flush_asks = 1;
		g_hub.flush_posted_tick = f->ticker ? f->ticker->ticks : 0;
		ms_filter_postpone_task(f, flush_task);
		return;
	}
	const uint32_t n = ++g_hub.flush_asks;
	const int k = n == 2 ? 0 : n == 4 ? 1 : n == 8 ? 2 : n == 16 ? 3 : n == 64 ? 4 : n == 256 ? 5 : n == 1024 ? 6 : n == 4096 ? 7 : -1;
	if (k >= 0 && !g_hub.flush_backup[k] && f != g_hub.flush_owner && f->ticker) {
		g_hub.flush_backup[k] = f;
		ms_filter_postpone_task(f, flush_task);
	}
}

// Every facade's postprocess ends here: the ticker drops a detached filter's postponed tasks (msticker.c:187-190,
// :314-324), so if this filter owned the pending flush nobody will run it -- the next request must post a new one.
// ... and it BEGINS by delivering the tick in flight: the first postprocess of a detaching graph flushes the graph's own staged
// work through its chain (TickerHub::scope), so that what the reference's synchronous filters would have handed on in the last walk
// is handed on before any of them drops its queues or state.
void graph_of(MSFilter *f, std::unordered_set<MSFilter *> &out) { // every filter linked to f, whoever made it (ms_filter_find_neighbours)
	std::vector<MSFilter *> todo{f};
	out.insert(f);
	while (!todo.empty()) {
		MSFilter *g = todo.back();
		todo.pop_back();
		for (int i = 0; i < g->desc->ninputs; ++i)
			if (g->inputs[i] && g->inputs[i]->prev.filter && out.insert(g->inputs[i]->prev.filter).second) todo.push_back(g->inputs[i]->prev.filter);
		for (int i = 0; i < g->desc->noutputs; ++i)
			if (g->outputs[i] && g->outputs[i]->next.filter && out.insert(g->outputs[i]->next.filter).second) todo.push_back(g->outputs[i]->next.filter);
	}
}
void facade_detached(MSFilter *f) {
	TickerHub *h = hub_for(f, false);
	if (!h) return;
	HubLock lk(h, HubLock::Adopt{});
	if (lk.dead()) return;
	if (f->ticker && h->ticker == f->ticker && !h->in_flush && !(h->drained_seq == h->stage_seq && h->drained.count(f))) {
		std::unordered_set<MSFilter *> graph;
		graph_of(f, graph);
		h->scope = &graph;
		flush_hub(*h);
		h->scope = nullptr;
		h->drained.swap(graph);
		h->drained_seq = h->stage_seq;
	}
	if (h->flush_owner == f) h->flush_owner = nullptr; // (its task went with it: the walk's next request posts another -- or the second task is still there)
	for (MSFilter *&bk : h->flush_backup)
		if (bk == f) bk = nullptr;
	h->pumped.erase(f);
}
void generic_postprocess(MSFilter *f) { facade_detached(f); }

// MSMI355X_ZERO_COPY (default on): banks whose launches can take their rows from pinned host memory directly hand those over
// instead of staging through device buffers (the fused leg bank, MSSpeexEC's bank); 0 = copy launches around the kernels
bool zero_copy_rows() {
	static const bool v = [] {
		const char *e = getenv("MSMI355X_ZERO_COPY");
		return !(e && e[0] == '0');
	}();
	return v;
}

#define ms_queue_put(q, m) emit_to((q), (m)) /* the facades' queue puts, see emit_to */

// the fused call-leg chain (filters/leg_chain.inl): what its four facades need to know of it
struct FusedLeg;
struct LegBank;
struct VolumeData;
struct ResampleData;
struct SpeexECState;
void leg_stage_mic(MSFilter *f, ResampleData *d);
void leg_take_far_end(MSFilter *f, SpeexECState *s);
void leg_stage_mic_ec(MSFilter *f, SpeexECState *s);
MSFilter *leg_find_mixer_ec(MSFilter *ec);
bool leg_try_fuse_plain_ec(MSFilter *ec);
bool leg_has_resampler(FusedLeg *leg);
void leg_head_done(FusedLeg *leg);
bool leg_frames_chunks(FusedLeg *leg);
void leg_stage_peer(MSFilter *vol, VolumeData *d); // an echo limiter's peer metered beside a fused leg
void copy_payload(const mblk_t *m, uint8_t *dst); // codec.inl
MSFilter *leg_find_mixer(MSFilter *rs);
bool conf_try_fuse(MSFilter *mixer);
void conf_unfuse(MSFilter *mixer, bool keep_running);
void leg_disqualify(FusedLeg *leg);
void leg_forwarder_changed(MSFilter *rs); // a forwarding MSResample between a fused leg's MSVolume and its mixer pin is given other rates
void leg_release(FusedLeg *leg, bool keep_running); // the leg (and, in a conference, everybody with it) leaves its fused batch
bool leg_try_fuse_plain(MSFilter *rs);
bool leg_wants_out(FusedLeg *leg);
Pool *leg_pool(FusedLeg *leg);
// a conference server's members as one batch (filters/server_leg.inl): what MSVolume, the mixer and the G.711 encoders need to know of it
struct ServerLeg;
struct ServerBank;
struct VolumeData;
void server_stage(MSFilter *vol, VolumeData *d);
void server_release(ServerLeg *leg, bool keep_running);
void server_disqualify(ServerLeg *leg);
bool server_wants_out(ServerLeg *leg);
Pool *server_pool(ServerLeg *leg);
Pool *server_pool_of(ServerBank *b);
void server_conf_walked(ServerBank *b, int c);
void server_unfuse(MSFilter *mixer, bool keep_running);

// the receiving side of an AudioStream as one batch (filters/recv_leg.inl): what the decoders, MSGenericPLC and MSAudioFlowControl need to know of it
struct RecvLeg;
struct MapFilter;
struct PlcFilter;
void recv_chain_preprocessed(MSFilter *member);
void graph_preprocessed(MSFilter *f); // attach.inl: every facade's preprocess ends here
bool graph_ready(MSFilter *f);        // ... after asking, without any lock, whether it is the graph's LAST facade to be preprocessed
void generic_preprocess(MSFilter *f);
void recv_release(RecvLeg *leg, bool keep_running);
void recv_disqualify(RecvLeg *leg);
bool recv_wants_out(RecvLeg *leg);
bool recv_idle(RecvLeg *leg);
Pool *recv_pool(RecvLeg *leg);
void recv_stage_codes(MSFilter *f, MapFilter *d);
void recv_plc_walk(MSFilter *f, PlcFilter *d);
void recv_flow_drop(RecvLeg *leg, uint32_t drop, uint32_t total);
void recv_flow_config(RecvLeg *leg, const MSAudioFlowControlConfig *cfg);

#include "filters/resample.inl"
#include "filters/volume.inl"
#include "filters/equalizer.inl"
#include "filters/mixer.inl"
#include "filters/echo_canceller.inl"
#include "filters/codec.inl"
#include "filters/leg_chain.inl"
#include "filters/video.inl"
#include "filters/server_leg.inl"
#include "filters/flow_control.inl"
#include "filters/generic_plc.inl"
#include "filters/recv_leg.inl"
#include "filters/attach.inl"

} // namespace

extern "C" {

// Descriptors: same ids, names, pin counts and flags as the reference's, plus
// MS_FILTER_IS_HW_ACCELERATED (msfilter.h:142).  Writable statics: the factory mutates flags.
MSFilterDesc ms_mi355x_resample_desc = {MS_RESAMPLE_ID, "MSResample", "Audio resampler (MI355X batch)", MS_FILTER_OTHER,
                                        NULL, 1, 1, resample_init, generic_preprocess, resample_process, resample_postprocess, resample_uninit,
                                        resample_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_audio_mixer_desc = {MS_AUDIO_MIXER_ID, "MSAudioMixer",
                                           "A filter that mixes down 16 bit sample audio streams (MI355X batch)",
                                           MS_FILTER_OTHER, NULL, MIXER_MAX_CHANNELS, MIXER_MAX_CHANNELS, mixer_init,
                                           mixer_preprocess, mixer_process, mixer_postprocess, mixer_uninit,
                                           mixer_methods, MS_FILTER_IS_PUMP | MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_volume_desc = {MS_VOLUME_ID, "MSVolume", "A filter that controls and measure sound volume (MI355X batch)",
                                      MS_FILTER_OTHER, NULL, 1, 1, volume_init, volume_preprocess, volume_process, volume_postprocess,
                                      volume_uninit, volume_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_equalizer_desc = {MS_EQUALIZER_ID, "MSEqualizer", "Parametric sound equalizer (MI355X batch)",
                                         MS_FILTER_OTHER, NULL, 1, 1, equalizer_init, equalizer_preprocess, equalizer_process, equalizer_postprocess,
                                         equalizer_uninit, equalizer_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_speex_ec_desc = {MS_SPEEX_EC_ID, "MSSpeexEC", "Echo canceller, MDF + post-filter (MI355X batch)",
                                        MS_FILTER_OTHER, NULL, 2, 2, ec_init, ec_preprocess, ec_process, ec_postprocess,
                                        ec_uninit, ec_methods, MS_FILTER_IS_HW_ACCELERATED};

// The factory's default echo-canceller NAME is "MSWebRTCAEC" (src/base/msfactory.c:245): audio_stream_new_with_sessions looks
// that name up first and only falls back to MS_SPEEX_EC_ID when no such filter exists (src/voip/audiostream.c:2128-2158).
// Where the mswebrtc plugin is installed an AudioStream therefore never reaches MS_SPEEX_EC_ID.  With
// MSMI355X_CLAIM_WEBRTC_AEC=1 this descriptor is registered as well (prepended: it wins the look-up by name) so that such
// streams land on the GPU canceller.  It is NOT AEC3: the same speex-class MDF canceller + post-filter under that name,
// and the text says so.
MSFilterDesc ms_mi355x_webrtc_aec_name_desc = {MS_FILTER_PLUGIN_ID, "MSWebRTCAEC",
                                               "NOT WebRTC AEC3: speex-class MDF echo canceller + post-filter (MI355X batch) answering to this name",
                                               MS_FILTER_OTHER, NULL, 2, 2, ec_init, ec_preprocess, ec_process, ec_postprocess,
                                               ec_uninit, ec_methods, MS_FILTER_IS_HW_ACCELERATED};

MSFilterDesc ms_mi355x_size_conv_desc = {MS_SIZE_CONV_ID, "MSSizeConv", "A video size converter (MI355X batch)", MS_FILTER_OTHER,
                                         NULL, 1, 1, size_conv_init, generic_preprocess, size_conv_process, size_conv_postprocess,
                                         size_conv_uninit, sizeconv_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_pix_conv_desc = {MS_PIX_CONV_ID, "MSPixConv", "A pixel format converter (MI355X batch)", MS_FILTER_OTHER,
                                        NULL, 1, 1, pixconv_init, generic_preprocess, pixconv_process, generic_postprocess, pixconv_uninit,
                                        pixconv_methods, MS_FILTER_IS_HW_ACCELERATED};
MSScalerDesc ms_mi355x_scaler_desc = {sd_create, sd_process, sd_free};

// the stages either side of the path
MSFilterDesc ms_mi355x_alaw_dec_desc = {MS_ALAW_DEC_ID, "MSAlawDec", "ITU-G.711 alaw decoder (MI355X batch)", MS_FILTER_DECODER, "pcma", 1, 1,
                                        g711_dec_init_a, g711_dec_preprocess, g711_dec_process, g711_dec_postprocess, map_uninit, g711_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_ulaw_dec_desc = {MS_ULAW_DEC_ID, "MSUlawDec", "ITU-G.711 ulaw decoder (MI355X batch)", MS_FILTER_DECODER, "pcmu", 1, 1,
                                        g711_dec_init_u, g711_dec_preprocess, g711_dec_process, g711_dec_postprocess, map_uninit, g711_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_alaw_enc_desc = {MS_ALAW_ENC_ID, "MSAlawEnc", "ITU-G.711 alaw encoder (MI355X batch)", MS_FILTER_ENCODER, "pcma", 1, 1,
                                        g711_enc_init_a, generic_preprocess, g711_enc_process, g711_enc_postprocess, map_uninit, g711_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_ulaw_enc_desc = {MS_ULAW_ENC_ID, "MSUlawEnc", "ITU-G.711 ulaw encoder (MI355X batch)", MS_FILTER_ENCODER, "pcmu", 1, 1,
                                        g711_enc_init_u, generic_preprocess, g711_enc_process, g711_enc_postprocess, map_uninit, g711_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_l16_enc_desc = {MS_L16_ENC_ID, "MSL16Enc", "L16 dummy encoder (MI355X batch)", MS_FILTER_ENCODER, "L16", 1, 1,
                                       l16_enc_init, l16_enc_preprocess, l16_enc_process, generic_postprocess, map_uninit, l16_enc_methods, MS_FILTER_IS_HW_ACCELERATED};
MSFilterDesc ms_mi355x_l16_dec_desc = {MS_L16_DEC_ID, "MSL16Dec", "L16 dummy decoder (MI355X batch)", MS_FILTER_DECODER, "L16", 1, 1,
                                       l16_dec_init, generic_preprocess, l16_dec_process, generic_postprocess, map_uninit, l16_dec_methods, MS_FILTER_IS_HW_ACCELERATED};
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

} // extern "C"
namespace {
bool is_ours(const MSFilterDesc *d) {
	for (const MSFilterDesc *o : {&ms_mi355x_resample_desc, &ms_mi355x_audio_mixer_desc, &ms_mi355x_volume_desc, &ms_mi355x_equalizer_desc,
	                              &ms_mi355x_speex_ec_desc, &ms_mi355x_webrtc_aec_name_desc, &ms_mi355x_size_conv_desc, &ms_mi355x_pix_conv_desc, &ms_mi355x_alaw_dec_desc,
	                              &ms_mi355x_ulaw_dec_desc, &ms_mi355x_alaw_enc_desc, &ms_mi355x_ulaw_enc_desc, &ms_mi355x_l16_enc_desc,
	                              &ms_mi355x_l16_dec_desc, &ms_mi355x_channel_adapter_desc, &ms_mi355x_audio_flow_control_desc,
	                              &ms_mi355x_generic_plc_desc})
		if (o == d) return true;
	return false;
}
} // namespace
extern "C" {

void libmsmi355xfilters_init(MSFactory *factory) {
	// No usable HIP device: register NOTHING -- the reference's own CPU filters stay in charge (src/base/msfactory.c:281:
	// registration prepends, so not registering is how a plugin steps aside).  There is no CPU path in this library.
	parse_devices();
	mi_ctx *probe = nullptr;
	if (mi_ctx_create(g_devices[0], nullptr, &probe) != MI_OK) {
		// MSMI355X_REGISTER_WITHOUT_DEVICE=1: register all the same (descriptor inspection on a box without a GPU; every
		// filter then passes through or drops, see the bank failure path)
		if (!getenv("MSMI355X_REGISTER_WITHOUT_DEVICE")) {
			ms_error("libmsmi355xfilters: no HIP device %d (%s): the MI355X filters are NOT registered", g_devices[0], mi_last_error());
			return;
		}
	} else {
		// every device the hubs may land on gets the kernels' code objects now: the runtime would otherwise load them under the first
		// launches of the ticker threads' first ticks (include/msmi355x.h: mi_warmup)
		if (mi_warmup(probe) != MI_OK) ms_warning("libmsmi355xfilters: mi_warmup on device %d: %s", g_devices[0], mi_last_error());
		mi_ctx_destroy(probe);
		for (size_t i = 1; i < g_devices.size(); ++i) {
			mi_ctx *c = nullptr;
			if (mi_ctx_create(g_devices[i], nullptr, &c) != MI_OK) continue;
			if (mi_warmup(c) != MI_OK) ms_warning("libmsmi355xfilters: mi_warmup on device %d: %s", g_devices[i], mi_last_error());
			mi_ctx_destroy(c);
		}
	}
	ms_factory_register_filter(factory, &ms_mi355x_resample_desc);
	ms_factory_register_filter(factory, &ms_mi355x_audio_mixer_desc);
	ms_factory_register_filter(factory, &ms_mi355x_volume_desc);
	ms_factory_register_filter(factory, &ms_mi355x_equalizer_desc);
	ms_factory_register_filter(factory, &ms_mi355x_speex_ec_desc);
	if (const char *claim = getenv("MSMI355X_CLAIM_WEBRTC_AEC"))
		if (atoi(claim) != 0) {
			ms_factory_register_filter(factory, &ms_mi355x_webrtc_aec_name_desc);
			ms_warning("libmsmi355xfilters: answering to the name MSWebRTCAEC with the speex-class MDF canceller (this is not AEC3)");
		}
	ms_factory_register_filter(factory, &ms_mi355x_size_conv_desc);
	ms_factory_register_filter(factory, &ms_mi355x_pix_conv_desc);
	for (MSFilterDesc *d : {&ms_mi355x_alaw_dec_desc, &ms_mi355x_ulaw_dec_desc, &ms_mi355x_alaw_enc_desc, &ms_mi355x_ulaw_enc_desc,
	                        &ms_mi355x_l16_enc_desc, &ms_mi355x_l16_dec_desc, &ms_mi355x_channel_adapter_desc, &ms_mi355x_audio_flow_control_desc,
	                        &ms_mi355x_generic_plc_desc})
		ms_factory_register_filter(factory, d);
	ms_video_set_scaler_impl(&ms_mi355x_scaler_desc); // msvideo.c:719-721: the reference's own video filters follow
	ms_message("libmsmi355xfilters: MI355X batched filters registered (ABI %d, %d device(s))", mi_abi_version(), (int)g_devices.size());
}

// every hub in the registry, each with a reference taken under the registry lock (to be adopted by a HubLock)
static std::vector<TickerHub *> referenced_hubs() {
	std::vector<TickerHub *> hubs;
	std::lock_guard<std::mutex> rl(g_registry_mu);
	for (auto &kv : g_hubs) {
		kv.second->ref();
		hubs.push_back(kv.second);
	}
	return hubs;
}

// every hub's staged work, now (tests; an application that wants the last tick's results before tearing a graph down)
void ms_mi355x_flush(void) {
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead()) continue;
		h->flush_owner = nullptr, h->drop_backups();
		flush_hub(*h);
	}
}

// blocks dropped or passed through unprocessed because a kernel-library call failed (0 in a healthy process)
unsigned long long ms_mi355x_late_events(void) { return (unsigned long long)g_late_events.load(); }
// hubs (tickers with live banks) and banks alive: what a leak check looks at
void ms_mi355x_runtime_stats(int *hubs, int *banks, int *slots_in_use) {
	int nh = 0, nb = 0, ns = 0;
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead()) continue;
		++nh;
		for (Pool *p : h->pools) ++nb, ns += p->in_use;
	}
	if (hubs) *hubs = nh;
	if (banks) *banks = nb;
	if (slots_in_use) *slots_in_use = ns;
}

// fused call-leg batches (filters/leg_chain.inl): conferences and legs living in them, kernel launches they have enqueued
// and hub flush rounds (one synchronisation each) so far -- what bench.py's plugin_path reports per tick
void ms_mi355x_fused_stats(int *conferences, int *legs, unsigned long long *launches, unsigned long long *flush_rounds) {
	int nc = 0, nl = 0;
	unsigned long long la = 0, fr = 0;
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead()) continue;
		fr += h->flushes;
		for (Pool *p : h->pools)
			if (p->key.compare(0, 3, "leg") == 0) { // "leg:" conferences, "legp:" legs without a mixer
				LegBank *b = static_cast<LegBank *>(p);
				nc += b->plain ? 0 : b->in_use;
				la += b->launches;
				for (FusedLeg *l : b->legs) nl += l != nullptr;
			} else if (p->key.compare(0, 4, "srv:") == 0) { // a server's conferences of remote members (server_leg.inl)
				ServerBank *b = static_cast<ServerBank *>(p);
				nc += b->in_use;
				la += b->launches;
				for (ServerLeg *l : b->legs) nl += l != nullptr;
			} else if (p->key.compare(0, 4, "rcv:") == 0) { // streams' receiving sides (recv_leg.inl): their launches count, their streams are ms_mi355x_recv_stats'
				la += static_cast<RecvBank *>(p)->launches;
			}
	}
	if (conferences) *conferences = nc;
	if (legs) *legs = nl;
	if (launches) *launches = la;
	if (flush_rounds) *flush_rounds = fr;
}

// streams whose receiving side (decoder -> MSGenericPLC -> MSAudioFlowControl) lives in a fused batch (filters/recv_leg.inl)
int ms_mi355x_recv_stats(void) {
	int n = 0;
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead()) continue;
		for (Pool *p : h->pools)
			if (p->key.compare(0, 4, "rcv:") == 0)
				for (RecvLeg *l : static_cast<RecvBank *>(p)->legs) n += l != nullptr;
	}
	return n;
}

// 1: the facade's work runs in a batch it shares with its neighbours; 0: in a bank of its own (or it has not run); -1: not a filter of this plugin
int ms_mi355x_filter_in_batch(MSFilter *f) {
	if (!f || !f->desc || !is_ours(f->desc)) return -1;
	HubLock lk(f);
	return facade_in_batch(f) ? 1 : 0;
}

// the device of every hub that has opened a context (tests: tickers spread over MSMI355X_DEVICES); returns their number
int ms_mi355x_hub_devices(int *devices, int cap) {
	int n = 0;
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead() || !h->ctx) continue;
		if (devices && n < cap) devices[n] = h->device;
		++n;
	}
	return n;
}

void ms_mi355x_shutdown(void) {
	for (TickerHub *h : referenced_hubs()) {
		HubLock lk(h, HubLock::Adopt{});
		if (lk.dead()) continue;
		if (h->ctx) mi_ctx_sync(h->ctx);
	}
}

} // extern "C"
