// mi_double.cpp -- TEST INFRASTRUCTURE: a host-memory double of libmsmi355x.so's C ABI (include/msmi355x.h).
//
// Built as tests/host/<variant>/libmsmi355x.so -- the product's soname, in a directory of its own -- so that a copy of the
// PLUGIN compiled from the product's sources (mediastreamer2_amd/host/filters.cpp) links against it there and its threaded
// host runtime (ticker hubs, banks, the registry, chain linking) can run on a box without a GPU, under ThreadSanitizer and
// AddressSanitizer (GPU sanitizers are not available on the pool).  Nothing under mediastreamer2_amd/ builds, links or loads
// this file; the product has no CPU path.
//
// "Device" memory is malloc'd, copies are memcpy, every entry point runs synchronously on the calling thread.  Bodies are
// trivial but touch exactly the byte ranges the kernels would (so ASAN checks the host's buffer sizing): the resampler
// holds samples, the canceller passes the microphone through, volume / equalizer leave the block alone, the scaler paints
// grey.  Two pieces are the real definition because tests rely on them: the conference mixer (int32 sum, own contribution
// removed, +-32767: audiomixer.c:33-51,:301-344) with its split form, and mi_exchange as an in-process rendezvous between
// threads (the RCCL all-reduce's contract: every rank's buffer ends up holding the sum).
// MSMI355X_DOUBLE_DEVICES = number of devices it pretends to have (default 1).
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/msmi355x.h"

static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) {
	snprintf(g_err, sizeof g_err, "mi_double: %s", msg);
	return code;
}
#define ARG(c)                                          \
	do {                                                \
		if (!(c)) return fail(MI_EINVAL, "bad argument: " #c); \
	} while (0)

struct mi_ctx {
	int device;
};
struct mi_graph {
	int unused;
};
struct mi_resampler {
	mi_ctx *ctx;
	int n;
	uint32_t in_rate, out_rate;
};
struct mi_mixer {
	mi_ctx *ctx;
	int nconf, mm, ns;
	std::vector<uint8_t> flags;
	std::vector<float> gain;
};
struct mi_volume {
	mi_ctx *ctx;
	int n, rate;
	std::vector<mi_volume_params> params;
	std::vector<mi_volume_state> state;
	std::vector<float> mx;
	mi_volume *ext = nullptr; // mi_volume_set_peer_batch
};
struct mi_equalizer {
	mi_ctx *ctx;
	int n, rate, nfft;
	std::vector<std::vector<float>> gains, taps;
	std::vector<int> active;
	std::vector<int16_t> last; // the "history": the stream's last input sample (enough to make a lost history audible)
};
struct mi_aec {
	mi_ctx *ctx;
	int n, rate, F, flen;
	std::vector<int> frames;
};
struct mi_scaler {
	mi_ctx *ctx;
	int sw, sh, dw, dh, fmt;
};
struct mi_pixconv {
	mi_ctx *ctx;
	int w, h, fmt, flip;
};
struct mi_fifo {
	mi_ctx *ctx;
	int n, cap;
	std::vector<std::vector<int16_t>> q;
	int overflow;
};
struct mi_flowctl {
	mi_ctx *ctx;
	int n, max_block;
	std::vector<uint32_t> target, total;
};
struct mi_plc {
	mi_ctx *ctx;
	int n, rate, max_block;
};
struct mi_session {
	int unused;
};

static int device_count() {
	const char *e = getenv("MSMI355X_DOUBLE_DEVICES");
	const int n = e ? atoi(e) : 1;
	return n < 0 ? 0 : n;
}

// ---- exchange: rendezvous of `nranks` threads on a 128-byte id
namespace {
struct Meeting {
	std::mutex mu;
	std::condition_variable cv;
	int nranks = 0, joined = 0, arrived = 0, generation = 0;
	std::vector<int32_t *> bufs;
	std::vector<int64_t> sum;
};
std::mutex g_meet_mu;
std::map<std::string, Meeting *> g_meetings;
} // namespace
struct mi_exchange {
	mi_ctx *ctx;
	Meeting *m;
	int nranks, rank;
};

extern "C" {

int mi_abi_version(void) { return MSMI355X_ABI_VERSION; }
const char *mi_last_error(void) { return g_err; }
int mi_device_count(void) {
	const int n = device_count();
	if (n == 0) fail(0, "no device (MSMI355X_DOUBLE_DEVICES=0); libmsmi355x has no CPU fallback");
	return n;
}

int mi_ctx_create(int device, void *hip_stream, mi_ctx **out) {
	ARG(out);
	*out = nullptr;
	if (device_count() == 0) return fail(MI_ENODEV, "no HIP device available; libmsmi355x has no CPU fallback");
	ARG(device >= 0 && device < device_count());
	*out = new mi_ctx{device};
	return MI_OK;
}
void mi_ctx_destroy(mi_ctx *c) { delete c; }
int mi_warmup(mi_ctx *c) {
	ARG(c);
	return MI_OK;
}
int mi_ctx_sync(mi_ctx *c) {
	ARG(c);
	return MI_OK;
}
void *mi_ctx_stream(mi_ctx *c) { return c; }
int mi_ctx_device(mi_ctx *c) { return c ? c->device : -1; }
int mi_ctx_props(mi_ctx *c, int *cu, size_t *hbm, char *name, int cap) {
	ARG(c);
	if (cu) *cu = 0;
	if (hbm) *hbm = 0;
	if (name && cap > 0) snprintf(name, (size_t)cap, "mi_double device %d", c->device);
	return MI_OK;
}
void *mi_dev_alloc(mi_ctx *c, size_t bytes) { return c ? malloc(bytes ? bytes : 1) : nullptr; }
void mi_dev_free(mi_ctx *c, void *p) { free(p); }
void *mi_host_alloc(mi_ctx *c, size_t bytes) { return c ? malloc(bytes ? bytes : 1) : nullptr; }
void mi_host_free(mi_ctx *c, void *p) { free(p); }
int mi_copy_h2d(mi_ctx *c, void *d, const void *h, size_t n) {
	ARG(c && d && h);
	memcpy(d, h, n);
	return MI_OK;
}
int mi_copy_d2h(mi_ctx *c, void *h, const void *d, size_t n) {
	ARG(c && d && h);
	memcpy(h, d, n);
	return MI_OK;
}
int mi_copy_h2d_pinned(mi_ctx *c, void *d, const void *h, size_t n) { return mi_copy_h2d(c, d, h, n); }
int mi_copy_d2h_pinned(mi_ctx *c, void *h, const void *d, size_t n) { return mi_copy_d2h(c, h, d, n); }
int mi_memset(mi_ctx *c, void *d, int v, size_t n) {
	ARG(c && d);
	memset(d, v, n);
	return MI_OK;
}
int mi_ctx_capture_begin(mi_ctx *c) { return fail(MI_ENOTSUP, "no graphs in the double"); }
int mi_ctx_capture_end(mi_ctx *c, mi_graph **out) { return fail(MI_ENOTSUP, "no graphs in the double"); }
int mi_graph_launch(mi_graph *g) { return fail(MI_ENOTSUP, "no graphs in the double"); }
void mi_graph_destroy(mi_graph *g) {}
int mi_timer_start(mi_ctx *c) { return MI_OK; }
int mi_timer_stop(mi_ctx *c, float *ms) {
	if (ms) *ms = 0;
	return MI_OK;
}

// ---- resampler: sample and hold
int mi_resampler_create(mi_ctx *ctx, int n, uint32_t in_rate, uint32_t out_rate, int quality, mi_resampler **out) {
	ARG(ctx && out && n > 0 && in_rate && out_rate);
	*out = new mi_resampler{ctx, n, in_rate, out_rate};
	return MI_OK;
}
void mi_resampler_destroy(mi_resampler *r) { delete r; }
// (the double's resampler repeats samples: it has no history -- a state of eight zero bytes)
int mi_resampler_state_bytes(const mi_resampler *r) { return r ? 8 : MI_EINVAL; }
int mi_resampler_get_state(mi_resampler *r, int stream, void *h_state, size_t cap) {
	ARG(r && h_state && stream >= 0 && stream < r->n && cap >= 8);
	memset(h_state, 0, 8);
	return MI_OK;
}
int mi_resampler_set_state(mi_resampler *r, int stream, const void *h_state, size_t bytes) {
	ARG(r && h_state && stream >= 0 && stream < r->n && bytes == 8);
	return MI_OK;
}
int mi_resampler_get_states(mi_resampler *r, int first, int count, void *h, size_t cap) {
	ARG(r && h && first >= 0 && count >= 0 && first + count <= r->n && cap >= 8 * (size_t)count);
	memset(h, 0, 8 * (size_t)count);
	return MI_OK;
}
int mi_resampler_set_states(mi_resampler *r, int first, int count, const void *h, size_t bytes) {
	ARG(r && h && first >= 0 && count >= 0 && first + count <= r->n && bytes == 8 * (size_t)count);
	return MI_OK;
}
int mi_resampler_reset(mi_resampler *r, int first, int count) {
	ARG(r && first >= 0 && count >= 0 && first + count <= r->n);
	return MI_OK;
}
int mi_resampler_out_capacity(const mi_resampler *r, int in_len) {
	if (!r || in_len < 0) return MI_EINVAL;
	return (int)((((uint32_t)in_len * r->out_rate) / r->in_rate) + 1);
}
int mi_resampler_info(const mi_resampler *r, int *fl, int *den, int *num, int *direct) {
	ARG(r);
	if (fl) *fl = 48;
	if (den) *den = 1;
	if (num) *num = 1;
	if (direct) *direct = 1;
	return MI_OK;
}
int mi_resampler_get_table(const mi_resampler *r, float *dst, int cap) { return 0; }
int mi_resampler_process_masked(mi_resampler *r, const int16_t *in, int in_len, int in_stride, int16_t *out, int out_stride,
                                int32_t *out_len, const uint8_t *run) {
	ARG(r && in && out && in_len >= 0 && in_stride >= in_len && out_stride >= mi_resampler_out_capacity(r, in_len) - 1);
	const int n_out = (int)(((uint64_t)in_len * r->out_rate) / r->in_rate);
	for (int s = 0; s < r->n; ++s) {
		if (run && !run[s]) continue;
		for (int i = 0; i < n_out; ++i) out[(size_t)s * out_stride + i] = in[(size_t)s * in_stride + (size_t)((uint64_t)i * r->in_rate / r->out_rate)];
		if (out_len) out_len[s] = n_out;
	}
	return MI_OK;
}
int mi_resampler_process(mi_resampler *r, const int16_t *in, int in_len, int in_stride, int16_t *out, int out_stride, int32_t *out_len) {
	return mi_resampler_process_masked(r, in, in_len, in_stride, out, out_stride, out_len, nullptr);
}
int mi_resampler_process_host(mi_resampler *r, const int16_t *in, int in_len, int in_stride, int16_t *out, int out_stride, int32_t *out_len) {
	return mi_resampler_process_masked(r, in, in_len, in_stride, out, out_stride, out_len, nullptr);
}

// ---- mixer: the definition
int mi_mixer_create(mi_ctx *ctx, int nconf, int mm, int ns, mi_mixer **out) {
	ARG(ctx && out && nconf > 0 && mm > 0 && mm <= MI_MIXER_MAX_CHANNELS && ns > 0);
	mi_mixer *m = new mi_mixer{ctx, nconf, mm, ns, {}, {}};
	m->flags.assign((size_t)nconf * mm, MI_MIX_LINKED | MI_MIX_ACTIVE | MI_MIX_OUTPUT);
	m->gain.assign((size_t)nconf * mm, 1.0f);
	*out = m;
	return MI_OK;
}
void mi_mixer_destroy(mi_mixer *m) { delete m; }
int mi_mixer_set_controls(mi_mixer *m, const uint8_t *f, const float *g) {
	ARG(m);
	if (f) m->flags.assign(f, f + m->flags.size());
	if (g) m->gain.assign(g, g + m->gain.size());
	return MI_OK;
}
static int sat(int64_t v) { return v > 32767 ? 32767 : (v < -32767 ? -32767 : (int)v); }
static int contrib(const mi_mixer *m, const int16_t *in, const uint8_t *hd, int c, int k, int i) {
	const size_t p = (size_t)c * m->mm + k;
	if ((m->flags[p] & (MI_MIX_LINKED | MI_MIX_ACTIVE)) != (MI_MIX_LINKED | MI_MIX_ACTIVE) || (hd && !hd[p])) return 0;
	const int v = in[p * m->ns + i];
	return m->gain[p] == 1.0f ? v : (int)(v * m->gain[p]);
}
int mi_mixer_partial_sum(mi_mixer *m, const int16_t *in, const uint8_t *hd, int32_t *sum) {
	ARG(m && in && sum);
	for (int c = 0; c < m->nconf; ++c)
		for (int i = 0; i < m->ns; ++i) {
			int32_t a = 0;
			for (int k = 0; k < m->mm; ++k) a += contrib(m, in, hd, c, k, i);
			sum[(size_t)c * m->ns + i] = a;
		}
	return MI_OK;
}
int mi_mixer_finalize(mi_mixer *m, const int16_t *in, const uint8_t *hd, const int32_t *sum, int conf_mode, int16_t *out) {
	ARG(m && in && sum && out);
	for (int c = 0; c < m->nconf; ++c)
		for (int k = 0; k < (conf_mode ? m->mm : 1); ++k) {
			const size_t p = (size_t)c * m->mm + k;
			if (conf_mode && !(m->flags[p] & MI_MIX_OUTPUT)) continue; /* (an output-only pin -- a listener -- hears everybody: mixer.hip) */
			for (int i = 0; i < m->ns; ++i)
				out[(conf_mode ? p : (size_t)c) * m->ns + i] = (int16_t)sat((int64_t)sum[(size_t)c * m->ns + i] - (conf_mode ? contrib(m, in, hd, c, k, i) : 0));
		}
	return MI_OK;
}
int mi_mixer_process_masked(mi_mixer *m, const int16_t *in, const uint8_t *hd, int conf_mode, const uint8_t *cmode, int16_t *out,
                            const uint8_t *run) {
	ARG(m && in && out);
	std::vector<int32_t> sum((size_t)m->ns);
	for (int c = 0; c < m->nconf; ++c) {
		if (run && !run[c]) continue;
		const int cm = cmode ? cmode[c] : conf_mode;
		for (int i = 0; i < m->ns; ++i) {
			int32_t a = 0;
			for (int k = 0; k < m->mm; ++k) a += contrib(m, in, hd, c, k, i);
			sum[(size_t)i] = a;
		}
		for (int k = 0; k < (cm ? m->mm : 1); ++k) {
			const size_t p = (size_t)c * m->mm + k;
			if (cm && !(m->flags[p] & MI_MIX_OUTPUT)) continue;
			for (int i = 0; i < m->ns; ++i) out[p * m->ns + i] = (int16_t)sat((int64_t)sum[(size_t)i] - (cm ? contrib(m, in, hd, c, k, i) : 0));
		}
	}
	return MI_OK;
}
int mi_mixer_process(mi_mixer *m, const int16_t *in, const uint8_t *hd, int conf_mode, int16_t *out) {
	ARG(m && in && out);
	if (conf_mode) return mi_mixer_process_masked(m, in, hd, 1, nullptr, out, nullptr);
	for (int c = 0; c < m->nconf; ++c)
		for (int i = 0; i < m->ns; ++i) {
			int64_t a = 0;
			for (int k = 0; k < m->mm; ++k) a += contrib(m, in, hd, c, k, i);
			out[(size_t)c * m->ns + i] = (int16_t)sat(a);
		}
	return MI_OK;
}
int mi_mixer_process_host(mi_mixer *m, const int16_t *in, const uint8_t *hd, int conf_mode, int16_t *out) {
	return mi_mixer_process(m, in, hd, conf_mode, out);
}

// ---- exchange
int mi_exchange_unique_id(void *id_out, size_t cap) {
	ARG(id_out && cap >= MI_EXCHANGE_ID_BYTES);
	static std::mutex mu;
	static unsigned long long next = 1;
	std::lock_guard<std::mutex> lk(mu);
	memset(id_out, 0, MI_EXCHANGE_ID_BYTES);
	snprintf((char *)id_out, MI_EXCHANGE_ID_BYTES, "mi_double-%llu", next++);
	return MI_OK;
}
int mi_exchange_create(mi_ctx *ctx, int nranks, int rank, const void *id, mi_exchange **out) {
	ARG(ctx && out && id && nranks >= 1 && rank >= 0 && rank < nranks);
	Meeting *m;
	{
		std::lock_guard<std::mutex> lk(g_meet_mu);
		Meeting *&slot = g_meetings[std::string((const char *)id, MI_EXCHANGE_ID_BYTES)];
		if (!slot) {
			slot = new Meeting();
			slot->nranks = nranks;
			slot->bufs.assign((size_t)nranks, nullptr);
		}
		m = slot;
	}
	std::unique_lock<std::mutex> lk(m->mu);
	if (m->nranks != nranks) return fail(MI_ENODEV, "ranks disagree on the size of the exchange");
	m->joined++;
	m->cv.notify_all();
	m->cv.wait(lk, [&] { return m->joined >= nranks; }); // like ncclCommInitRank: returns when everybody is in
	*out = new mi_exchange{ctx, m, nranks, rank};
	return MI_OK;
}
void mi_exchange_destroy(mi_exchange *x) { delete x; } // the meeting itself lives until the process ends (a handful of bytes per test)
int mi_exchange_ranks(const mi_exchange *x, int *nranks, int *rank) {
	ARG(x);
	if (nranks) *nranks = x->nranks;
	if (rank) *rank = x->rank;
	return MI_OK;
}
int mi_exchange_allreduce_i32(mi_exchange *x, int32_t *buf, size_t count) {
	ARG(x && buf && count > 0);
	Meeting *m = x->m;
	std::unique_lock<std::mutex> lk(m->mu);
	const int gen = m->generation;
	m->bufs[(size_t)x->rank] = buf;
	if (++m->arrived == m->nranks) { // the last one in adds everybody up and hands the totals out
		m->sum.assign(count, 0);
		for (int r = 0; r < m->nranks; ++r)
			for (size_t i = 0; i < count; ++i) m->sum[i] += m->bufs[(size_t)r][i];
		for (int r = 0; r < m->nranks; ++r)
			for (size_t i = 0; i < count; ++i) m->bufs[(size_t)r][i] = (int32_t)m->sum[i];
		m->arrived = 0;
		m->generation++;
		m->cv.notify_all();
	} else {
		m->cv.wait(lk, [&] { return m->generation != gen; });
	}
	return MI_OK;
}

// ---- volume: the running gain applied chunk by chunk (it follows its target), a plain energy meter scaled by the static gain --
// not MSVolume's arithmetic, but enough of it that WHEN a method's gain or parameter reaches the batch shows in samples and meters
void mi_volume_default_params(mi_volume_params *p) {
	if (!p) return;
	memset(p, 0, sizeof *p);
	p->static_gain = 1;
	p->vol_upramp = 0.4f, p->vol_fast_upramp = 1.2f, p->vol_downramp = 0.4f;
	p->ea_thres = 0.1f, p->ea_transmit_thres = 4, p->force = 4.0f;
	p->sustain_time = 200, p->ng_cut_time = 400;
	p->ng_threshold = 0.1f, p->ng_floorgain = 0.005f;
	p->peer = -1;
}
int mi_volume_create(mi_ctx *ctx, int n, int rate, mi_volume **out) {
	ARG(ctx && out && n > 0 && rate > 0);
	mi_volume *v = new mi_volume{ctx, n, rate, {}, {}, {}};
	mi_volume_params p;
	mi_volume_default_params(&p);
	v->params.assign((size_t)n, p);
	mi_volume_state s;
	memset(&s, 0, sizeof s);
	s.gain = s.target_gain = s.ng_gain = 1;
	v->state.assign((size_t)n, s);
	v->mx.assign((size_t)n, 0.f);
	*out = v;
	return MI_OK;
}
void mi_volume_destroy(mi_volume *v) { delete v; }
int mi_volume_set_params(mi_volume *v, int first, int count, const mi_volume_params *h) {
	ARG(v && h && first >= 0 && count >= 0 && first + count <= v->n);
	std::copy(h, h + count, v->params.begin() + first);
	return MI_OK;
}
int mi_volume_set_peer_batch(mi_volume *v, mi_volume *peers) {
	ARG(v && (!peers || peers->n >= v->n));
	v->ext = peers;
	return MI_OK;
}
int mi_volume_get_state(mi_volume *v, int first, int count, mi_volume_state *h) {
	ARG(v && h && first >= 0 && count >= 0 && first + count <= v->n);
	std::copy(v->state.begin() + first, v->state.begin() + first + count, h);
	return MI_OK;
}
int mi_volume_get_state_async(mi_volume *v, int first, int count, mi_volume_state *h) { return mi_volume_get_state(v, first, count, h); }
int mi_volume_set_state(mi_volume *v, int first, int count, const mi_volume_state *h) {
	ARG(v && h && first >= 0 && count >= 0 && first + count <= v->n);
	std::copy(h, h + count, v->state.begin() + first);
	return MI_OK;
}
int mi_volume_get_max(mi_volume *v, int first, int count, float *h) {
	ARG(v && h && first >= 0 && count >= 0 && first + count <= v->n);
	std::copy(v->mx.begin() + first, v->mx.begin() + first + count, h);
	return MI_OK;
}
int mi_volume_reset_max(mi_volume *v, int first, int count) {
	ARG(v && first >= 0 && count >= 0 && first + count <= v->n);
	std::fill(v->mx.begin() + first, v->mx.begin() + first + count, 0.f);
	return MI_OK;
}
int mi_volume_process(mi_volume *v, int16_t *x, int ns, int stride, const int32_t *per) {
	ARG(v && x && ns > 0 && stride >= ns);
	std::vector<float> before((size_t)v->n); // a peer in the same batch is read as the PREVIOUS launch left it (volume.hip: the energy double buffer)
	for (int s = 0; s < v->n; ++s) before[(size_t)s] = v->state[(size_t)s].energy;
	for (int s = 0; s < v->n; ++s) {
		const int n = per ? std::min(std::max(per[s], 0), ns) : ns;
		if (!n) continue;
		double acc = 0;
		mi_volume_state &st = v->state[(size_t)s];
		const int peer = v->params[(size_t)s].peer; // an echo limiter anybody can hear: the target follows the peer's energy
		if (peer >= 0) st.target_gain = v->params[(size_t)s].static_gain / (1.f + 1000.f * before[(size_t)peer]);
		else if (peer == MI_VOLUME_PEER_EXTERNAL && v->ext) st.target_gain = v->params[(size_t)s].static_gain / (1.f + 1000.f * v->ext->state[(size_t)s].energy);
		for (int i = 0; i < n; ++i) {
			int16_t &smp = x[(size_t)s * stride + i];
			acc += (double)smp * smp;
			const float y = (float)smp * st.gain;
			smp = (int16_t)(y > 32767.f ? 32767 : (y < -32767.f ? -32767 : (int)y));
		}
		st.gain += 0.25f * (st.target_gain - st.gain);
		v->state[(size_t)s].energy = (float)(acc / n / (32768.0 * 32768.0)) * v->params[(size_t)s].static_gain;
		v->mx[(size_t)s] = std::max(v->mx[(size_t)s], v->state[(size_t)s].energy);
	}
	return MI_OK;
}
int mi_volume_process_host(mi_volume *v, int16_t *x, int ns, int stride, const int32_t *per) { return mi_volume_process(v, x, ns, stride, per); }

// ---- equalizer: holds gains and taps, leaves the audio alone
int mi_equalizer_create(mi_ctx *ctx, int n, int rate, mi_equalizer **out) {
	ARG(ctx && out && n > 0 && rate > 0);
	const int nfft = rate < 16000 ? 128 : (rate < 32000 ? 256 : 512); // equalizer.c:60-66
	mi_equalizer *e = new mi_equalizer{ctx, n, rate, nfft, {}, {}, {}, {}};
	e->gains.assign((size_t)n, std::vector<float>((size_t)nfft / 2, 1.f));
	e->taps.assign((size_t)n, std::vector<float>((size_t)nfft, 0.f));
	e->active.assign((size_t)n, 1);
	e->last.assign((size_t)n, 0);
	*out = e;
	return MI_OK;
}
int mi_equalizer_get_history(mi_equalizer *e, int s, int16_t *h, int n) {
	ARG(e && h && s >= 0 && s < e->n && n == e->nfft);
	memset(h, 0, (size_t)n * 2);
	h[n - 2] = e->last[(size_t)s];
	return MI_OK;
}
int mi_equalizer_set_history(mi_equalizer *e, int s, const int16_t *h, int n) {
	ARG(e && s >= 0 && s < e->n && n == e->nfft);
	e->last[(size_t)s] = h ? h[n - 2] : 0;
	return MI_OK;
}
void mi_equalizer_destroy(mi_equalizer *e) { delete e; }
int mi_equalizer_fir_len(const mi_equalizer *e) { return e ? e->nfft : MI_EINVAL; }
int mi_equalizer_set_gain(mi_equalizer *e, int s, float f, float g, float w) {
	ARG(e && s >= 0 && s < e->n);
	const int i = std::min(e->nfft / 2 - 1, std::max(0, (int)(f * e->nfft / e->rate)));
	e->gains[(size_t)s][(size_t)i] *= g;
	return MI_OK;
}
int mi_equalizer_flatten(mi_equalizer *e, int s) {
	ARG(e && s >= 0 && s < e->n);
	std::fill(e->gains[(size_t)s].begin(), e->gains[(size_t)s].end(), 1.f);
	return MI_OK;
}
int mi_equalizer_prepare(mi_equalizer *e) {
	ARG(e);
	return MI_OK;
}
int mi_equalizer_set_active(mi_equalizer *e, int s, int on) {
	ARG(e && s >= 0 && s < e->n);
	e->active[(size_t)s] = on;
	return MI_OK;
}
int mi_equalizer_dump(mi_equalizer *e, int s, float *dst, int cap) {
	ARG(e && dst && s >= 0 && s < e->n);
	const int n = std::min(cap, e->nfft / 2);
	std::copy(e->gains[(size_t)s].begin(), e->gains[(size_t)s].begin() + n, dst);
	return n;
}
int mi_equalizer_get_taps(mi_equalizer *e, int s, float *dst, int cap) {
	ARG(e && dst && s >= 0 && s < e->n);
	const int n = std::min(cap, e->nfft);
	std::copy(e->taps[(size_t)s].begin(), e->taps[(size_t)s].begin() + n, dst);
	return n;
}
int mi_equalizer_set_taps(mi_equalizer *e, int s, const float *t, int n) {
	ARG(e && t && s >= 0 && s < e->n && n == e->nfft);
	e->taps[(size_t)s].assign(t, t + n);
	return MI_OK;
}
int mi_equalizer_process_masked(mi_equalizer *e, int16_t *x, int ns, int stride, const int32_t *per) {
	ARG(e && x && ns > 0 && stride >= ns);
	// a two-tap filter anybody can hear: y[i] = g * x[i] - x[i-1] / 4 with g from the stream's gains, its memory carried from block
	// to block (inactive: untouched, memory and all -- equalizer.c:283)
	for (int s = 0; s < e->n; ++s) {
		if (!e->active[(size_t)s]) continue;
		float g = 0;
		for (float v : e->gains[(size_t)s]) g += v;
		g /= (float)e->gains[(size_t)s].size();
		int16_t prev = e->last[(size_t)s];
		for (int i = 0; i < (per ? std::min(std::max(per[s], 0), ns) : ns); ++i) {
			const int16_t cur = x[(size_t)s * stride + i];
			const float y = g * (float)cur - 0.25f * (float)prev;
			x[(size_t)s * stride + i] = (int16_t)(y > 32767.f ? 32767 : (y < -32767.f ? -32767 : (int)y));
			prev = cur;
		}
		e->last[(size_t)s] = prev;
	}
	return MI_OK;
}
int mi_equalizer_process(mi_equalizer *e, int16_t *x, int ns, int stride) { return mi_equalizer_process_masked(e, x, ns, stride, nullptr); }
int mi_equalizer_process_host(mi_equalizer *e, int16_t *x, int ns, int stride) { return mi_equalizer_process_masked(e, x, ns, stride, nullptr); }

// ---- canceller: the microphone passes through
int mi_aec_framesize(int at8000, int rate) {
	const int newsize = (at8000 * rate) / 8000;
	int n = 1, next;
	while ((next = n << 1) <= newsize) n = next;
	return n;
}
int mi_aec_create(mi_ctx *ctx, int n, int rate, int F, int flen, mi_aec **out) {
	ARG(ctx && out && n > 0 && rate > 0 && flen > 0);
	if (F != 64 && F != 128 && F != 256) return fail(MI_ENOTSUP, "frame size");
	mi_aec *a = new mi_aec{ctx, n, rate, F, flen, {}};
	a->frames.assign((size_t)n, 0);
	*out = a;
	return MI_OK;
}
void mi_aec_destroy(mi_aec *a) { delete a; }
int mi_aec_reset(mi_aec *a, int first, int count) {
	ARG(a && first >= 0 && count >= 0 && first + count <= a->n);
	std::fill(a->frames.begin() + first, a->frames.begin() + first + count, 0);
	return MI_OK;
}
int mi_aec_process(mi_aec *a, const int16_t *mic, const int16_t *ref, int16_t *out, int stride, const uint8_t *run, unsigned flags) {
	ARG(a && mic && ref && out && stride >= a->F);
	for (int s = 0; s < a->n; ++s) {
		if (run && !run[s]) continue;
		volatile int16_t touch = ref[(size_t)s * stride + a->F - 1];
		(void)touch;
		memcpy(out + (size_t)s * stride, mic + (size_t)s * stride, (size_t)a->F * 2);
		a->frames[(size_t)s]++;
	}
	return MI_OK;
}
int mi_aec_process_host(mi_aec *a, const int16_t *mic, const int16_t *ref, int16_t *out, int stride, const uint8_t *run, unsigned flags) {
	return mi_aec_process(a, mic, ref, out, stride, run, flags);
}
int mi_aec_process_frames(mi_aec *a, const int16_t *mic, const int16_t *ref, int16_t *out, int stride, const uint8_t *count, int maxf, unsigned flags) {
	ARG(a && mic && ref && out && count && maxf >= 1 && maxf <= MI_AEC_MAX_TICK_FRAMES && stride >= maxf * a->F);
	for (int s = 0; s < a->n; ++s) {
		const int nf = std::min((int)count[s], maxf);
		memcpy(out + (size_t)s * stride, mic + (size_t)s * stride, (size_t)nf * a->F * 2);
		a->frames[(size_t)s] += nf;
	}
	return MI_OK;
}
// The FIFO entry: both blocks queued, every whole frame the microphone queue then holds (<= maxf) "cancelled" -- the double
// passes the microphone frame through, like its frame entry -- against the far end's frame or silence, results queued.
static void fifo_put(mi_fifo *f, int s, const int16_t *p, int n);
static int aec_fifo_tick(mi_aec *a, mi_fifo *fm, const int16_t *mic, int mic_stride, int mic_len, mi_fifo *fr, const int16_t *ref, int ref_stride,
                         const int32_t *ref_len, int tick_len, mi_fifo *fo, int maxf, uint8_t *count_out, const uint8_t *gate) {
	for (int s = 0; s < a->n; ++s) {
		if (!gate || gate[s]) fifo_put(fm, s, mic + (size_t)s * mic_stride, mic_len);
		const int rl = ref_len ? std::min(std::max(ref_len[s], 0), tick_len) : tick_len;
		fifo_put(fr, s, ref + (size_t)s * ref_stride, rl);
		std::vector<int16_t> &qm = fm->q[(size_t)s], &qr = fr->q[(size_t)s];
		int nf = std::min(maxf, (int)qm.size() / a->F);
		if ((int)fo->q[(size_t)s].size() + nf * a->F > fo->cap) {
			fo->overflow++;
			nf = 0;
		}
		for (int k = 0; k < nf; ++k) {
			fifo_put(fo, s, qm.data(), a->F);
			qm.erase(qm.begin(), qm.begin() + a->F);
			if ((int)qr.size() >= a->F) qr.erase(qr.begin(), qr.begin() + a->F);
		}
		a->frames[(size_t)s] += nf;
		if (count_out) count_out[s] = (uint8_t)nf;
	}
	return MI_OK;
}
int mi_aec_process_fifos_masked(mi_aec *a, mi_fifo *fm, const int16_t *mic, int mic_stride, mi_fifo *fr, const int16_t *ref, int ref_stride, const int32_t *ref_len,
                                int tick_len, mi_fifo *fo, int maxf, unsigned, uint8_t *count_out, const uint8_t *gate) {
	ARG(a && fm && fr && fo && mic && ref && tick_len > 0 && mic_stride >= tick_len && ref_stride >= tick_len && maxf >= 1 && maxf <= MI_AEC_MAX_TICK_FRAMES);
	ARG(fm->n == a->n && fr->n == a->n && fo->n == a->n);
	return aec_fifo_tick(a, fm, mic, mic_stride, tick_len, fr, ref, ref_stride, ref_len, tick_len, fo, maxf, count_out, gate);
}
int mi_aec_process_fifos(mi_aec *a, mi_fifo *fm, const int16_t *mic, int mic_stride, mi_fifo *fr, const int16_t *ref, int ref_stride, const int32_t *ref_len,
                         int tick_len, mi_fifo *fo, int maxf, unsigned flags, uint8_t *count_out) {
	return mi_aec_process_fifos_masked(a, fm, mic, mic_stride, fr, ref, ref_stride, ref_len, tick_len, fo, maxf, flags, count_out, nullptr);
}
int mi_aec_process_fifos_resampled_masked(mi_aec *a, mi_resampler *rs, const int16_t *mic_in, int in_len, int in_stride, mi_fifo *fm, mi_fifo *fr,
                                          const int16_t *ref, int ref_stride, const int32_t *ref_len, mi_fifo *fo, int maxf, unsigned, uint8_t *count_out,
                                          const uint8_t *gate) {
	ARG(a && rs && mic_in && fm && fr && fo && ref && in_len > 0 && in_stride >= in_len && maxf >= 1 && maxf <= MI_AEC_MAX_TICK_FRAMES);
	ARG(fm->n == a->n && fr->n == a->n && fo->n == a->n && rs->n == a->n);
	if (rs->out_rate % rs->in_rate) return fail(MI_ENOTSUP, "integer ratios only");
	const int tick_len = (int)((uint64_t)in_len * rs->out_rate / rs->in_rate);
	ARG(ref_stride >= tick_len);
	std::vector<int16_t> up((size_t)a->n * tick_len);
	const int rc = mi_resampler_process_masked(rs, mic_in, in_len, in_stride, up.data(), tick_len, nullptr, gate); // the double's own up-sampler
	if (rc != MI_OK) return rc;
	return aec_fifo_tick(a, fm, up.data(), tick_len, tick_len, fr, ref, ref_stride, ref_len, tick_len, fo, maxf, count_out, gate);
}
int mi_aec_process_fifos_resampled(mi_aec *a, mi_resampler *rs, const int16_t *mic_in, int in_len, int in_stride, mi_fifo *fm, mi_fifo *fr, const int16_t *ref,
                                   int ref_stride, const int32_t *ref_len, mi_fifo *fo, int maxf, unsigned flags, uint8_t *count_out) {
	return mi_aec_process_fifos_resampled_masked(a, rs, mic_in, in_len, in_stride, fm, fr, ref, ref_stride, ref_len, fo, maxf, flags, count_out, nullptr);
}
int mi_aec_stagger_info(const mi_aec *a, int tick_len, int *unit, int *phases) {
	ARG(a && tick_len > 0);
	if (unit) *unit = a->F / 8;
	if (phases) *phases = 8;
	return MI_OK;
}
int mi_aec_stagger_fifos(mi_aec *, mi_fifo *, mi_fifo *, int, int, int) { return fail(MI_ENOTSUP, "not modelled by the double"); }
size_t mi_aec_state_bytes(const mi_aec *a) { return a ? 64 : 0; }
size_t mi_aec_blob_bytes(const mi_aec *a) { return a ? 16 : 0; }
int mi_aec_export_state(mi_aec *a, int s, void *blob, size_t cap) {
	ARG(a && blob && s >= 0 && s < a->n && cap >= 16);
	memset(blob, 0, 16);
	memcpy(blob, "MIEC", 4);
	memcpy((char *)blob + 8, &a->frames[(size_t)s], sizeof(int));
	return MI_OK;
}
int mi_aec_import_state(mi_aec *a, int s, const void *blob, size_t size) {
	ARG(a && blob && s >= 0 && s < a->n);
	if (size != 16 || memcmp(blob, "MIEC", 4) != 0) return fail(MI_EINVAL, "not a blob of this library");
	memcpy(&a->frames[(size_t)s], (const char *)blob + 8, sizeof(int));
	return MI_OK;
}
int mi_aec_copy_state(mi_aec *dst, int df, const mi_aec *src, int sf, int count) {
	ARG(dst && src && count >= 0 && df >= 0 && sf >= 0 && df + count <= dst->n && sf + count <= src->n);
	std::copy(src->frames.begin() + sf, src->frames.begin() + sf + count, dst->frames.begin() + df);
	return MI_OK;
}
int mi_aec_get(mi_aec *a, int s, const char *what, float *dst, int cap) { return fail(MI_ENOTSUP, "no state read-back in the double"); }

// ---- scaler / pixconv: grey frames of the right size
static size_t i420_bytes(int w, int h) { return (size_t)w * (h + (h & 1)) * 3 / 2; }
int mi_scaler_create(mi_ctx *ctx, int sw, int sh, int dw, int dh, int fmt, mi_scaler **out) {
	ARG(ctx && out && sw > 0 && sh > 0 && dw > 0 && dh > 0 && (fmt == MI_PIX_I420 || fmt == MI_PIX_RGB24));
	*out = new mi_scaler{ctx, sw, sh, dw, dh, fmt};
	return MI_OK;
}
void mi_scaler_destroy(mi_scaler *s) { delete s; }
size_t mi_scaler_src_bytes(const mi_scaler *s) { return s ? (size_t)s->sw * s->sh + 2 * (size_t)((s->sw + 1) / 2) * ((s->sh + 1) / 2) : 0; }
size_t mi_scaler_dst_bytes(const mi_scaler *s) {
	if (!s) return 0;
	return s->fmt == MI_PIX_RGB24 ? (size_t)s->dw * s->dh * 3 : (size_t)s->dw * s->dh + 2 * (size_t)((s->dw + 1) / 2) * ((s->dh + 1) / 2);
}
int mi_scaler_process(mi_scaler *s, int nframes, const uint8_t *src, size_t sp, uint8_t *dst, size_t dp) {
	ARG(s && src && dst && nframes > 0 && sp >= mi_scaler_src_bytes(s) && dp >= mi_scaler_dst_bytes(s));
	for (int f = 0; f < nframes; ++f) {
		volatile uint8_t touch = src[(size_t)f * sp + mi_scaler_src_bytes(s) - 1];
		(void)touch;
		memset(dst + (size_t)f * dp, 128, mi_scaler_dst_bytes(s));
	}
	return MI_OK;
}
int mi_scaler_process_host(mi_scaler *s, int nframes, const uint8_t *src, size_t sp, uint8_t *dst, size_t dp) {
	return mi_scaler_process(s, nframes, src, sp, dst, dp);
}
// the pipelined form: the same ring of buffers, every submit done on the spot
struct mi_scaler_pipe {
	mi_scaler *sc;
	int batch, depth;
	size_t sp, dp;
	std::vector<std::vector<uint8_t>> src, dst;
	std::vector<int> n;
	uint64_t submitted = 0, collected = 0;
	bool acquired = false;
};
int mi_scaler_pipe_create(mi_scaler *s, int batch, int depth, mi_scaler_pipe **out) {
	ARG(s && out && batch > 0 && depth >= 1 && depth <= 8);
	mi_scaler_pipe *p = new mi_scaler_pipe{s, batch, depth, (mi_scaler_src_bytes(s) + 31) & ~(size_t)15, (mi_scaler_dst_bytes(s) + 15) & ~(size_t)15, {}, {}, {}};
	p->src.assign((size_t)depth, std::vector<uint8_t>((size_t)batch * p->sp));
	p->dst.assign((size_t)depth, std::vector<uint8_t>((size_t)batch * p->dp));
	p->n.assign((size_t)depth, 0);
	*out = p;
	return MI_OK;
}
void mi_scaler_pipe_destroy(mi_scaler_pipe *p) { delete p; }
int mi_scaler_pipe_in_flight(const mi_scaler_pipe *p) { return p ? (int)(p->submitted - p->collected) : MI_EINVAL; }
int mi_scaler_pipe_acquire(mi_scaler_pipe *p, uint8_t **h_src, size_t *pitch) {
	ARG(p && h_src);
	if ((int)(p->submitted - p->collected) >= p->depth) return fail(MI_EINVAL, "every batch of the ring is in flight");
	*h_src = p->src[(size_t)(p->submitted % (uint64_t)p->depth)].data();
	if (pitch) *pitch = p->sp;
	p->acquired = true;
	return MI_OK;
}
int mi_scaler_pipe_submit(mi_scaler_pipe *p, int nframes) {
	ARG(p && nframes > 0 && nframes <= p->batch && p->acquired);
	const size_t k = (size_t)(p->submitted % (uint64_t)p->depth);
	const int rc = mi_scaler_process(p->sc, nframes, p->src[k].data(), p->sp, p->dst[k].data(), p->dp);
	if (rc != MI_OK) return rc;
	p->n[k] = nframes;
	p->acquired = false;
	p->submitted++;
	return MI_OK;
}
int mi_scaler_pipe_collect(mi_scaler_pipe *p, const uint8_t **h_dst, size_t *pitch, int *nframes) {
	ARG(p && h_dst && p->submitted > p->collected);
	const size_t k = (size_t)(p->collected % (uint64_t)p->depth);
	*h_dst = p->dst[k].data();
	if (pitch) *pitch = p->dp;
	if (nframes) *nframes = p->n[k];
	p->collected++;
	return MI_OK;
}
int mi_scaler_process_planes_host(mi_scaler *s, const uint8_t *const src[3], const int ss[3], uint8_t *const dst[3], const int ds[3]) {
	ARG(s && src && ss && dst && ds);
	volatile uint8_t touch = src[0][(size_t)(s->sh - 1) * ss[0] + s->sw - 1];
	(void)touch;
	if (s->fmt == MI_PIX_RGB24) {
		for (int y = 0; y < s->dh; ++y) memset(dst[0] + (size_t)y * ds[0], 128, (size_t)s->dw * 3);
	} else {
		for (int y = 0; y < s->dh; ++y) memset(dst[0] + (size_t)y * ds[0], 128, (size_t)s->dw);
		for (int p = 1; p < 3; ++p)
			for (int y = 0; y < (s->dh + 1) / 2; ++y) memset(dst[p] + (size_t)y * ds[p], 128, (size_t)(s->dw + 1) / 2);
	}
	return MI_OK;
}
static int pix_bpp(int fmt) { return fmt == MI_PIX_YUY2 || fmt == MI_PIX_UYVY ? 2 : (fmt == MI_PIX_BGRA32 ? 4 : 3); }
int mi_pixconv_create(mi_ctx *ctx, int w, int h, int fmt, int flip, mi_pixconv **out) {
	ARG(ctx && out && w > 0 && h > 0 && !(w & 1) && fmt >= MI_PIX_YUY2 && fmt <= MI_PIX_BGRA32);
	*out = new mi_pixconv{ctx, w, h, fmt, flip};
	return MI_OK;
}
void mi_pixconv_destroy(mi_pixconv *p) { delete p; }
size_t mi_pixconv_src_bytes(const mi_pixconv *p) { return p ? (size_t)p->w * p->h * pix_bpp(p->fmt) : 0; }
size_t mi_pixconv_dst_bytes(const mi_pixconv *p) { return p ? i420_bytes(p->w, p->h) : 0; }
int mi_pixconv_process(mi_pixconv *p, int nframes, const uint8_t *src, size_t sp, uint8_t *dst, size_t dp) {
	ARG(p && src && dst && nframes > 0 && sp >= mi_pixconv_src_bytes(p) && dp >= mi_pixconv_dst_bytes(p));
	for (int f = 0; f < nframes; ++f) {
		volatile uint8_t touch = src[(size_t)f * sp + mi_pixconv_src_bytes(p) - 1];
		(void)touch;
		memset(dst + (size_t)f * dp, 128, mi_pixconv_dst_bytes(p));
	}
	return MI_OK;
}
int mi_pixconv_process_host(mi_pixconv *p, int nframes, const uint8_t *src, size_t sp, uint8_t *dst, size_t dp) {
	return mi_pixconv_process(p, nframes, src, sp, dst, dp);
}

// ---- FIFOs: MSBufferizer for a batch, plainly
int mi_fifo_create(mi_ctx *ctx, int n, int cap, mi_fifo **out) {
	ARG(ctx && out && n > 0 && cap > 0);
	mi_fifo *f = new mi_fifo{ctx, n, cap, {}, 0};
	f->q.assign((size_t)n, {});
	*out = f;
	return MI_OK;
}
void mi_fifo_destroy(mi_fifo *f) { delete f; }
static void fifo_put(mi_fifo *f, int s, const int16_t *p, int n) {
	if (n <= 0) return;
	if ((int)f->q[(size_t)s].size() + n > f->cap) {
		f->overflow++;
		return;
	}
	f->q[(size_t)s].insert(f->q[(size_t)s].end(), p, p + n);
}
int mi_fifo_push(mi_fifo *f, const int16_t *in, int ns, int stride, const int32_t *count) {
	ARG(f && in && ns > 0 && stride >= ns);
	for (int s = 0; s < f->n; ++s) fifo_put(f, s, in + (size_t)s * stride, count ? std::min(std::max(count[s], 0), ns) : ns);
	return MI_OK;
}
int mi_fifo_push_gated(mi_fifo *f, const int16_t *in, int ns, int stride, const uint8_t *gate) {
	ARG(f && in && ns > 0 && stride >= ns);
	for (int s = 0; s < f->n; ++s)
		if (!gate || gate[s]) fifo_put(f, s, in + (size_t)s * stride, ns);
	return MI_OK;
}
int mi_fifo_pop(mi_fifo *f, int frame, int16_t *out, int stride, uint8_t *ok, const uint8_t *gate, int zero_fill) {
	ARG(f && out && frame > 0 && stride >= frame);
	for (int s = 0; s < f->n; ++s) {
		std::vector<int16_t> &q = f->q[(size_t)s];
		const bool take = (!gate || gate[s]) && (int)q.size() >= frame;
		if (take) {
			std::copy(q.begin(), q.begin() + frame, out + (size_t)s * stride);
			q.erase(q.begin(), q.begin() + frame);
		} else if (zero_fill) {
			memset(out + (size_t)s * stride, 0, (size_t)frame * 2);
		}
		if (ok) ok[s] = take;
	}
	return MI_OK;
}
int mi_fifo_pop_frames(mi_fifo *, int, int, int16_t *, int, uint8_t *, const uint8_t *, int) { return fail(MI_ENOTSUP, "not modelled by the double"); }
int mi_fifo_push_frames(mi_fifo *, const int16_t *, int, int, int, const uint8_t *) { return fail(MI_ENOTSUP, "not modelled by the double"); }
int mi_fifo_push_lead(mi_fifo *, int, int, int, int) { return fail(MI_ENOTSUP, "not modelled by the double"); }
int mi_fifo_phase_of(int stream, int phases) { return phases > 0 ? (int)((((unsigned)stream * 0x9E3779B1u) >> 16) % (unsigned)phases) : 0; }
int mi_fifo_push_silence(mi_fifo *f, const int32_t *count) {
	ARG(f && count);
	for (int s = 0; s < f->n; ++s)
		if (count[s] > 0) {
			const std::vector<int16_t> z((size_t)count[s], 0);
			fifo_put(f, s, z.data(), count[s]);
		}
	return MI_OK;
}
int mi_fifo_levels(mi_fifo *f, int32_t *lv) {
	ARG(f && lv);
	for (int s = 0; s < f->n; ++s) lv[s] = (int32_t)f->q[(size_t)s].size();
	return MI_OK;
}
int mi_fifo_snapshot(mi_fifo *f, int16_t *rings, int32_t *head, int32_t *level) { // (the double's queues start at 0)
	ARG(f);
	for (int s = 0; s < f->n; ++s) {
		if (rings) {
			memset(rings + (size_t)s * f->cap, 0, (size_t)f->cap * 2);
			std::copy(f->q[(size_t)s].begin(), f->q[(size_t)s].end(), rings + (size_t)s * f->cap);
		}
		if (head) head[s] = 0;
		if (level) level[s] = (int32_t)f->q[(size_t)s].size();
	}
	return MI_OK;
}
int mi_fifo_overflows(mi_fifo *f, int32_t *h) {
	ARG(f && h);
	*h = f->overflow;
	return MI_OK;
}
int mi_fifo_reset_range(mi_fifo *f, int first, int count) {
	ARG(f && first >= 0 && count >= 0 && first + count <= f->n);
	for (int s = first; s < first + count; ++s) f->q[(size_t)s].clear();
	return MI_OK;
}
int mi_fifo_reset_range_at(mi_fifo *f, int first, int count, int head) { // (the double's queues have no ring: the offset is moot)
	ARG(f && head >= 0 && (head & 7) == 0);
	return mi_fifo_reset_range(f, first, count);
}
int mi_fifo_export_range(mi_fifo *f, int first, int count, int16_t *h, int stride, int32_t *level) {
	ARG(f && h && level && first >= 0 && count >= 0 && first + count <= f->n && stride >= f->cap);
	for (int k = 0; k < count; ++k) {
		const std::vector<int16_t> &q = f->q[(size_t)(first + k)];
		level[k] = (int32_t)q.size();
		std::copy(q.begin(), q.end(), h + (size_t)k * stride);
	}
	return MI_OK;
}
int mi_fifo_import_range(mi_fifo *f, int first, int count, const int16_t *h, int stride, const int32_t *level, int tail_at_end) {
	ARG(f && h && level && first >= 0 && count >= 0 && first + count <= f->n);
	for (int k = 0; k < count; ++k) {
		ARG(level[k] >= 0 && level[k] <= f->cap && level[k] <= stride && (!tail_at_end || (level[k] & 7) == 0));
		f->q[(size_t)(first + k)].assign(h + (size_t)k * stride, h + (size_t)k * stride + level[k]);
	}
	return MI_OK;
}
int mi_fifo_reset(mi_fifo *f) {
	ARG(f);
	f->overflow = 0;
	return mi_fifo_reset_range(f, 0, f->n);
}
int mi_volume_process_fifo(mi_volume *v, mi_fifo *f, int16_t *out, int ns, int stride) {
	ARG(v && f && out && f->n == v->n);
	const int rc = mi_fifo_pop(f, ns, out, stride, nullptr, nullptr, 1);
	return rc != MI_OK ? rc : mi_volume_process(v, out, ns, stride, nullptr);
}

int mi_volume_process_fifo_flags(mi_volume *v, mi_fifo *f, int16_t *out, int ns, int stride, unsigned flags) {
	ARG(v && f && out && f->n == v->n);
	std::vector<int32_t> per((size_t)v->n, 0);
	for (int s = 0; s < v->n; ++s) {
		std::vector<int16_t> &q = f->q[(size_t)s];
		const bool has = (int)q.size() >= ns;
		if (has) {
			std::copy(q.begin(), q.begin() + ns, out + (size_t)s * stride);
			q.erase(q.begin(), q.begin() + ns);
		} else if (!(flags & MI_VOLMIX_DRY_SKIPS)) {
			memset(out + (size_t)s * stride, 0, (size_t)ns * 2);
		}
		per[(size_t)s] = (has || !(flags & MI_VOLMIX_DRY_SKIPS)) ? ns : 0;
	}
	return mi_volume_process(v, out, ns, stride, per.data());
}
int mi_volume_process_fifo_range(mi_volume *v, mi_fifo *f, int16_t *out, int ns, int stride, int first, int count) {
	ARG(v && f && out && f->n == v->n && first >= 0 && count >= 0 && first + count <= v->n);
	for (int s = first; s < first + count; ++s) {
		std::vector<int16_t> &q = f->q[(size_t)s];
		if ((int)q.size() >= ns) {
			std::copy(q.begin(), q.begin() + ns, out + (size_t)s * stride);
			q.erase(q.begin(), q.begin() + ns);
		} else {
			memset(out + (size_t)s * stride, 0, (size_t)ns * 2);
		}
	}
	return MI_OK;
}
int mi_mixer_process_volume_fifo_flags(mi_mixer *m, mi_volume *v, int first, mi_fifo *f, int16_t *out, unsigned flags, const uint8_t *run) {
	ARG(m && v && f && out && first >= 0 && first + m->nconf * m->mm <= v->n && f->n == v->n);
	std::vector<int16_t> ticks((size_t)v->n * m->ns);
	std::vector<int32_t> per((size_t)v->n, 0);
	for (int c = 0; c < m->nconf; ++c) {
		if (run && !run[c]) continue; // the conference does not tick: nothing popped, nothing written
		const int s0 = first + c * m->mm;
		for (int s = s0; s < s0 + m->mm; ++s) { // a dry leg is metered on silence, or (MI_VOLMIX_DRY_SKIPS) not at all
			const bool has = (int)f->q[(size_t)s].size() >= m->ns;
			per[(size_t)s] = (has || !(flags & MI_VOLMIX_DRY_SKIPS)) ? m->ns : 0;
		}
		int rc = mi_volume_process_fifo_range(v, f, ticks.data(), m->ns, m->ns, s0, m->mm); // pops (zeros for the dry ones)
		if (rc != MI_OK) return rc;
	}
	int rc = mi_volume_process(v, ticks.data(), m->ns, m->ns, per.data());
	if (rc != MI_OK) return rc;
	std::vector<uint8_t> all;
	std::vector<int16_t> mixed((size_t)m->nconf * m->mm * m->ns);
	rc = mi_mixer_process(m, ticks.data() + (size_t)first * m->ns, nullptr, 1, mixed.data());
	for (int c = 0; c < m->nconf && rc == MI_OK; ++c)
		if (!run || run[c]) memcpy(out + (size_t)c * m->mm * m->ns, mixed.data() + (size_t)c * m->mm * m->ns, (size_t)m->mm * m->ns * 2);
	return rc;
}
int mi_mixer_process_volume_fifo(mi_mixer *m, mi_volume *v, int first, mi_fifo *f, int16_t *out) {
	return mi_mixer_process_volume_fifo_flags(m, v, first, f, out, 0u, nullptr);
}

// ---- codecs and friends
int mi_g711_decode(mi_ctx *ctx, int law, const uint8_t *codes, size_t cs, int16_t *pcm, size_t ps, const int32_t *len, int n, size_t rows) {
	ARG(ctx && codes && pcm && n >= 0 && (law == MI_LAW_PCMA || law == MI_LAW_PCMU));
	for (size_t r = 0; r < rows; ++r)
		for (int i = 0; i < (len ? std::min(std::max(len[r], 0), n) : n); ++i) pcm[r * ps + i] = (int16_t)(((int)codes[r * cs + i] - 128) * 256);
	return MI_OK;
}
int mi_g711_encode(mi_ctx *ctx, int law, const int16_t *pcm, size_t ps, uint8_t *codes, size_t cs, const int32_t *len, int n, size_t rows) {
	ARG(ctx && codes && pcm && n >= 0 && (law == MI_LAW_PCMA || law == MI_LAW_PCMU));
	for (size_t r = 0; r < rows; ++r)
		for (int i = 0; i < (len ? std::min(std::max(len[r], 0), n) : n); ++i) codes[r * cs + i] = (uint8_t)((pcm[r * ps + i] >> 8) + 128);
	return MI_OK;
}
int mi_l16_swap(mi_ctx *ctx, const int16_t *in, int16_t *out, size_t n) {
	ARG(ctx && in && out);
	for (size_t i = 0; i < n; ++i) out[i] = (int16_t)(((uint16_t)in[i] >> 8) | ((uint16_t)in[i] << 8));
	return MI_OK;
}
int mi_chan_adapt(mi_ctx *ctx, int mode, const int16_t *a, const int16_t *b, int16_t *out, size_t frames) {
	ARG(ctx && out && (a || mode == MI_CHAN_TWO_MONO_TO_STEREO));
	for (size_t i = 0; i < frames; ++i) {
		if (mode == MI_CHAN_MONO_TO_STEREO) out[2 * i] = out[2 * i + 1] = a[i];
		else if (mode == MI_CHAN_STEREO_TO_MONO) out[i] = a[2 * i];
		else out[2 * i] = a ? a[i] : 0, out[2 * i + 1] = b ? b[i] : 0;
	}
	return MI_OK;
}
int mi_flowctl_create(mi_ctx *ctx, int n, int max_block, mi_flowctl **out) {
	ARG(ctx && out && n > 0 && max_block >= 3 && max_block <= 2048);
	mi_flowctl *f = new mi_flowctl{ctx, n, max_block, {}, {}};
	f->target.assign((size_t)n, 0);
	f->total.assign((size_t)n, 0);
	*out = f;
	return MI_OK;
}
void mi_flowctl_destroy(mi_flowctl *f) { delete f; }
int mi_flowctl_set_config(mi_flowctl *f, int first, int count, int strategy, float thr) {
	ARG(f && first >= 0 && count >= 0 && first + count <= f->n);
	return MI_OK;
}
int mi_flowctl_request_drop(mi_flowctl *f, const uint32_t *drop, const uint32_t *total) {
	ARG(f && drop && total);
	for (int s = 0; s < f->n; ++s)
		if (drop[s]) f->target[(size_t)s] = drop[s], f->total[(size_t)s] = total[s];
	return MI_OK;
}
int mi_flowctl_process(mi_flowctl *f, const int16_t *in, size_t is, const int32_t *len, int n, int16_t *out, size_t os, int32_t *out_len) {
	ARG(f && in && out && n >= 0 && n <= f->max_block);
	for (int s = 0; s < f->n; ++s) {
		const int k = len ? std::min(std::max(len[s], 0), n) : n;
		if (in != out) memmove(out + (size_t)s * os, in + (size_t)s * is, (size_t)k * 2);
		if (out_len) out_len[s] = k;
	}
	return MI_OK;
}
int mi_flowctl_get_state(mi_flowctl *f, int s, uint32_t o[4]) {
	ARG(f && o && s >= 0 && s < f->n);
	o[0] = f->target[(size_t)s], o[1] = f->total[(size_t)s], o[2] = o[3] = 0;
	return MI_OK;
}
int mi_flowctl_reset(mi_flowctl *f, int first, int count) {
	ARG(f && first >= 0 && count >= 0 && first + count <= f->n);
	return MI_OK;
}
int mi_plc_create(mi_ctx *ctx, int n, int rate, int max_block, mi_plc **out) {
	ARG(ctx && out && n > 0 && rate > 0 && max_block > 0);
	*out = new mi_plc{ctx, n, rate, max_block};
	return MI_OK;
}
void mi_plc_destroy(mi_plc *p) { delete p; }
int mi_plc_reset(mi_plc *p, int first, int count) {
	ARG(p && first >= 0 && count >= 0 && first + count <= p->n);
	return MI_OK;
}
int mi_plc_process(mi_plc *p, int16_t *blocks, size_t stride, const int32_t *len, const uint8_t *mode) {
	ARG(p && blocks && len && mode);
	for (int s = 0; s < p->n; ++s)
		if (mode[s] & MI_PLC_CONCEAL) memset(blocks + (size_t)s * stride, 0, (size_t)std::min(std::max(len[s], 0), p->max_block) * 2);
	return MI_OK;
}
int mi_plc_info(mi_plc *p, int s, int32_t o[3]) {
	ARG(p && o);
	o[0] = o[1] = o[2] = 0;
	return MI_OK;
}

// ---- session: not modelled
void mi_session_default_config(mi_session_config *c) {
	if (c) memset(c, 0, sizeof *c);
}
int mi_session_create(mi_ctx *, const mi_session_config *, mi_session **) { return fail(MI_ENOTSUP, "mi_session is not modelled by the double"); }
void mi_session_destroy(mi_session *) {}
int mi_session_tick_samples(const mi_session *, int *, int *) { return MI_ENOTSUP; }
int mi_session_tick_bytes(const mi_session *, int *, int *, int *) { return MI_ENOTSUP; }
int mi_session_acquire(mi_session *, int16_t **, int16_t **) { return MI_ENOTSUP; }
int mi_session_events(mi_session *, uint8_t **) { return MI_ENOTSUP; }
int mi_session_submit(mi_session *) { return MI_ENOTSUP; }
int mi_session_collect(mi_session *, const int16_t **) { return MI_ENOTSUP; }
int mi_session_in_flight(const mi_session *) { return 0; }
int mi_session_set_controls(mi_session *, const uint8_t *, const float *) { return MI_ENOTSUP; }
int mi_session_reset_streams(mi_session *, int, int) { return MI_ENOTSUP; }
int mi_session_get_levels(mi_session *, float *) { return MI_ENOTSUP; }
int mi_session_add_member(mi_session *, int) { return MI_ENOTSUP; }
int mi_session_remove_member(mi_session *, int) { return MI_ENOTSUP; }
int mi_session_member_count(const mi_session *, int) { return MI_ENOTSUP; }
int mi_session_active_speakers(mi_session *, uint64_t, int32_t *, float *) { return MI_ENOTSUP; }

} // extern "C"
