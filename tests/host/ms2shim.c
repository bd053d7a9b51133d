/* ms2shim.c -- a small host runtime with mediastreamer2's plugin-facing API, so the
 * MI355X filter plugin can be loaded, linked into graphs and ticked in this
 * repository's tests exactly the way the reference's factory / ticker would do it
 * (include/ms2_plugin_abi.h lists what is mirrored and from where).  Our own code:
 * it restates the BEHAVIOUR of src/base/{msfactory,msfilter,msticker,msqueue}.c
 * that the plugin depends on (registration order, first-match lookup, method
 * dispatch, bufferizer all-or-nothing reads, graph execution order), nothing more.
 * In a deployment the real libmediastreamer/oRTP provide these symbols instead.
 */
#define _GNU_SOURCE
#include "../../include/ms2_plugin_abi.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ misc */
void *ms_malloc0(size_t sz) { return calloc(1, sz ? sz : 1); }
void ms_free(void *p) { free(p); }

/* Messages are dropped, warnings and errors COUNTED and the first 32 of each printed (MS2SHIM_VERBOSE=1: everything).  The
 * reference's filters warn per filter -- "Getting reference signal but no echo to synchronize on" (speexec.c:247), "Not enough ref
 * samples, using zeroes" (:265) -- and so do the plugin's: with 10^4 - 10^5 legs attached at once that is as many lines in the first two
 * ticks, and a handler that writes each to an unbuffered stderr (three calls under its lock, sixteen ticker threads) took those ticks
 * to 0.2 - 0.8 s by itself.  What printing costs is the application's log handler's business (bctbx_set_log_handler); this runtime counts. */
static int g_verbose = -1;
static volatile long g_log_counts[2];
long ms2shim_log_count(int errors) { return __atomic_load_n(&g_log_counts[errors ? 1 : 0], __ATOMIC_RELAXED); }
static void vlog(const char *lvl, const char *fmt, va_list ap) {
	if (g_verbose < 0) g_verbose = getenv("MS2SHIM_VERBOSE") ? 1 : 0;
	if (!g_verbose && lvl[0] == 'm') return;
	if (lvl[0] != 'm' && !g_verbose) { /* the first 32 of a kind are printed, the rest counted -- per thread, folded into the total in lots (sixteen tickers' warnings, one per leg, on ONE counter were 7 % of the first ticks after a large attach) */
		static __thread long mine[2];
		const int k = lvl[0] == 'e';
		if (__atomic_load_n(&g_log_counts[k], __ATOMIC_RELAXED) > 32) {
			if (++mine[k] >= 256) __sync_add_and_fetch(&g_log_counts[k], mine[k]), mine[k] = 0;
			return;
		}
		if (__sync_add_and_fetch(&g_log_counts[k], 1) > 32) return;
	}
	char line[512];
	int n = snprintf(line, sizeof(line), "ms2shim-%s: ", lvl);
	n += vsnprintf(line + n, sizeof(line) - (size_t)n - 1, fmt, ap);
	if (n > (int)sizeof(line) - 2) n = (int)sizeof(line) - 2;
	line[n++] = '\n';
	fwrite(line, 1, (size_t)n, stderr);
}
void ms_message(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vlog("message", fmt, ap); va_end(ap); }
void ms_warning(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vlog("warning", fmt, ap); va_end(ap); }
void ms_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vlog("error", fmt, ap); va_end(ap); }

/* ------------------------------------------------------------ mblk / queue */
mblk_t *allocb(size_t size, int unused) {
	(void)unused;
	mblk_t *m = (mblk_t *)calloc(1, sizeof(mblk_t));
	dblk_t *d = (dblk_t *)malloc(sizeof(dblk_t) + size);
	d->db_base = (unsigned char *)(d + 1);
	d->db_lim = d->db_base + size;
	d->db_freefn = NULL;
	d->db_ref = 1;
	m->b_datap = d;
	m->b_rptr = m->b_wptr = d->db_base;
	return m;
}

mblk_t *esballoc(uint8_t *buf, size_t size, int pri, void (*freefn)(void *)) { /* ortp str_utils.c */
	(void)pri;
	mblk_t *m = (mblk_t *)calloc(1, sizeof(mblk_t));
	dblk_t *d = (dblk_t *)malloc(sizeof(dblk_t));
	d->db_base = buf;
	d->db_lim = buf + size;
	d->db_freefn = freefn;
	d->db_ref = 1;
	m->b_datap = d;
	m->b_rptr = m->b_wptr = buf;
	return m;
}

void freeb(mblk_t *m) {
	if (!m) return;
	if (m->b_datap && __atomic_sub_fetch(&m->b_datap->db_ref, 1, __ATOMIC_ACQ_REL) == 0) {
		if (m->b_datap->db_freefn) m->b_datap->db_freefn(m->b_datap->db_base); /* dblk_unref: the caller's buffer goes back to its owner */
		free(m->b_datap);
	}
	free(m);
}

void freemsg(mblk_t *m) {
	while (m) {
		mblk_t *n = m->b_cont;
		freeb(m);
		m = n;
	}
}

mblk_t *dupb(mblk_t *m) {
	mblk_t *n = (mblk_t *)calloc(1, sizeof(mblk_t));
	__atomic_add_fetch(&m->b_datap->db_ref, 1, __ATOMIC_ACQ_REL);
	n->b_datap = m->b_datap;
	n->b_rptr = m->b_rptr;
	n->b_wptr = m->b_wptr;
	mblk_meta_copy(m, n);
	return n;
}

mblk_t *dupmsg(mblk_t *m) {
	mblk_t *head = dupb(m), *tail = head;
	for (m = m->b_cont; m; m = m->b_cont) {
		tail->b_cont = dupb(m);
		tail = tail->b_cont;
	}
	return head;
}

size_t msgdsize(const mblk_t *m) {
	size_t n = 0;
	for (; m; m = m->b_cont) n += (size_t)(m->b_wptr - m->b_rptr);
	return n;
}

void mblk_meta_copy(const mblk_t *src, mblk_t *dst) {
	dst->reserved1 = src->reserved1;
	dst->reserved2 = src->reserved2;
	dst->ttl_or_hl = src->ttl_or_hl;
}

void qinit(queue_t *q) {
	memset(q, 0, sizeof(*q));
	q->_q_stopper.b_next = &q->_q_stopper;
	q->_q_stopper.b_prev = &q->_q_stopper;
}

void putq(queue_t *q, mblk_t *m) {
	m->b_next = &q->_q_stopper;
	m->b_prev = q->_q_stopper.b_prev;
	q->_q_stopper.b_prev->b_next = m;
	q->_q_stopper.b_prev = m;
	q->q_mcount++;
}

mblk_t *getq(queue_t *q) {
	mblk_t *m = q->_q_stopper.b_next;
	if (m == &q->_q_stopper) return NULL;
	q->_q_stopper.b_next = m->b_next;
	m->b_next->b_prev = &q->_q_stopper;
	m->b_next = m->b_prev = NULL;
	q->q_mcount--;
	return m;
}

mblk_t *peekq(queue_t *q) {
	mblk_t *m = q->_q_stopper.b_next;
	return m == &q->_q_stopper ? NULL : m;
}

void flushq(queue_t *q, int how) {
	(void)how;
	mblk_t *m;
	while ((m = getq(q)) != NULL) freemsg(m);
}

void ms_queue_flush(MSQueue *q) { flushq(&q->q, 0); }

/* --------------------------------------------------------------- bufferizer
 * behaviour of src/base/msqueue.c:54-113: byte FIFO over mblks, read is all-or-nothing */
void ms_bufferizer_init(MSBufferizer *obj) {
	qinit(&obj->q);
	obj->size = 0;
}
MSBufferizer *ms_bufferizer_new(void) {
	MSBufferizer *b = (MSBufferizer *)ms_malloc0(sizeof(*b));
	ms_bufferizer_init(b);
	return b;
}
void ms_bufferizer_put(MSBufferizer *obj, mblk_t *m) {
	obj->size += msgdsize(m);
	putq(&obj->q, m);
}
void ms_bufferizer_put_from_queue(MSBufferizer *obj, MSQueue *q) {
	mblk_t *m;
	while ((m = ms_queue_get(q)) != NULL) ms_bufferizer_put(obj, m);
}
size_t ms_bufferizer_read(MSBufferizer *obj, uint8_t *data, size_t datalen) {
	size_t done = 0;
	if (datalen == 0 || obj->size < datalen) return 0;
	mblk_t *m = peekq(&obj->q);
	mblk_meta_copy(m, &obj->q._q_stopper);
	while (done < datalen) {
		size_t avail = (size_t)(m->b_wptr - m->b_rptr);
		size_t n = avail < datalen - done ? avail : datalen - done;
		if (data) memcpy(data + done, m->b_rptr, n);
		done += n;
		m->b_rptr += n;
		if (m->b_rptr == m->b_wptr) {
			if (m->b_cont) {
				m = m->b_cont;
			} else {
				freemsg(getq(&obj->q));
				m = peekq(&obj->q);
			}
		}
	}
	obj->size -= datalen;
	return datalen;
}
void ms_bufferizer_skip_bytes(MSBufferizer *obj, int bytes) { ms_bufferizer_read(obj, NULL, (size_t)bytes); }
void ms_bufferizer_flush(MSBufferizer *obj) {
	obj->size = 0;
	flushq(&obj->q, 0);
}
void ms_bufferizer_uninit(MSBufferizer *obj) { flushq(&obj->q, 0); }
void ms_bufferizer_destroy(MSBufferizer *obj) {
	ms_bufferizer_uninit(obj);
	ms_free(obj);
}

/* ------------------------------------------------------------------ factory */
typedef struct DescNode {
	MSFilterDesc *desc;
	struct DescNode *next;
} DescNode;

struct _MSFactory {
	DescNode *descs; /* most recently registered FIRST (msfactory.c:281) */
	void *plugins[16];
	int nplugins;
};

MSFactory *ms_factory_new(void) { return (MSFactory *)ms_malloc0(sizeof(MSFactory)); }

void ms_factory_destroy(MSFactory *f) {
	if (!f) return;
	while (f->descs) {
		DescNode *n = f->descs->next;
		free(f->descs);
		f->descs = n;
	}
	/* plugins stay mapped: like the reference there is no uninit hook on Unix (msfactory.c:761-771) */
	free(f);
}

void ms_factory_register_filter(MSFactory *f, MSFilterDesc *desc) {
	if (desc->id == MS_FILTER_NOT_SET_ID) {
		ms_error("MSFilterId for %s not set !", desc->name);
		abort(); /* ms_fatal, msfactory.c:260-262 */
	}
	desc->flags |= MS_FILTER_IS_ENABLED;
	DescNode *n = (DescNode *)malloc(sizeof(*n));
	n->desc = desc;
	n->next = f->descs;
	f->descs = n;
}

MSFilterDesc *ms_factory_lookup_filter_by_id(MSFactory *f, MSFilterId id) {
	for (DescNode *n = f->descs; n; n = n->next)
		if (n->desc->id == id) return n->desc;
	return NULL;
}

MSFilterDesc *ms_factory_lookup_filter_by_name(MSFactory *f, const char *name) {
	for (DescNode *n = f->descs; n; n = n->next)
		if (strcmp(n->desc->name, name) == 0) return n->desc;
	return NULL;
}

int ms_factory_load_plugin(MSFactory *f, const char *path) {
	void *h = dlopen(path, RTLD_NOW);
	if (!h) {
		ms_error("ms_factory_load_plugin: %s", dlerror());
		return -1;
	}
	const char *base = strrchr(path, '/');
	base = base ? base + 1 : path;
	char sym[256];
	snprintf(sym, sizeof(sym), "%s", base);
	char *dot = strstr(sym, ".so");
	if (dot) *dot = 0;
	strncat(sym, "_init", sizeof(sym) - strlen(sym) - 1);
	void (*init)(MSFactory *) = (void (*)(MSFactory *))dlsym(h, sym);
	if (!init) {
		ms_error("plugin %s lacks %s()", path, sym);
		dlclose(h);
		return -1;
	}
	init(f);
	if (f->nplugins < 16) f->plugins[f->nplugins++] = h;
	return 0;
}

typedef struct ShimHdr {
	int idx; /* index in TickerImpl::filters while attached, else -1 */
	int pad[3];
} ShimHdr;
#define SHIM_HDR(f) (((ShimHdr *)(f)) - 1)
MSFilter *ms_factory_create_filter(MSFactory *fac, MSFilterId id) {
	if (id == MS_FILTER_PLUGIN_ID) return NULL; /* msfactory.c:419-422 */
	MSFilterDesc *d = ms_factory_lookup_filter_by_id(fac, id);
	if (!d) {
		ms_error("No such filter with id %i", (int)id);
		return NULL;
	}
	/* (this runtime's own word in FRONT of the filter -- its place in its ticker's array -- so that a detach costs the graph it
	 * detaches, as msticker.c:197-218 does, not a pass over every filter of the ticker) */
	ShimHdr *hdr = (ShimHdr *)ms_malloc0(sizeof(ShimHdr) + sizeof(MSFilter));
	MSFilter *f = (MSFilter *)(hdr + 1);
	hdr->idx = -1;
	pthread_mutex_init(&f->lock, NULL);
	f->desc = d;
	f->factory = fac;
	if (d->ninputs > 0) f->inputs = (MSQueue **)ms_malloc0(sizeof(MSQueue *) * (size_t)d->ninputs);
	if (d->noutputs > 0) f->outputs = (MSQueue **)ms_malloc0(sizeof(MSQueue *) * (size_t)d->noutputs);
	if (d->init) d->init(f);
	return f;
}

/* ------------------------------------------------------------------- filter */
typedef struct Notify {
	MSFilterNotifyFunc fn;
	void *ud;
	struct Notify *next;
} Notify;

static void ms2shim_purge_tasks(MSFilter *f);
void ms_filter_destroy(MSFilter *f) {
	if (!f) return;
	if (f->ticker && f->postponed_task) ms2shim_purge_tasks(f); /* never leave a task pointing at freed memory */
	if (f->desc->uninit) f->desc->uninit(f);
	for (Notify *n = (Notify *)f->notify_callbacks; n;) {
		Notify *nx = n->next;
		free(n);
		n = nx;
	}
	free(f->inputs);
	free(f->outputs);
	pthread_mutex_destroy(&f->lock);
	free(SHIM_HDR(f));
}

int ms_filter_link(MSFilter *f1, int pin1, MSFilter *f2, int pin2) {
	if (pin1 >= f1->desc->noutputs || pin2 >= f2->desc->ninputs) return -1;
	if (f1->outputs[pin1] || f2->inputs[pin2]) return -1;
	MSQueue *q = (MSQueue *)ms_malloc0(sizeof(MSQueue));
	qinit(&q->q);
	q->prev.filter = f1;
	q->prev.pin = pin1;
	q->next.filter = f2;
	q->next.pin = pin2;
	f1->outputs[pin1] = q;
	f2->inputs[pin2] = q;
	f1->n_connected_outputs++;
	f2->n_connected_inputs++;
	return 0;
}

int ms_filter_unlink(MSFilter *f1, int pin1, MSFilter *f2, int pin2) {
	MSQueue *q = f1->outputs[pin1];
	if (!q || q != f2->inputs[pin2]) return -1;
	f1->outputs[pin1] = NULL;
	f2->inputs[pin2] = NULL;
	f1->n_connected_outputs--;
	f2->n_connected_inputs--;
	flushq(&q->q, 0);
	free(q);
	return 0;
}

int ms_filter_call_method(MSFilter *f, unsigned int id, void *arg) {
	/* msfilter.c:171-197: the method's owner must be this filter, the base id or an interface */
	unsigned int magic = id >> 16;
	if (magic != (unsigned int)f->desc->id && magic != MS_FILTER_BASE_ID && magic <= MSFilterInterfaceBegin) {
		ms_error("Method type checking failed when calling %u on filter %s", id, f->desc->name);
		abort();
	}
	MSFilterMethod *m = f->desc->methods;
	for (; m && m->method; ++m)
		if (m->id == id) return m->method(f, arg);
	if (magic != MS_FILTER_BASE_ID) ms_error("no such method on filter %s, fid=%u method index=%u", f->desc->name, magic, (id >> 8) & 0xff);
	return -1;
}

void ms_filter_add_notify_callback(MSFilter *f, MSFilterNotifyFunc fn, void *ud, bool_t synchronous) {
	(void)synchronous;
	Notify *n = (Notify *)malloc(sizeof(*n));
	n->fn = fn;
	n->ud = ud;
	n->next = (Notify *)f->notify_callbacks;
	f->notify_callbacks = n;
}

void ms_filter_notify(MSFilter *f, unsigned int id, void *arg) {
	for (Notify *n = (Notify *)f->notify_callbacks; n; n = n->next) n->fn(n->ud, f, id, arg);
}

void ms_filter_notify_no_arg(MSFilter *f, unsigned int id) { ms_filter_notify(f, id, NULL); }

/* -------------------------------------------------------------------- video */
/* Frame layout, block header and scaler indirection of src/voip/msvideo.c (test runtime only). */
typedef struct _mblk_video_header { /* msvideo.c:79-83 */
	uint16_t w, h;
	int pad[3];
} mblk_video_header;

void ms_yuv_buf_init(YuvBuf *buf, int w, int h, int stride, uint8_t *ptr) { /* msvideo.c:85-99 */
	int ysize = stride * ((h & 1) ? h + 1 : h), usize = ysize / 4;
	buf->w = w;
	buf->h = h;
	buf->planes[0] = ptr;
	buf->planes[1] = buf->planes[0] + ysize;
	buf->planes[2] = buf->planes[1] + usize;
	buf->planes[3] = 0;
	buf->strides[0] = stride;
	buf->strides[1] = stride / 2;
	buf->strides[2] = buf->strides[1];
	buf->strides[3] = 0;
}

int ms_yuv_buf_init_from_mblk(YuvBuf *buf, mblk_t *m) { /* msvideo.c:101-113 */
	mblk_video_header *hdr = (mblk_video_header *)m->b_datap->db_base;
	int w = hdr->w, h = hdr->h;
	if (m->b_cont == NULL) ms_yuv_buf_init(buf, w, h, w, m->b_rptr);
	else ms_yuv_buf_init(buf, w, h, w, m->b_cont->b_rptr);
	return 0;
}

int ms_yuv_buf_init_from_mblk_with_size(YuvBuf *buf, mblk_t *m, int w, int h) { /* msvideo.c:115-119 */
	if (m->b_cont != NULL) m = m->b_cont;
	ms_yuv_buf_init(buf, w, h, w, m->b_rptr);
	return 0;
}

int ms_picture_init_from_mblk_with_size(MSPicture *buf, mblk_t *m, MSPixFmt fmt, int w, int h) { /* :121-160 */
	if (m->b_cont != NULL) m = m->b_cont;
	int bpp;
	switch (fmt) {
		case MS_YUV420P: return ms_yuv_buf_init_from_mblk_with_size(buf, m, w, h);
		case MS_YUY2:
		case MS_YUYV:
		case MS_UYVY: bpp = 2; break;
		case MS_RGB24:
		case MS_RGB24_REV: bpp = 3; break;
		case MS_RGBA32:
		case MS_RGBA32_REV: bpp = 4; break;
		default: ms_error("Unsupported format %i with %dx%d", fmt, w, h); return -1;
	}
	memset(buf, 0, sizeof(*buf));
	buf->w = w;
	buf->h = h;
	buf->planes[0] = m->b_rptr;
	buf->strides[0] = w * bpp;
	return 0;
}

static mblk_t *yuv_block(int size, int w, int h) { /* ms_yuv_allocator_get msvideo.c:282-296 */
	const int header_size = (int)sizeof(mblk_video_header), padding = 16;
	mblk_t *msg = allocb((size_t)(header_size + size + padding), 0);
	mblk_video_header *hdr = (mblk_video_header *)msg->b_wptr;
	hdr->w = (uint16_t)w;
	hdr->h = (uint16_t)h;
	msg->b_rptr += header_size;
	msg->b_wptr += header_size;
	msg->b_wptr += size;
	return msg;
}

mblk_t *ms_yuv_buf_alloc(YuvBuf *buf, int w, int h) { /* msvideo.c:162-176 */
	int size = (w * ((h & 1) ? h + 1 : h) * 3) / 2;
	mblk_t *msg = yuv_block(size, w, h);
	ms_yuv_buf_init(buf, w, h, w, msg->b_rptr);
	return msg;
}

struct _MSYuvBufAllocator {
	int unused; /* the reference recycles blocks (msgb_allocator, <= 15 in flight); the shim just allocates */
};
MSYuvBufAllocator *ms_yuv_buf_allocator_new(void) { return (MSYuvBufAllocator *)ms_malloc0(sizeof(MSYuvBufAllocator)); }
mblk_t *ms_yuv_buf_allocator_get(MSYuvBufAllocator *obj, MSPicture *buf, int w, int h) { /* msvideo.c:298-304 */
	(void)obj;
	return ms_yuv_buf_alloc(buf, w, h);
}
void ms_yuv_buf_allocator_free(MSYuvBufAllocator *obj) { ms_free(obj); }

static MSScalerDesc *scaler_impl = NULL; /* msvideo.c:700 */
MSScalerContext *ms_scaler_create_context(int sw, int sh, MSPixFmt sf, int dw, int dh, MSPixFmt df, int flags) {
	if (!scaler_impl) {
		ms_error("No scaler implementation built-in, please supply one with ms_video_set_scaler_impl ()");
		return NULL;
	}
	return scaler_impl->create_context(sw, sh, sf, dw, dh, df, flags);
}
int ms_scaler_process(MSScalerContext *ctx, uint8_t *src[], int src_strides[], uint8_t *dst[], int dst_strides[]) {
	return scaler_impl->context_process(ctx, src, src_strides, dst, dst_strides);
}
void ms_scaler_context_free(MSScalerContext *ctx) { scaler_impl->context_free(ctx); }
void ms_video_set_scaler_impl(MSScalerDesc *desc) { scaler_impl = desc; }
MSScalerDesc *ms_video_get_scaler_impl(void) { return scaler_impl; }

/* ------------------------------------------------------------------- ticker */
typedef struct Task {
	MSFilter *f;
	MSFilterFunc fn;
	struct Task *next;
} Task;

typedef struct TickerImpl {
	/* MSTicker::lock (msticker.c:59): held by the ticker's thread while it runs its tasks and graphs (:462-493), taken by ms_ticker_attach only to
	 * splice the new graph's sources in AFTER every preprocess has run on the attaching thread (:163-181), and by ms_ticker_detach to take them out
	 * BEFORE the postprocess calls (:205-221) -- an application re-plumbs a running ticker from its own thread */
	pthread_mutex_t lock;
	pthread_mutex_t tasks_lock; /* the task list on its own (the reference leaves it unguarded: "cannot be called outside of filter's process method", msfilter.c:289-301) */
	MSFilter **filters; /* every filter of the attached graphs, in the order they were attached; NULL: a detached filter's place (compacted when a quarter is holes) */
	uint8_t *is_source; /* beside `filters`: 1 = a filter without inputs -- the step starts its walks from these alone, as MSTicker walks its execution_list of sources (msticker.c:163-166,:326-343) */
	int nfilters, cap, holes;
	Task *tasks;
	uint64_t tasks_ns, step_ns; /* the last step, by phase */
	/* MS2SHIM_PROFILE=1: the last step's process() calls by filter id (time summed, the longest single call) */
	int prof_ids[16];
	uint64_t prof_ns[16], max_call_ns;
	int max_call_id;
} TickerImpl;

static void ti_add(TickerImpl *ti, MSFilter *f) {
	if (ti->nfilters == ti->cap) {
		ti->cap = ti->cap ? 2 * ti->cap : 1024;
		ti->filters = (MSFilter **)realloc(ti->filters, sizeof(MSFilter *) * (size_t)ti->cap);
		ti->is_source = (uint8_t *)realloc(ti->is_source, (size_t)ti->cap);
	}
	SHIM_HDR(f)->idx = ti->nfilters;
	ti->is_source[ti->nfilters] = f->desc->ninputs == 0;
	ti->filters[ti->nfilters++] = f;
}

MSTicker *ms_ticker_new(void) {
	MSTicker *t = (MSTicker *)ms_malloc0(sizeof(*t));
	t->interval = 10; /* TICKER_INTERVAL msticker.c:46 */
	t->impl = ms_malloc0(sizeof(TickerImpl));
	pthread_mutex_init(&((TickerImpl *)t->impl)->lock, NULL);
	pthread_mutex_init(&((TickerImpl *)t->impl)->tasks_lock, NULL);
	return t;
}

void ms_ticker_destroy(MSTicker *t) {
	if (!t) return;
	TickerImpl *ti = (TickerImpl *)t->impl;
	while (ti->tasks) {
		Task *n = ti->tasks->next;
		free(ti->tasks);
		ti->tasks = n;
	}
	free(ti->filters);
	free(ti->is_source);
	pthread_mutex_destroy(&ti->lock);
	pthread_mutex_destroy(&ti->tasks_lock);
	free(ti);
	free(t);
}

typedef struct FList {
	MSFilter **v;
	int n, cap;
} FList;
static void find_neighbours(MSFilter *f, FList *l) { /* msfilter.c:303-344 (`seen` is its mark: clear again when the caller is done) */
	if (f->seen) return;
	f->seen = TRUE;
	if (l->n == l->cap) {
		l->cap = l->cap ? 2 * l->cap : 64;
		l->v = (MSFilter **)realloc(l->v, sizeof(MSFilter *) * (size_t)l->cap);
	}
	l->v[l->n++] = f;
	for (int i = 0; i < f->desc->ninputs; ++i)
		if (f->inputs[i]) find_neighbours(f->inputs[i]->prev.filter, l);
	for (int i = 0; i < f->desc->noutputs; ++i)
		if (f->outputs[i]) find_neighbours(f->outputs[i]->next.filter, l);
}

int ms_ticker_attach(MSTicker *t, MSFilter *f) {
	TickerImpl *ti = (TickerImpl *)t->impl;
	if (f->ticker == t) return 0; /* msticker.c:172-174: already being scheduled, nothing to do */
	FList l = {0};
	find_neighbours(f, &l);
	for (int i = 0; i < l.n; ++i) { /* ms_filter_preprocess msfilter.c:240-250, on the attaching thread, the ticker's lock NOT held */
		MSFilter *g = l.v[i];
		__atomic_store_n(&g->ticker, t, __ATOMIC_RELAXED); /* (a plain store in the reference; a neighbour's flush may be reading it: relaxed atomics on both sides keep the sanitizer on the races that matter) */
		g->last_tick = 0;
		if (g->desc->preprocess) g->desc->preprocess(g);
	}
	for (int i = 0; i < l.n; ++i) l.v[i]->seen = FALSE; /* (`seen` is find_neighbours' mark: clear outside of it) */
	pthread_mutex_lock(&ti->lock);
	for (int i = 0; i < l.n; ++i) ti_add(ti, l.v[i]);
	pthread_mutex_unlock(&ti->lock);
	free(l.v);
	return 0;
}

static void remove_tasks_for_filter(TickerImpl *ti, MSFilter *f) { /* msticker.c:314-324 */
	pthread_mutex_lock(&ti->tasks_lock);
	Task **pp = &ti->tasks;
	while (*pp) {
		if ((*pp)->f == f) {
			Task *dead = *pp;
			*pp = dead->next;
			free(dead);
		} else pp = &(*pp)->next;
	}
	f->postponed_task = 0;
	pthread_mutex_unlock(&ti->tasks_lock);
}

static void ms2shim_purge_tasks(MSFilter *f) { remove_tasks_for_filter((TickerImpl *)f->ticker->impl, f); }

int ms_ticker_detach(MSTicker *t, MSFilter *f) {
	TickerImpl *ti = (TickerImpl *)t->impl;
	/* detach the whole connected graph of f */
	FList tmp = {0};
	if (f->ticker != t) return 0; /* msticker.c:197-203: not scheduled (by this ticker): nothing to do */
	pthread_mutex_lock(&ti->lock); /* :205: the graph leaves the execution list between two ticks */
	find_neighbours(f, &tmp);
	for (int i = 0; i < tmp.n; ++i) {
		tmp.v[i]->seen = FALSE;
		if (tmp.v[i]->postponed_task) remove_tasks_for_filter(ti, tmp.v[i]); /* call_postprocess msticker.c:187-190: BEFORE postprocess (here under the lock: the task list is the ticker thread's) */
	}
	for (int i = 0; i < tmp.n; ++i) { /* their places in the ticker's array become holes */
		MSFilter *g = tmp.v[i];
		const int at = SHIM_HDR(g)->idx;
		if (at >= 0 && at < ti->nfilters && ti->filters[at] == g) ti->filters[at] = NULL, ti->is_source[at] = 0, ti->holes++;
		SHIM_HDR(g)->idx = -1;
	}
	if (ti->holes * 4 > ti->nfilters) { /* compaction, order kept */
		int j = 0;
		for (int i = 0; i < ti->nfilters; ++i)
			if (ti->filters[i]) {
				SHIM_HDR(ti->filters[i])->idx = j;
				ti->is_source[j] = ti->is_source[i];
				ti->filters[j++] = ti->filters[i];
			}
		ti->nfilters = j, ti->holes = 0;
	}
	pthread_mutex_unlock(&ti->lock);
	for (int i = 0; i < tmp.n; ++i) { /* :221: the postprocess calls, on the detaching thread, the lock released */
		MSFilter *g = tmp.v[i];
		if (g->desc->postprocess) g->desc->postprocess(g);
		g->ticker = NULL;
	}
	free(tmp.v);
	return 0;
}

void ms_filter_postpone_task(MSFilter *f, MSFilterFunc task) {
	if (!f->ticker) return;
	TickerImpl *ti = (TickerImpl *)f->ticker->impl;
	Task *t = (Task *)malloc(sizeof(*t));
	t->f = f;
	t->fn = task;
	t->next = NULL;
	pthread_mutex_lock(&ti->tasks_lock);
	Task **pp = &ti->tasks;
	while (*pp) pp = &(*pp)->next;
	*pp = t;
	f->postponed_task++;
	pthread_mutex_unlock(&ti->tasks_lock);
}

/* (both scans stop at the last CONNECTED pin, as the reference's do: an MSAudioMixer has 128 pins, a stream's local_mixer one linked) */
static int inputs_have_data(MSFilter *f) { /* ms_filter_inputs_have_data msfilter.c:277-287 */
	for (int i = 0, j = 0; i < f->desc->ninputs && j < f->n_connected_inputs; ++i)
		if (f->inputs[i]) {
			++j;
			if (!ms_queue_empty(f->inputs[i])) return 1;
		}
	return 0;
}

static int can_process(MSFilter *f, uint32_t tick) { /* filter_can_process msticker.c:230-242 */
	for (int i = 0, j = 0; i < f->desc->ninputs && j < f->n_connected_inputs; ++i)
		if (f->inputs[i]) {
			++j;
			if (f->inputs[i]->prev.filter->last_tick != tick) return 0;
		}
	return 1;
}

static uint64_t now_ns(void);
static int g_profile;
__attribute__((constructor)) static void profile_init(void) { g_profile = getenv("MS2SHIM_PROFILE") ? 1 : 0; }
static void call_process_profiled(MSFilter *f) {
	TickerImpl *ti = (TickerImpl *)f->ticker->impl;
	const uint64_t t0 = now_ns();
	f->desc->process(f);
	const uint64_t d = now_ns() - t0;
	const int id = (int)f->desc->id;
	for (int i = 0; i < 16; ++i) {
		if (ti->prof_ids[i] == id || ti->prof_ids[i] == 0) {
			ti->prof_ids[i] = id;
			ti->prof_ns[i] += d;
			break;
		}
	}
	if (d > ti->max_call_ns) ti->max_call_ns = d, ti->max_call_id = id;
}

static void call_process(MSFilter *f) { /* msticker.c:244-259 */
	if (g_profile > 0) {
		if (f->desc->ninputs == 0 || (f->desc->flags & MS_FILTER_IS_PUMP)) {
			call_process_profiled(f);
		} else {
			while (inputs_have_data(f)) {
				call_process_profiled(f);
				if (f->postponed_task) break;
			}
		}
		return;
	}
	if (f->desc->ninputs == 0 || (f->desc->flags & MS_FILTER_IS_PUMP)) {
		f->desc->process(f);
	} else {
		while (inputs_have_data(f)) {
			f->desc->process(f);
			if (f->postponed_task) break;
		}
	}
}

static void run_graph(MSFilter *f, MSTicker *t, MSFilter **unsched, int *nunsched, int force) { /* :261-282 */
	if (f->last_tick == t->ticks) return;
	if (can_process(f, t->ticks) || force) {
		f->last_tick = t->ticks;
		call_process(f);
		for (int i = 0, j = 0; i < f->desc->noutputs && j < f->n_connected_outputs; ++i)
			if (f->outputs[i]) {
				++j;
				run_graph(f->outputs[i]->next.filter, t, unsched, nunsched, force);
			}
	} else if (*nunsched < 256) {
		unsched[(*nunsched)++] = f;
	}
}

static uint64_t now_ns(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

void ms_ticker_step(MSTicker *t) {
	TickerImpl *ti = (TickerImpl *)t->impl;
	MSFilter *unsched[256];
	int nunsched = 0;
	if (g_profile > 0) {
		memset(ti->prof_ns, 0, sizeof(ti->prof_ns));
		ti->max_call_ns = 0, ti->max_call_id = 0;
	}
	const uint64_t t0 = now_ns();
	pthread_mutex_lock(&ti->lock); /* msticker.c:462-493: held while the tasks and the graphs run */
	t->ticks++;
	/* run_tasks msticker.c:301-312: postponed tasks run before the graphs */
	pthread_mutex_lock(&ti->tasks_lock);
	Task *tasks = ti->tasks;
	ti->tasks = NULL;
	for (Task *k = tasks; k; k = k->next) k->f->postponed_task--;
	pthread_mutex_unlock(&ti->tasks_lock);
	while (tasks) {
		Task *n = tasks->next;
		tasks->fn(tasks->f);
		free(tasks);
		tasks = n;
	}
	ti->tasks_ns = now_ns() - t0;
	for (int i = 0; i < ti->nfilters; ++i)
		if (ti->is_source[i]) run_graph(ti->filters[i], t, unsched, &nunsched, 0);
	/* filters inside loops: scheduled anyway on a second pass (msticker.c:284-299) */
	for (int i = 0, n = nunsched, dummy = 0; i < n; ++i) run_graph(unsched[i], t, unsched, &dummy, 1);
	__atomic_store_n(&t->time, t->time + (uint64_t)t->interval, __ATOMIC_RELAXED);
	pthread_mutex_unlock(&ti->lock);
	ti->step_ns = now_ns() - t0;
}
/* the last step: ns in the postponed tasks (the plugin's flush) and in the whole step (tasks + graph walk) */
void ms2shim_ticker_last_step(MSTicker *t, uint64_t *tasks_ns, uint64_t *step_ns) {
	TickerImpl *ti = (TickerImpl *)t->impl;
	if (tasks_ns) *tasks_ns = ti->tasks_ns;
	if (step_ns) *step_ns = ti->step_ns;
}

/* MS2SHIM_PROFILE=1: the last step's process() time by filter id (up to 16 ids; returns how many) and its longest single call */
int ms2shim_ticker_profile(MSTicker *t, int *ids, uint64_t *ns, int cap, int *max_id, uint64_t *max_ns) {
	TickerImpl *ti = (TickerImpl *)t->impl;
	int n = 0;
	for (int i = 0; i < 16 && n < cap && ti->prof_ids[i]; ++i, ++n) ids[n] = ti->prof_ids[i], ns[n] = ti->prof_ns[i];
	if (max_id) *max_id = ti->max_call_id;
	if (max_ns) *max_ns = ti->max_call_ns;
	return n;
}

/* -------------------------------------------------- test source / sink filters
 * (the role MS_VOID_SOURCE / file player / recorder play in the reference's testers) */
#define SHIM_SOURCE_ID ((MSFilterId)9001)
#define SHIM_SINK_ID ((MSFilterId)9002)
#define SHIM_PASS_ID ((MSFilterId)9003)

typedef struct {
	queue_t pending; /* blocks the test queued; one is emitted per tick ... */
	int burst;       /* ... or all of them (an RTP receiver after a network hiccup) */
	/* loop mode (rate measurements): every tick a fresh block of loop_bytes copied from a shared ring of loop_n blocks --
	 * what a decoder behind an RTP receiver does every tick, allocation included */
	const uint8_t *loop_ring;
	size_t loop_bytes;
	int loop_n, loop_at;
} SrcData;
typedef struct {
	uint8_t *buf;
	size_t len, cap;
	int nblocks;
	uint32_t last_ts;
	int discard;      /* 1: count, do not keep; 2: count and fold every byte into `sum` (a run's output as one number per sink) */
	uint64_t sum;
} SinkData;

static void src_init(MSFilter *f) {
	SrcData *d = (SrcData *)ms_malloc0(sizeof(*d));
	qinit(&d->pending);
	f->data = d;
}
static void src_uninit(MSFilter *f) {
	SrcData *d = (SrcData *)f->data;
	flushq(&d->pending, 0);
	free(d);
}
static void src_process(MSFilter *f) {
	SrcData *d = (SrcData *)f->data;
	mblk_t *m;
	if (d->loop_ring) {
		m = allocb(d->loop_bytes, 0);
		memcpy(m->b_wptr, d->loop_ring + (size_t)d->loop_at * d->loop_bytes, d->loop_bytes);
		m->b_wptr += d->loop_bytes;
		if (++d->loop_at == d->loop_n) d->loop_at = 0;
		if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
		else freemsg(m);
		return;
	}
	while ((m = getq(&d->pending)) != NULL) {
		if (f->outputs[0] && msgdsize(m) > 0) ms_queue_put(f->outputs[0], m);
		else freemsg(m); /* an empty block = a tick in which the source delivers nothing (a late packet) */
		if (!d->burst) break;
	}
}
static void sink_init(MSFilter *f) { f->data = ms_malloc0(sizeof(SinkData)); }
static void sink_uninit(MSFilter *f) {
	SinkData *d = (SinkData *)f->data;
	free(d->buf);
	free(d);
}
static void sink_process(MSFilter *f) {
	SinkData *d = (SinkData *)f->data;
	mblk_t *m;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		size_t n = msgdsize(m);
		if (d->discard) { /* rate measurements: count, do not keep */
			if (d->discard == 2) { /* ... but remember what went by: order-sensitive, byte-exact */
				for (mblk_t *c = m; c; c = c->b_cont)
					for (const uint8_t *q = c->b_rptr; q < c->b_wptr; ++q) d->sum = (d->sum ^ *q) * 1099511628211ull; /* FNV-1a */
			}
			d->len += n;
			d->last_ts = mblk_get_timestamp_info(m);
			d->nblocks++;
			freemsg(m);
			continue;
		}
		if (d->len + n > d->cap) {
			d->cap = (d->len + n) * 2 + 4096;
			d->buf = (uint8_t *)realloc(d->buf, d->cap);
		}
		for (mblk_t *c = m; c; c = c->b_cont) {
			memcpy(d->buf + d->len, c->b_rptr, (size_t)(c->b_wptr - c->b_rptr));
			d->len += (size_t)(c->b_wptr - c->b_rptr);
		}
		d->last_ts = mblk_get_timestamp_info(m);
		d->nblocks++;
		freemsg(m);
	}
}

static MSFilterDesc shim_source_desc = {SHIM_SOURCE_ID, "ShimSource", "test source", MS_FILTER_OTHER, NULL, 0, 1,
                                        src_init, NULL, src_process, NULL, src_uninit, NULL, 0};
static MSFilterDesc shim_sink_desc = {SHIM_SINK_ID, "ShimSink", "test sink", MS_FILTER_OTHER, NULL, 1, 0,
                                      sink_init, NULL, sink_process, NULL, sink_uninit, NULL, 0};

/* a CPU filter of the host application's that hands its blocks on in the walk: stands for dtmfgen, recv_tee, a CPU encoder's front .. */
static void pass_process(MSFilter *f) {
	mblk_t *m;
	while ((m = ms_queue_get(f->inputs[0])) != NULL) {
		if (f->outputs[0]) ms_queue_put(f->outputs[0], m);
		else freemsg(m);
	}
}
static MSFilterDesc shim_pass_desc = {SHIM_PASS_ID, "ShimPass", "test pass-through", MS_FILTER_OTHER, NULL, 1, 1, NULL, NULL, pass_process, NULL, NULL, NULL, 0};

void ms2shim_sink_set_discard(MSFilter *f, int on) { ((SinkData *)f->data)->discard = on; }

void ms2shim_register_test_filters(MSFactory *f) {
	ms_factory_register_filter(f, &shim_source_desc);
	ms_factory_register_filter(f, &shim_sink_desc);
	ms_factory_register_filter(f, &shim_pass_desc);
}
MSFilter *ms2shim_new_pass(MSFactory *f) { return ms_factory_create_filter(f, SHIM_PASS_ID); }
MSFilter *ms2shim_new_source(MSFactory *f) { return ms_factory_create_filter(f, SHIM_SOURCE_ID); }
MSFilter *ms2shim_new_sink(MSFactory *f) { return ms_factory_create_filter(f, SHIM_SINK_ID); }
/* MS_EQUALIZER_SET_GAIN takes a struct: spelled here, where the header is */
int ms2shim_equalizer_set_gain(MSFilter *eq, float frequency, float gain, float width) {
	MSEqualizerGain g;
	g.frequency = frequency, g.gain = gain, g.width = width;
	return ms_filter_call_method(eq, MS_EQUALIZER_SET_GAIN, &g);
}
int ms2shim_equalizer_set_active(MSFilter *eq, int active) {
	int a = active;
	return ms_filter_call_method(eq, MS_EQUALIZER_SET_ACTIVE, &a);
}
/* MS_VOLUME_SET_PEER's id carries sizeof(MSFilter): spelled here, where the header is */
int ms2shim_volume_set_peer(MSFilter *vol, MSFilter *peer) { return ms_filter_call_method(vol, MS_VOLUME_SET_PEER, peer); }
/* MS_AUDIO_FLOW_CONTROL_DROP takes a struct (what MSSpeexEC's and the sound card's drop events end up as, audiostream.c:1757-1763) */
int ms2shim_flow_control_drop(MSFilter *fc, unsigned interval_ms, unsigned drop_ms) {
	MSAudioFlowControlDropEvent ev;
	ev.flow_control_interval_ms = interval_ms, ev.drop_ms = drop_ms;
	return ms_filter_call_method(fc, MS_AUDIO_FLOW_CONTROL_DROP, &ev);
}
void ms2shim_source_set_burst(MSFilter *src, int burst) { ((SrcData *)src->data)->burst = burst; }
/* the ring stays the caller's; phase = the block the source starts with */
void ms2shim_source_set_loop(MSFilter *src, const void *ring, size_t block_bytes, int nblocks, int phase) {
	SrcData *d = (SrcData *)src->data;
	d->loop_ring = (const uint8_t *)ring;
	d->loop_bytes = block_bytes;
	d->loop_n = nblocks;
	d->loop_at = nblocks > 0 ? phase % nblocks : 0;
}
void ms2shim_source_push(MSFilter *src, const void *data, size_t nbytes) {
	SrcData *d = (SrcData *)src->data;
	mblk_t *m = allocb(nbytes, 0);
	memcpy(m->b_wptr, data, nbytes);
	m->b_wptr += nbytes;
	putq(&d->pending, m);
}
void ms2shim_source_push_ts(MSFilter *src, const void *data, size_t nbytes, uint32_t ts) {
	SrcData *d = (SrcData *)src->data;
	mblk_t *m = allocb(nbytes, 0);
	memcpy(m->b_wptr, data, nbytes);
	m->b_wptr += nbytes;
	mblk_set_timestamp_info(m, ts);
	putq(&d->pending, m);
}
/* an I420 frame as the reference's capture filters emit it: video header before b_rptr (msvideo.c:162-176) */
void ms2shim_source_push_yuv(MSFilter *src, const void *i420, int w, int h, uint32_t ts) {
	SrcData *d = (SrcData *)src->data;
	YuvBuf buf;
	mblk_t *m = ms_yuv_buf_alloc(&buf, w, h);
	memcpy(buf.planes[0], i420, (size_t)(w * ((h & 1) ? h + 1 : h) * 3) / 2);
	mblk_set_timestamp_info(m, ts);
	putq(&d->pending, m);
}
static int g_notify_count;
static unsigned g_notify_last;
static void count_notify(void *ud, MSFilter *f, unsigned int id, void *arg) {
	(void)ud, (void)f, (void)arg;
	g_notify_count++;
	g_notify_last = id;
}
void ms2shim_watch(MSFilter *f) { ms_filter_add_notify_callback(f, count_notify, NULL, TRUE); }
int ms2shim_notify_count(void) { return g_notify_count; }
unsigned ms2shim_notify_last(void) { return g_notify_last; }
uint64_t ms2shim_sink_sum(MSFilter *sink) { return ((SinkData *)sink->data)->sum; }
size_t ms2shim_sink_size(MSFilter *sink) { return ((SinkData *)sink->data)->len; }
int ms2shim_sink_blocks(MSFilter *sink) { return ((SinkData *)sink->data)->nblocks; }
uint32_t ms2shim_sink_last_ts(MSFilter *sink) { return ((SinkData *)sink->data)->last_ts; }
size_t ms2shim_sink_read(MSFilter *sink, void *dst, size_t cap) {
	SinkData *d = (SinkData *)sink->data;
	size_t n = d->len < cap ? d->len : cap;
	memcpy(dst, d->buf, n);
	memmove(d->buf, d->buf + n, d->len - n);
	d->len -= n;
	return n;
}
const char *ms2shim_filter_name(MSFilter *f) { return f->desc->name; }
unsigned ms2shim_filter_flags(MSFilter *f) { return f->desc->flags; }
unsigned ms2shim_method_id(int filter_id, int index, int argsize) { return MS_FILTER_METHOD_ID(filter_id, index, argsize); }
