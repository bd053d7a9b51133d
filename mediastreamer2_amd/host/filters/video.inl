// filters/video.inl -- MSSizeConv / MSPixConv facades and the MSScalerDesc (src/videofilters/sizeconv.c, pixconv.c, src/voip/msvideo.c).
// Part of the single translation unit filters.cpp (included inside its anonymous namespace, after the pool / hub
// infrastructure); not compiled on its own.

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
	TickerHub *hub = nullptr; // the hub whose context the objects above were created on (pinned while this lives)
};

MSScalerContext *sd_create(int sw, int sh, MSPixFmt sf, int dw, int dh, MSPixFmt df, int flags) {
	(void)flags; // bilinear either way, like yuv_create_scale_context msvideo.c:526-540
	HubLock lk((MSFilter *)nullptr); // MSScalerDesc callers have no ticker: the process-wide hub
	ScalerCtx *c = new ScalerCtx();
	c->sw = sw, c->sh = sh, c->dw = dw, c->dh = dh, c->sf = sf, c->df = df;
	if (sf == MS_YUV420P) {
		const int fmt = (df == MS_RGB24) ? MI_PIX_RGB24 : MI_PIX_I420;
		if ((df != MS_RGB24 && df != MS_YUV420P) || !g_hub.ensure_ctx() || mi_scaler_create(g_hub.ctx, sw, sh, dw, dh, fmt, &c->sc) != MI_OK) {
			ms_error("msmi355x scaler: %dx%d fmt %d -> %dx%d fmt %d unsupported: %s", sw, sh, (int)sf, dw, dh, (int)df, mi_last_error());
			delete c;
			return NULL;
		}
	} else if (pix_to_mi(sf) < 0 || sw != dw || sh != dh) {
		ms_warning("msmi355x scaler: unsupported format %d or size change on a packed source", (int)sf); // msvideo.c:574-576
		delete c;
		return NULL;
	} else if (!g_hub.ensure_ctx()) {
		delete c;
		return NULL;
	}
	c->hub = &g_hub;
	g_hub.pins++;
	return (MSScalerContext *)c;
}

int sd_process(MSScalerContext *ctx, uint8_t *src[], int src_strides[], uint8_t *dst[], int dst_strides[]) {
	ScalerCtx *c = (ScalerCtx *)ctx;
	if (!c) return -1;
	HubLock lk(c->hub);
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
	if (!c->pc[flip] && mi_pixconv_create(g_hub.ctx, c->sw, c->sh, pix_to_mi(c->sf), flip, &c->pc[flip]) != MI_OK) {
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
	HubLock lk(c->hub);
	if (c->sc) mi_scaler_destroy(c->sc);
	for (int i = 0; i < 2; ++i)
		if (c->pc[i]) mi_pixconv_destroy(c->pc[i]);
	c->hub->pins--; // the scope's end retires the hub if this was the last thing on it
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
	virtual uint8_t *stage(MSFilter *f, uint32_t ts) {
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
		if (failed) { // a broken context is not given more work: the frames are dropped (sizeconv.c:162-166)
			staged.clear();
			return;
		}
		mi_ctx *ctx = hub->ctx;
		MI_MUST(mi_copy_h2d_pinned(ctx, d_src, h_src, (size_t)n * src_pitch));
		MI_MUST(launch(n));
		MI_MUST(mi_copy_d2h_pinned(ctx, h_dst, d_dst, (size_t)n * dst_pitch));
		MI_MUST(mi_ctx_sync(ctx));
		if (failed) staged.clear(); // the frames are dropped, like a failing ms_scaler_process (sizeconv.c:162-166)
		else ready.swap(staged);
	}
	void emit(MSFilter *f, int slot) override;
	virtual const uint8_t *result(size_t k) const { return h_dst + k * dst_pitch; } // where ready[k]'s frame lies
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
		memcpy(ob.planes[0], result(k), dst_bytes);
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

// MSSizeConv's frames cross PCIe both ways and the kernel is ~100x faster than either copy: the copies are the path.  The
// frames a tick stages travel in CHUNKS through mi_scaler_pipe (upload | kernel | download on three streams): a chunk leaves
// as soon as it is full -- during the graph walk, while the other filters are still staging -- so that at the flush only
// the last chunk's download is left to wait for, and the two PCIe directions work at the same time.
constexpr int kFrameChunk = 8;
struct ScalerPool : FramePool {
	mi_scaler *sc = nullptr;
	mi_scaler_pipe *pipe = nullptr;
	int depth = 0;
	uint8_t *cur = nullptr; // the chunk being filled (pinned staging of the pipe), `cur_n` frames of it taken
	int cur_n = 0;
	std::vector<const uint8_t *> res; // ready[k]'s frame in the pipe's pinned results
	ScalerPool(int cap, int w, int h, int dw, int dh) {
		Building b(this, cap);
		// a geometry the kernels cannot take is not fatal: the bank fails and the frame is dropped with an error, as a
		// failing ms_scaler_process is in the reference (sizeconv.c:162-166)
		if (!failed && mi_scaler_create(hub->ctx, w, h, dw, dh, MI_PIX_I420, &sc) != MI_OK) {
			ms_error("MSSizeConv: %dx%d -> %dx%d: %s", w, h, dw, dh, mi_last_error());
			failed = true;
		}
		if (failed) return;
		frame_cap = frame_slots();
		depth = std::min(8, (frame_cap + kFrameChunk - 1) / kFrameChunk);
		frame_cap = std::min(frame_cap, depth * kFrameChunk); // (more than 64 frames per tick and geometry: raise kFrameChunk)
		src_bytes = mi_scaler_src_bytes(sc);
		dst_bytes = mi_scaler_dst_bytes(sc);
		out_w = dw, out_h = dh;
		if (mi_scaler_pipe_create(sc, kFrameChunk, depth, &pipe) != MI_OK) failed = mi_failed("mi_scaler_pipe_create");
	}
	~ScalerPool() override {
		if (pipe) mi_scaler_pipe_destroy(pipe);
		if (sc) mi_scaler_destroy(sc);
	}
	int launch(int) override { return MI_OK; }
	void submit_chunk() {
		if (cur && cur_n > 0) MI_MUST(mi_scaler_pipe_submit(pipe, cur_n));
		cur = nullptr, cur_n = 0;
	}
	uint8_t *stage(MSFilter *f, uint32_t ts) override {
		if ((int)staged.size() >= frame_cap) {
			ms_error("msmi355x plugin: frame pool full (%d frames per tick; raise MSMI355X_FRAME_SLOTS)", frame_cap);
			return nullptr;
		}
		if (cur && cur_n == kFrameChunk) submit_chunk(); // the previous chunk is complete (its last frame was copied in after stage() returned)
		if (!cur) {
			if (failed || mi_scaler_pipe_acquire(pipe, &cur, &src_pitch) != MI_OK) {
				failed = failed || mi_failed("mi_scaler_pipe_acquire");
				return nullptr;
			}
			cur_n = 0;
		}
		staged.push_back({f, ts});
		return cur + (size_t)(cur_n++) * src_pitch;
	}
	bool enqueue() override {
		if (!failed) submit_chunk();
		return false; // (the waits are the pipe's own: one event per chunk, in finish())
	}
	void finish() override {
		ready.clear();
		res.clear();
		const int inflight = pipe ? mi_scaler_pipe_in_flight(pipe) : 0;
		for (int b = 0; b < inflight; ++b) {
			const uint8_t *h = nullptr;
			int n = 0;
			if (mi_scaler_pipe_collect(pipe, &h, &dst_pitch, &n) != MI_OK) {
				failed = mi_failed("mi_scaler_pipe_collect");
				break;
			}
			for (int i = 0; i < n; ++i) res.push_back(h + (size_t)i * dst_pitch);
		}
		if (failed || res.size() != staged.size()) { // the frames are dropped, like a failing ms_scaler_process (sizeconv.c:162-166)
			staged.clear();
			res.clear();
		} else ready.swap(staged);
	}
	void flush() override {
		enqueue();
		finish();
	}
	const uint8_t *result(size_t k) const override { return res[k]; }
};

struct PixPool : FramePool {
	mi_pixconv *pc = nullptr;
	PixPool(int cap, int w, int h, int fmt, int flip, int ms_fmt) {
		Building b(this, cap);
		if (!failed && mi_pixconv_create(hub->ctx, w, h, fmt, flip, &pc) != MI_OK) {
			ms_error("MSPixConv: %dx%d format %d: %s", w, h, ms_fmt, mi_last_error());
			failed = true;
		}
		if (failed) return;
		frame_cap = frame_slots();
		src_bytes = mi_pixconv_src_bytes(pc);
		dst_bytes = mi_pixconv_dst_bytes(pc);
		out_w = w, out_h = h;
		alloc_buffers();
	}
	~PixPool() override {
		if (pc) mi_pixconv_destroy(pc);
	}
	int launch(int n) override { return mi_pixconv_process(pc, n, d_src, src_pitch, d_dst, dst_pitch); }
};

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
		HubLock lk(f);
		size_conv_leave_pool(s, f);
	}
	ms_yuv_buf_allocator_free(s->allocator);
	ms_free(s);
}
void size_conv_postprocess(MSFilter *f) { // :68-76 (the scaler context there == our pool membership)
	SizeConvState *s = (SizeConvState *)f->data;
	{
		HubLock lk(f);
		size_conv_leave_pool(s, f);
	}
	flushq(&s->rq, 0);
	s->frame_count = -1;
}

// get_resampler sizeconv.c:82-95: (re)join the pool of this geometry
ScalerPool *size_conv_pool(MSFilter *f, SizeConvState *s, int w, int h) {
	if (s->pool && !s->pool->failed && s->in_vsize.width == w && s->in_vsize.height == h && s->pool->hub->ticker == f->ticker &&
	    s->pool->out_w == s->target_vsize.width && s->pool->out_h == s->target_vsize.height)
		return s->pool;
	size_conv_leave_pool(s, f);
	const int dw = s->target_vsize.width, dh = s->target_vsize.height;
	s->pool = bank<ScalerPool>("scaler:" + std::to_string(w) + "x" + std::to_string(h) + ">" + std::to_string(dw) + "x" + std::to_string(dh), 1,
	                           [&](int cap) { return new ScalerPool(cap, w, h, dw, dh); });
	if (!s->pool) return nullptr;
	s->slot = s->pool->acquire(f);
	if (s->slot < 0) s->pool = nullptr;
	else note_slot(f);
	s->in_vsize.width = w;
	s->in_vsize.height = h;
	ms_message("mi355x size converter: joined the %dx%d -> %dx%d batch", w, h, dw, dh);
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
		ms_message("mi355x size converter: a frame beyond the configured fps was dropped");
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
	HubLock lk(f);
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
	HubLock lk(f); // lock order everywhere: the hub first, the filter's own lock inside it
	ms_filter_lock(f);
	for (mblk_t *m; (m = ms_queue_get(f->inputs[0])) != NULL;) putq(&s->rq, m);
	if (!size_conv_rate_gate(f, s)) {
		ms_filter_unlock(f);
		return;
	}
	for (mblk_t *im; (im = getq(&s->rq)) != NULL;) {
		YuvBuf in;
		if (ms_yuv_buf_init_from_mblk(&in, im) != 0) {
			ms_warning("mi355x size converter: the input block is no I420 frame (header / size mismatch); dropped");
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
			ms_warning("mi355x size converter: target size changed with the input orientation; frames held until the sink reconfigures");
		} else if (size_conv_stage(f, s, in, mblk_get_timestamp_info(im))) {
			staged = true;
		} else {
			ms_error("mi355x size converter: no batch for this geometry; frame dropped");
		}
		freemsg(im);
	}
	ms_filter_unlock(f);
	if (staged) request_flush(f);
}

int sizeconv_set_vsize(MSFilter *f, void *arg) { // sizeconv.c:186-197
	SizeConvState *s = (SizeConvState *)f->data;
	HubLock lk(f);
	ms_filter_lock(f);
	s->target_vsize = *(MSVideoSize *)arg;
	ms_message("mi355x size converter: target size %dx%d", s->target_vsize.width, s->target_vsize.height);
	size_conv_leave_pool(s, f);
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
		HubLock lk(f);
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
			HubLock lk(f);
			const int fmt = pix_to_mi(s->in_fmt);
			const int flip = s->in_fmt == MS_RGB24_REV; // :78-81
			if (fmt < 0 || (inbuf.w & 1)) {
				ms_error("mi355x pixel converter: format %d / width %d cannot be converted; frame dropped", (int)s->in_fmt, inbuf.w); // pixconv.c:84 logs and drops too
			} else {
				if (!s->pool || s->pool->failed || s->pool->hub->ticker != f->ticker || s->pool->out_w != inbuf.w || s->pool->out_h != inbuf.h) {
					pixconv_leave_pool(s, f);
					const int w = inbuf.w, h = inbuf.h, msfmt = (int)s->in_fmt;
					s->pool = bank<PixPool>("pixconv:" + std::to_string(w) + "x" + std::to_string(h) + ":" + std::to_string(msfmt), 1,
					                        [&](int cap) { return new PixPool(cap, w, h, fmt, flip, msfmt); });
					if (!s->pool) {
						freemsg(im);
						continue;
					}
					s->slot = s->pool->acquire(f);
					if (s->slot < 0) s->pool = nullptr;
					else note_slot(f);
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
		HubLock lk(f);
		request_flush(f);
	}
}
int pixconv_set_vsize(MSFilter *f, void *arg) { // :96-100
	((PixConvState *)f->data)->size = *(MSVideoSize *)arg;
	return 0;
}
int pixconv_set_pixfmt(MSFilter *f, void *arg) { // :102-107
	PixConvState *s = (PixConvState *)f->data;
	HubLock lk(f);
	s->in_fmt = *(MSPixFmt *)arg;
	pixconv_leave_pool(s, f);
	return 0;
}
MSFilterMethod pixconv_methods[] = {{MS_FILTER_SET_VIDEO_SIZE, pixconv_set_vsize}, // pixconv.c:109-110
                                    {MS_FILTER_SET_PIX_FMT, pixconv_set_pixfmt},
                                    {0, NULL}};
