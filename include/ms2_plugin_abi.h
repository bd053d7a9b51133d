/*
 * ms2_plugin_abi.h -- the slice of mediastreamer2's plugin ABI that the MI355X
 * filters are written against.
 *
 * In a real deployment define MSMI355X_USE_REAL_MS2_HEADERS and this header just
 * includes mediastreamer2's own headers (oRTP >= 5.3, bctoolbox), so the plugin
 * is compiled against the true struct layouts.  Those headers (and the libraries
 * behind them) are NOT available in this repository's build image, so by default
 * the declarations below restate, field for field and macro for macro, what the
 * plugin uses, each with the reference location it mirrors; the host shim
 * (tests/host/ms2shim.c, test infrastructure) implements the functions for the tests.
 * The layouts are ABI-plausible, not ABI-verified.
 */
#ifndef MS2_PLUGIN_ABI_H
#define MS2_PLUGIN_ABI_H

#ifdef MSMI355X_USE_REAL_MS2_HEADERS
#include <mediastreamer2/allfilters.h>
#include <mediastreamer2/flowcontrol.h>
#include <mediastreamer2/msaudiomixer.h>
#include <mediastreamer2/mschanadapter.h>
#include <mediastreamer2/msequalizer.h>
#include <mediastreamer2/msfactory.h>
#include <mediastreamer2/msfilter.h>
#include <mediastreamer2/msgenericplc.h>
#include <mediastreamer2/msinterfaces.h>
#include <mediastreamer2/msticker.h>
#include <mediastreamer2/msvideo.h>
#include <mediastreamer2/msvolume.h>
#else

#include <pthread.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned char bool_t;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

/* ---- oRTP message blocks (ortp/str_utils.h; field names as used in
 * include/mediastreamer2/msqueue.h:58-127, src/base/msqueue.c:90,110-116) ---- */
typedef struct datab {
	unsigned char *db_base;
	unsigned char *db_lim;
	void (*db_freefn)(void *);
	int db_ref;
} dblk_t;

typedef struct msgb {
	struct msgb *b_prev;
	struct msgb *b_next;
	struct msgb *b_cont;
	struct datab *b_datap;
	unsigned char *b_rptr;
	unsigned char *b_wptr;
	uint32_t reserved1; /* timestamp, msqueue.h:101-108 */
	uint32_t reserved2;
	uint8_t ttl_or_hl;
} mblk_t;

typedef struct _queue {
	mblk_t _q_stopper;
	int q_mcount;
} queue_t;

mblk_t *allocb(size_t size, int unused);
/* ortp/str_utils.h: a block whose payload is the CALLER'S buffer; freefn(buf) runs when the last reference goes (the
 * reference's capture filters hand mmap'd frames downstream this way: src/videofilters/msv4l2.c:521, msv4l.c:191) */
mblk_t *esballoc(uint8_t *buf, size_t size, int pri, void (*freefn)(void *));
void freemsg(mblk_t *m);
void freeb(mblk_t *m);
mblk_t *dupb(mblk_t *m);
mblk_t *dupmsg(mblk_t *m);
size_t msgdsize(const mblk_t *m);
void mblk_meta_copy(const mblk_t *src, mblk_t *dst);
void qinit(queue_t *q);
void putq(queue_t *q, mblk_t *m);
mblk_t *getq(queue_t *q);
mblk_t *peekq(queue_t *q);
void flushq(queue_t *q, int how);
#define qempty(q) (&(q)->_q_stopper == (q)->_q_stopper.b_next)

static inline void mblk_set_timestamp_info(mblk_t *m, uint32_t ts) { m->reserved1 = ts; }
static inline uint32_t mblk_get_timestamp_info(const mblk_t *m) { return m->reserved1; }

/* ---- MSQueue / MSBufferizer (include/mediastreamer2/msqueue.h:28-47,:131-170) ---- */
struct _MSFilter;
typedef struct _MSCPoint {
	struct _MSFilter *filter;
	int pin;
} MSCPoint;

typedef struct _MSQueue {
	queue_t q;
	MSCPoint prev;
	MSCPoint next;
} MSQueue;

static inline mblk_t *ms_queue_get(MSQueue *q) { return getq(&q->q); }
static inline void ms_queue_put(MSQueue *q, mblk_t *m) { putq(&q->q, m); }
static inline bool_t ms_queue_empty(const MSQueue *q) { return (bool_t)qempty(&q->q); }
void ms_queue_flush(MSQueue *q);

typedef struct _MSBufferizer {
	queue_t q;
	size_t size;
} MSBufferizer;
MSBufferizer *ms_bufferizer_new(void);
void ms_bufferizer_init(MSBufferizer *obj);
void ms_bufferizer_put(MSBufferizer *obj, mblk_t *m);
void ms_bufferizer_put_from_queue(MSBufferizer *obj, MSQueue *q);
size_t ms_bufferizer_read(MSBufferizer *obj, uint8_t *data, size_t datalen);
static inline size_t ms_bufferizer_get_avail(MSBufferizer *obj) { return obj->size; }
void ms_bufferizer_skip_bytes(MSBufferizer *obj, int bytes);
void ms_bufferizer_flush(MSBufferizer *obj);
void ms_bufferizer_uninit(MSBufferizer *obj);
void ms_bufferizer_destroy(MSBufferizer *obj);

/* ---- filter ids (include/mediastreamer2/allfilters.h:28-192, counted) ---- */
typedef enum MSFilterId {
	MS_FILTER_NOT_SET_ID = 0,
	MS_FILTER_PLUGIN_ID = 1,
	MS_FILTER_BASE_ID = 2,
	MS_ULAW_ENC_ID = 7,
	MS_ULAW_DEC_ID = 8,
	MS_ALAW_ENC_ID = 9,
	MS_ALAW_DEC_ID = 10,
	MS_SPEEX_EC_ID = 28,
	MS_PIX_CONV_ID = 29,
	MS_SIZE_CONV_ID = 31,
	MS_RESAMPLE_ID = 41,
	MS_VOLUME_ID = 43,
	MS_VOID_SOURCE_ID = 56,
	MS_VOID_SINK_ID = 57,
	MS_EQUALIZER_ID = 61,
	MS_CHANNEL_ADAPTER_ID = 67,
	MS_AUDIO_MIXER_ID = 68,
	MS_L16_ENC_ID = 107,
	MS_L16_DEC_ID = 108,
	MS_GENERIC_PLC_ID = 111,
	MS_AUDIO_FLOW_CONTROL_ID = 141
} MSFilterId;

/* msfilter.h:76-95 */
enum {
	MSFilterInterfaceBegin = 16384,
	MSFilterEchoCancellerInterface = 16384 + 4,
	MSFilterAudioDecoderInterface = 16384 + 7,
	MSFilterAudioEncoderInterface = 16384 + 11
};

typedef enum _MSFilterCategory { MS_FILTER_OTHER = 0, MS_FILTER_ENCODER = 1, MS_FILTER_DECODER = 2 } MSFilterCategory; /* msfilter.h:111-122 */
enum _MSFilterFlags { MS_FILTER_IS_PUMP = 1, MS_FILTER_IS_HW_ACCELERATED = 1 << 1, MS_FILTER_IS_ENABLED = 1u << 31 };

typedef void (*MSFilterFunc)(struct _MSFilter *f);                 /* msfilter.h:51 */
typedef int (*MSFilterMethodFunc)(struct _MSFilter *f, void *arg); /* msfilter.h:57 */
typedef void (*MSFilterNotifyFunc)(void *userdata, struct _MSFilter *f, unsigned int id, void *arg);

typedef struct _MSFilterMethod { /* msfilter.h:65-68 */
	unsigned int id;
	MSFilterMethodFunc method;
} MSFilterMethod;

typedef struct _MSFilterDesc { /* msfilter.h:161-178 */
	MSFilterId id;
	const char *name;
	const char *text;
	MSFilterCategory category;
	const char *enc_fmt;
	int ninputs;
	int noutputs;
	MSFilterFunc init;
	MSFilterFunc preprocess;
	MSFilterFunc process;
	MSFilterFunc postprocess;
	MSFilterFunc uninit;
	MSFilterMethod *methods;
	unsigned int flags;
} MSFilterDesc;

struct _MSTicker;
struct _MSFactory;
typedef struct _MSFilter { /* msfilter.h:186-210 */
	MSFilterDesc *desc;
	pthread_mutex_t lock;
	MSQueue **inputs;
	MSQueue **outputs;
	struct _MSFactory *factory;
	int n_connected_inputs, n_connected_outputs;
	void *data;
	struct _MSTicker *ticker;
	/* private part */
	void *notify_callbacks;
	uint32_t last_tick;
	void *stats;
	int postponed_task;
	bool_t seen;
} MSFilter;

/* include/mediastreamer2/msticker.h:73-98, field for field up to `time` (the last field a filter reads: `interval`,
 * `ticks`, `time`): ms_mutex_t / ms_cond_t / ms_thread_t are the pthread types on Linux (ortp/port.h), MSList is
 * bctbx_list_t.  What follows `time` in the reference is never touched by filters; the test runtime keeps its own state
 * behind `impl` there. */
typedef struct _MSTicker {
	pthread_mutex_t lock;  /* main lock protecting the filter execution list */
	pthread_cond_t cond;
	void *execution_list;  /* MSList * of source filters */
	void *task_list;       /* MSList * of postponed tasks (ms_filter_postpone_task) */
	pthread_t thread;
	int interval;          /* ms, TICKER_INTERVAL = 10 (msticker.c:46) */
	int exec_id;
	uint32_t ticks;
	uint64_t time;         /* ms since the ticker started */
	/* -- end of the part filters see -- */
	void *impl;
} MSTicker;

typedef struct _MSFactory MSFactory;

#define ms_filter_lock(f) pthread_mutex_lock(&(f)->lock)     /* msfilter.h:728 */
#define ms_filter_unlock(f) pthread_mutex_unlock(&(f)->lock) /* msfilter.h:729 */

/* msfilter.h:606-625 */
#define MS_FILTER_METHOD_ID(_id_, _cnt_, _argsize_)                                                      \
	(unsigned int)(((((unsigned int)(_id_)) & 0xFFFF) << 16) | (((unsigned int)(_cnt_)) << 8) |          \
	               (((unsigned int)_argsize_) & 0xFF))
#define MS_FILTER_METHOD(_id_, _count_, _argtype_) MS_FILTER_METHOD_ID(_id_, _count_, sizeof(_argtype_))
#define MS_FILTER_METHOD_NO_ARG(_id_, _count_) MS_FILTER_METHOD_ID(_id_, _count_, 0)
#define MS_FILTER_BASE_METHOD(_count_, _argtype_) MS_FILTER_METHOD_ID(MS_FILTER_BASE_ID, _count_, sizeof(_argtype_))
#define MS_FILTER_EVENT(_id_, _count_, _argtype_) MS_FILTER_METHOD_ID(_id_, _count_, sizeof(_argtype_))

/* msfilter.h:641-715 */
#define MS_FILTER_SET_SAMPLE_RATE MS_FILTER_BASE_METHOD(0, int)
#define MS_FILTER_GET_SAMPLE_RATE MS_FILTER_BASE_METHOD(1, int)
#define MS_FILTER_GET_NCHANNELS MS_FILTER_BASE_METHOD(5, int)
#define MS_FILTER_SET_NCHANNELS MS_FILTER_BASE_METHOD(6, int)
#define MS_FILTER_ADD_FMTP MS_FILTER_BASE_METHOD(7, const char) /* msfilter.h:660 */
#define MS_FILTER_ADD_ATTR MS_FILTER_BASE_METHOD(8, const char) /* msfilter.h:662 */
#define MS_AUDIO_DECODER_HAVE_PLC MS_FILTER_METHOD(MSFilterAudioDecoderInterface, 0, int) /* msinterfaces.h:214-216 */
#define MS_DECODER_HAVE_PLC MS_AUDIO_DECODER_HAVE_PLC
#define MS_AUDIO_ENCODER_GET_PTIME MS_FILTER_METHOD(MSFilterAudioEncoderInterface, 1, int) /* msinterfaces.h:285 */
#define MS_CHANNEL_ADAPTER_SET_OUTPUT_NCHANNELS MS_FILTER_METHOD(MS_CHANNEL_ADAPTER_ID, 0, int) /* mschanadapter.h:25-26 */
#define MS_CHANNEL_ADAPTER_GET_OUTPUT_NCHANNELS MS_FILTER_METHOD(MS_CHANNEL_ADAPTER_ID, 1, int)
#define MS_DEFAULT_MAX_PTIME 140 /* mscommon.h */
typedef struct _MSCngData { /* msvaddtx.h:26-29 */
	int datasize;
	uint8_t data[32];
} MSCngData;
#define MS_GENERIC_PLC_SET_CN MS_FILTER_METHOD(MS_GENERIC_PLC_ID, 0, MSCngData) /* msgenericplc.h:26 */
#define MS_FILTER_SET_OUTPUT_SAMPLE_RATE MS_FILTER_BASE_METHOD(13, int)
#define MS_FILTER_SET_OUTPUT_NCHANNELS MS_FILTER_BASE_METHOD(28, int)

/* msvolume.h:35-85 */
#define MS_VOLUME_GET MS_FILTER_METHOD(MS_VOLUME_ID, 0, float)
#define MS_VOLUME_GET_LINEAR MS_FILTER_METHOD(MS_VOLUME_ID, 1, float)
#define MS_VOLUME_SET_GAIN MS_FILTER_METHOD(MS_VOLUME_ID, 2, float)
#define MS_VOLUME_SET_PEER MS_FILTER_METHOD(MS_VOLUME_ID, 4, MSFilter)
#define MS_VOLUME_SET_EA_THRESHOLD MS_FILTER_METHOD(MS_VOLUME_ID, 5, float)
#define MS_VOLUME_SET_EA_SPEED MS_FILTER_METHOD(MS_VOLUME_ID, 6, float)
#define MS_VOLUME_SET_EA_FORCE MS_FILTER_METHOD(MS_VOLUME_ID, 7, float)
#define MS_VOLUME_ENABLE_AGC MS_FILTER_METHOD(MS_VOLUME_ID, 8, int)
#define MS_VOLUME_ENABLE_NOISE_GATE MS_FILTER_METHOD(MS_VOLUME_ID, 9, int)
#define MS_VOLUME_SET_NOISE_GATE_THRESHOLD MS_FILTER_METHOD(MS_VOLUME_ID, 10, float)
#define MS_VOLUME_SET_EA_SUSTAIN MS_FILTER_METHOD(MS_VOLUME_ID, 11, int)
#define MS_VOLUME_SET_NOISE_GATE_FLOORGAIN MS_FILTER_METHOD(MS_VOLUME_ID, 12, float)
#define MS_VOLUME_SET_DB_GAIN MS_FILTER_METHOD(MS_VOLUME_ID, 13, float)
#define MS_VOLUME_GET_GAIN MS_FILTER_METHOD(MS_VOLUME_ID, 14, float)
#define MS_VOLUME_GET_GAIN_DB MS_FILTER_METHOD(MS_VOLUME_ID, 15, float)
#define MS_VOLUME_REMOVE_DC MS_FILTER_METHOD(MS_VOLUME_ID, 16, int)
#define MS_VOLUME_SET_EA_TRANSMIT_THRESHOLD MS_FILTER_METHOD(MS_VOLUME_ID, 17, float)
#define MS_VOLUME_GET_MIN MS_FILTER_METHOD(MS_VOLUME_ID, 18, float)
#define MS_VOLUME_GET_MAX MS_FILTER_METHOD(MS_VOLUME_ID, 19, float)
#define MS_VOLUME_DB_LOWEST (-120)

/* msaudiomixer.h:26-42 */
typedef struct MSAudioMixerCtl {
	int pin;
	union param_t {
		float gain;
		int active;
		int enabled;
	} param;
} MSAudioMixerCtl;
#define MS_AUDIO_MIXER_SET_INPUT_GAIN MS_FILTER_METHOD(MS_AUDIO_MIXER_ID, 0, MSAudioMixerCtl)
#define MS_AUDIO_MIXER_SET_ACTIVE MS_FILTER_METHOD(MS_AUDIO_MIXER_ID, 1, MSAudioMixerCtl)
#define MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE MS_FILTER_METHOD(MS_AUDIO_MIXER_ID, 2, int)
#define MS_AUDIO_MIXER_SET_MASTER_CHANNEL MS_FILTER_METHOD(MS_AUDIO_MIXER_ID, 3, int)
#define MS_AUDIO_MIXER_ENABLE_OUTPUT MS_FILTER_METHOD(MS_AUDIO_MIXER_ID, 4, MSAudioMixerCtl)

/* msequalizer.h:26-53 */
typedef struct _MSEqualizerGain {
	float frequency;
	float gain;
	float width;
} MSEqualizerGain;
#define MS_EQUALIZER_SET_GAIN MS_FILTER_METHOD(MS_EQUALIZER_ID, 0, MSEqualizerGain)
#define MS_EQUALIZER_GET_GAIN MS_FILTER_METHOD(MS_EQUALIZER_ID, 1, MSEqualizerGain)
#define MS_EQUALIZER_SET_ACTIVE MS_FILTER_METHOD(MS_EQUALIZER_ID, 2, int)
#define MS_EQUALIZER_DUMP_STATE MS_FILTER_METHOD(MS_EQUALIZER_ID, 3, float)
#define MS_EQUALIZER_GET_NUM_FREQUENCIES MS_FILTER_METHOD(MS_EQUALIZER_ID, 4, int)

/* msinterfaces.h:164-183 */
#define MS_ECHO_CANCELLER_SET_DELAY MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 0, int)
#define MS_ECHO_CANCELLER_SET_FRAMESIZE MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 1, int)
#define MS_ECHO_CANCELLER_SET_TAIL_LENGTH MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 2, int)
#define MS_ECHO_CANCELLER_SET_BYPASS_MODE MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 3, bool_t)
#define MS_ECHO_CANCELLER_GET_BYPASS_MODE MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 4, bool_t)
#define MS_ECHO_CANCELLER_GET_STATE_STRING MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 5, char *)
#define MS_ECHO_CANCELLER_SET_STATE_STRING MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 6, const char)
#define MS_ECHO_CANCELLER_GET_DELAY MS_FILTER_METHOD(MSFilterEchoCancellerInterface, 7, int)

/* flowcontrol.h:61-70 */
typedef struct _MSAudioFlowControlDropEvent {
	uint32_t flow_control_interval_ms;
	uint32_t drop_ms;
} MSAudioFlowControlDropEvent;
#define MS_AUDIO_FLOW_CONTROL_DROP_EVENT MS_FILTER_EVENT(MS_AUDIO_FLOW_CONTROL_ID, 0, MSAudioFlowControlDropEvent)
/* flowcontrol.h:25-38, :75-80 */
typedef enum _MSAudioFlowControlStrategy { MSAudioFlowControlBasic, MSAudioFlowControlSoft } MSAudioFlowControlStrategy;
typedef struct _MSAudioFlowControlConfig {
	MSAudioFlowControlStrategy strategy;
	float silent_threshold;
} MSAudioFlowControlConfig;
#define MS_AUDIO_FLOW_CONTROL_SET_CONFIG MS_FILTER_METHOD(MS_AUDIO_FLOW_CONTROL_ID, 0, MSAudioFlowControlConfig)
#define MS_AUDIO_FLOW_CONTROL_DROP MS_FILTER_METHOD(MS_AUDIO_FLOW_CONTROL_ID, 1, MSAudioFlowControlDropEvent)

/* ---- video (include/mediastreamer2/msvideo.h) ---- */
#define MS_VIDEO_SIZE_CIF_W 352 /* msvideo.h:39-40 */
#define MS_VIDEO_SIZE_CIF_H 288
typedef struct MSVideoSize { /* msvideo.h:231-233 */
	int width, height;
} MSVideoSize;
typedef enum MSVideoOrientation { MS_VIDEO_LANDSCAPE = 0, MS_VIDEO_PORTRAIT = 1 } MSVideoOrientation; /* :265 */
typedef enum { /* msvideo.h:267-280 */
	MS_PIX_FMT_UNKNOWN,
	MS_YUV420P,
	MS_YUYV,
	MS_RGB24,
	MS_RGB24_REV,
	MS_MJPEG,
	MS_UYVY,
	MS_YUY2,
	MS_RGBA32,
	MS_RGB565,
	MS_H264,
	MS_RGBA32_REV
} MSPixFmt;
typedef struct _MSPicture { /* msvideo.h:282-288 */
	int w, h;
	uint8_t *planes[4];
	int strides[4];
} MSPicture;
typedef struct _MSPicture YuvBuf;
static inline MSVideoOrientation ms_video_size_get_orientation(MSVideoSize vs) { /* msvideo.h:453-455 */
	return vs.width >= vs.height ? MS_VIDEO_LANDSCAPE : MS_VIDEO_PORTRAIT;
}
#define MS_SCALER_METHOD_NEIGHBOUR 1
#define MS_SCALER_METHOD_BILINEAR (1 << 1)
typedef struct _MSScalerContext MSScalerContext;
typedef struct _MSScalerDesc { /* msvideo.h:473-478 */
	MSScalerContext *(*create_context)(int src_w, int src_h, MSPixFmt src_fmt, int dst_w, int dst_h, MSPixFmt dst_fmt,
	                                   int flags);
	int (*context_process)(MSScalerContext *ctx, uint8_t *src[], int src_strides[], uint8_t *dst[], int dst_strides[]);
	void (*context_free)(MSScalerContext *ctx);
} MSScalerDesc;
/* src/voip/msvideo.c:702-725 */
MSScalerContext *ms_scaler_create_context(int src_w, int src_h, MSPixFmt src_fmt, int dst_w, int dst_h, MSPixFmt dst_fmt,
                                          int flags);
int ms_scaler_process(MSScalerContext *ctx, uint8_t *src[], int src_strides[], uint8_t *dst[], int dst_strides[]);
void ms_scaler_context_free(MSScalerContext *ctx);
void ms_video_set_scaler_impl(MSScalerDesc *desc);
MSScalerDesc *ms_video_get_scaler_impl(void);
/* frame layout and block helpers, src/voip/msvideo.c:79-99,:101-160,:162-176,:272-304 */
void ms_yuv_buf_init(YuvBuf *buf, int w, int h, int stride, uint8_t *ptr);
int ms_yuv_buf_init_from_mblk(YuvBuf *buf, mblk_t *m);
int ms_yuv_buf_init_from_mblk_with_size(YuvBuf *buf, mblk_t *m, int w, int h);
int ms_picture_init_from_mblk_with_size(MSPicture *buf, mblk_t *m, MSPixFmt fmt, int w, int h);
mblk_t *ms_yuv_buf_alloc(YuvBuf *buf, int w, int h);
typedef struct _MSYuvBufAllocator MSYuvBufAllocator;
MSYuvBufAllocator *ms_yuv_buf_allocator_new(void);
mblk_t *ms_yuv_buf_allocator_get(MSYuvBufAllocator *obj, MSPicture *buf, int w, int h);
void ms_yuv_buf_allocator_free(MSYuvBufAllocator *obj);
/* msvideo.h:615-622 */
#define MS_FILTER_SET_VIDEO_SIZE MS_FILTER_BASE_METHOD(100, MSVideoSize)
#define MS_FILTER_GET_VIDEO_SIZE MS_FILTER_BASE_METHOD(101, MSVideoSize)
#define MS_FILTER_SET_PIX_FMT MS_FILTER_BASE_METHOD(102, MSPixFmt)
#define MS_FILTER_GET_PIX_FMT MS_FILTER_BASE_METHOD(103, MSPixFmt)
#define MS_FILTER_SET_FPS MS_FILTER_BASE_METHOD(104, float)
#define MS_FILTER_GET_FPS MS_FILTER_BASE_METHOD(105, float)
/* msfilter.h:631-633,:693-695 */
#define MS_FILTER_EVENT_NO_ARG(_id_, _count_) MS_FILTER_METHOD_ID(_id_, _count_, 0)
#define MS_FILTER_BASE_EVENT_NO_ARG(_count_) MS_FILTER_EVENT_NO_ARG(MS_FILTER_BASE_ID, _count_)
#define MS_FILTER_OUTPUT_FMT_CHANGED MS_FILTER_BASE_EVENT_NO_ARG(0)

/* ---- factory / filter API (msfactory.h, msfilter.h) ---- */
MSFactory *ms_factory_new(void);
void ms_factory_destroy(MSFactory *f);
void ms_factory_register_filter(MSFactory *f, MSFilterDesc *desc); /* src/base/msfactory.c:259-282: PREPENDS */
MSFilter *ms_factory_create_filter(MSFactory *f, MSFilterId id);   /* :417-427, first match :440-450 */
MSFilterDesc *ms_factory_lookup_filter_by_id(MSFactory *f, MSFilterId id);
MSFilterDesc *ms_factory_lookup_filter_by_name(MSFactory *f, const char *name);
/* dlopen(path), dlsym("<basename up to .so>_init"), call it -- src/base/msfactory.c:531-586 */
int ms_factory_load_plugin(MSFactory *f, const char *path);
void ms_filter_destroy(MSFilter *f);
int ms_filter_link(MSFilter *f1, int pin1, MSFilter *f2, int pin2);   /* src/base/msfilter.c:118-140 */
int ms_filter_unlink(MSFilter *f1, int pin1, MSFilter *f2, int pin2);
int ms_filter_call_method(MSFilter *f, unsigned int id, void *arg);    /* src/base/msfilter.c:171-197 */
void ms_filter_notify(MSFilter *f, unsigned int id, void *arg);
void ms_filter_notify_no_arg(MSFilter *f, unsigned int id); /* msfilter.h:725 */
void ms_filter_add_notify_callback(MSFilter *f, MSFilterNotifyFunc fn, void *userdata, bool_t synchronous);
void ms_filter_postpone_task(MSFilter *f, MSFilterFunc task);           /* src/base/msfilter.c:289-300 */

/* ticker (src/base/msticker.c): the shim exposes single steps instead of a thread */
MSTicker *ms_ticker_new(void);
void ms_ticker_destroy(MSTicker *t);
int ms_ticker_attach(MSTicker *t, MSFilter *f); /* preprocess of every reachable filter, :141-185 */
int ms_ticker_detach(MSTicker *t, MSFilter *f); /* postprocess, :187-230 */
void ms_ticker_step(MSTicker *t);               /* one 10 ms iteration of ms_ticker_run, :472-516 */

void *ms_malloc0(size_t sz);
void ms_free(void *p);
void ms_message(const char *fmt, ...);
void ms_warning(const char *fmt, ...);
void ms_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif /* MSMI355X_USE_REAL_MS2_HEADERS */

#ifdef __cplusplus
extern "C" {
#endif
/* The plugin entry point the factory's loader resolves for "libmsmi355xfilters.so"
 * (src/base/msfactory.c:549-555) and the descriptors it registers. */
void libmsmi355xfilters_init(MSFactory *factory);
extern MSFilterDesc ms_mi355x_resample_desc;    /* .id = MS_RESAMPLE_ID,    replaces src/audiofilters/msresample.c:253-263 */
extern MSFilterDesc ms_mi355x_audio_mixer_desc; /* .id = MS_AUDIO_MIXER_ID, replaces src/audiofilters/audiomixer.c:452-464 */
extern MSFilterDesc ms_mi355x_volume_desc;      /* .id = MS_VOLUME_ID,      replaces src/audiofilters/msvolume.c:538-548 */
extern MSFilterDesc ms_mi355x_equalizer_desc;   /* .id = MS_EQUALIZER_ID,   replaces src/audiofilters/equalizer.c:366-375 */
extern MSFilterDesc ms_mi355x_speex_ec_desc;    /* .id = MS_SPEEX_EC_ID,    replaces src/audiofilters/speexec.c:411-422 */
extern MSFilterDesc ms_mi355x_webrtc_aec_name_desc; /* .id = MS_FILTER_PLUGIN_ID, name "MSWebRTCAEC": only with MSMI355X_CLAIM_WEBRTC_AEC=1 (audiostream.c:2128-2158); not AEC3 */
extern MSFilterDesc ms_mi355x_size_conv_desc;   /* .id = MS_SIZE_CONV_ID,   replaces src/videofilters/sizeconv.c:221-247 */
extern MSFilterDesc ms_mi355x_pix_conv_desc;    /* .id = MS_PIX_CONV_ID,    replaces src/videofilters/pixconv.c:112-138 */
extern MSFilterDesc ms_mi355x_alaw_dec_desc;   /* .id = MS_ALAW_DEC_ID, replaces src/audiofilters/alaw.c:235-246 */
extern MSFilterDesc ms_mi355x_ulaw_dec_desc;   /* .id = MS_ULAW_DEC_ID, replaces src/audiofilters/ulaw.c (same shape) */
extern MSFilterDesc ms_mi355x_alaw_enc_desc;   /* .id = MS_ALAW_ENC_ID, replaces src/audiofilters/alaw.c:193-204 */
extern MSFilterDesc ms_mi355x_ulaw_enc_desc;   /* .id = MS_ULAW_ENC_ID */
extern MSFilterDesc ms_mi355x_l16_enc_desc;    /* .id = MS_L16_ENC_ID,  replaces src/audiofilters/l16.c:162-174 */
extern MSFilterDesc ms_mi355x_l16_dec_desc;    /* .id = MS_L16_DEC_ID,  replaces src/audiofilters/l16.c:241-252 */
extern MSFilterDesc ms_mi355x_channel_adapter_desc;    /* .id = MS_CHANNEL_ADAPTER_ID,    replaces src/audiofilters/chanadapt.c:190-203 */
extern MSFilterDesc ms_mi355x_generic_plc_desc;        /* .id = MS_GENERIC_PLC_ID,        replaces src/audiofilters/msgenericplc.c:224-248 */
extern MSFilterDesc ms_mi355x_audio_flow_control_desc; /* .id = MS_AUDIO_FLOW_CONTROL_ID, replaces src/audiofilters/flowcontrol.c:262-277 */
/* MSScalerDesc (msvideo.h:473-478) backed by the scaler / pixconv kernels; libmsmi355xfilters_init installs
 * it with ms_video_set_scaler_impl (msvideo.c:719-721), so the reference's OWN MSSizeConv / MSPixConv /
 * display filters reach the GPU too (one frame per call, synchronous, as that interface demands). */
extern MSScalerDesc ms_mi355x_scaler_desc;
/* Runs every ticker hub's staged work now.  Normally unnecessary: the facades postpone that task on their ticker
 * themselves (msfilter.c:289-300); for an application that wants the last tick's results before tearing a graph down.
 * It emits into the filters' output queues like the ticker's own task does: call it on the ticker's thread or while the
 * tickers of the graphs concerned are not running (MSQueue has no lock, src/base/msqueue.c). */
void ms_mi355x_flush(void);
/* Blocks dropped or passed through unprocessed because a HIP call failed (0 in a healthy process). */
unsigned long long ms_mi355x_late_events(void);
/* Tickers with live banks, banks alive, bank slots held by filters (any pointer may be NULL): a leak check's view. */
void ms_mi355x_runtime_stats(int *hubs, int *banks, int *slots_in_use);
/* The device of every ticker hub that has opened a context (the hubs of one process spread over MSMI355X_DEVICES,
 * default: every visible device); returns their number, fills at most `cap` entries. */
int ms_mi355x_hub_devices(int *devices, int cap);
/* Streams whose receiving side (MSAlawDec / MSUlawDec -> MSGenericPLC -> MSAudioFlowControl, src/voip/audiostream.c:1812-1824) lives
 * in one device-resident batch per ticker instead of a bank per filter. */
int ms_mi355x_recv_stats(void);
/* 1: this filter's work runs in a device-resident batch it shares with its neighbours (a fused call leg, a conference, a conference server's
 * member, a stream's receiving side); 0: in a bank of its own type (or it has not run yet); -1: not one of this plugin's filters. */
int ms_mi355x_filter_in_batch(MSFilter *f);
/* Waits for every hub's stream (tests, orderly shutdown). */
void ms_mi355x_shutdown(void);
#ifdef __cplusplus
}
#endif

#endif /* MS2_PLUGIN_ABI_H */
