/*
 * msmi355x.h -- C ABI of libmsmi355x: the MI355X (gfx950) batched DSP backend
 * behind mediastreamer2's audio/video filter hot path.
 *
 * One object = one filter TYPE for a whole batch of streams; one call = one
 * 10 ms tick (or one frame) of every stream in the batch = one kernel launch.
 * The per-stream MSFilter facades (mediastreamer2_amd/host/) stage their frame
 * into a batch slot and call these entry points where the reference calls its
 * per-stream inner loops:
 *
 *   mi_resampler_*  replaces speex_resampler_init / _process_int as called from
 *                   src/audiofilters/msresample.c:102-115 and :150-177
 *   mi_mixer_*      replaces the loops of mixer_process,
 *                   src/audiofilters/audiomixer.c:301-344 (accumulate :33-38,
 *                   apply_gain :46-51, channel_process_out :113-130)
 *   mi_volume_*     replaces update_energy / apply_gain and the control chain of
 *                   volume_process, src/audiofilters/msvolume.c:388-445,:480-513
 *   mi_equalizer_*  replaces equalizer_state_run -> ms_fir_mem16,
 *                   src/audiofilters/equalizer.c:263-269, src/utils/dsptools.c:253-268,
 *                   and the design path equalizer.c:147-172,:215-237
 *   mi_aec_*        replaces speex_echo_cancellation + speex_preprocess_run as
 *                   called from src/audiofilters/speexec.c:200-203,:297-298
 *   mi_scaler_*     replaces MSScalerDesc.context_process,
 *                   include/mediastreamer2/msvideo.h:473-478, src/voip/msvideo.c:542-581
 *   mi_pixconv_*    replaces the packed -> I420 conversions yuv_scale reaches for MSPixConv,
 *                   src/voip/msvideo.c:553-571 (src/videofilters/pixconv.c:62-94)
 *   mi_g711_*, mi_l16_swap, mi_chan_adapt
 *                   replace the sample loops of MSAlawDec/Enc, MSUlawDec/Enc (src/audiofilters/alaw.c,
 *                   ulaw.c, g711.c), MSL16Enc/Dec (l16.c:58-70) and MSChannelAdapter (chanadapt.c:87-121)
 *   mi_flowctl_*    replaces ms_audio_flow_controller_process, src/audiofilters/flowcontrol.c:107-152
 *   mi_plc_*        replaces the plc_context_t calls of generic_plc_process,
 *                   src/audiofilters/msgenericplc.c:59-167 (genericplc.c:83-241)
 *   mi_fifo_*       MSBufferizer (src/base/msqueue.c:70-113) for a batch of streams, on the device
 *   mi_session_*    the chained path of a batch of call legs behind one call per tick (no reference
 *                   counterpart: what a server does instead of one MSTicker graph per call)
 *
 * (all paths relative to the mediastreamer2 5.5.0 tree).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - every function returns MI_OK (0) or a negative MI_E* code;
 *     mi_last_error() gives the message for the calling thread.
 *   - pointers named d_* are DEVICE pointers (HBM), h_* are host pointers.
 *   - PCM is little-endian int16, one contiguous row per stream ("packed
 *     10 ms frames"); `stride` arguments are in samples.
 *   - launches are asynchronous on the context's HIP stream; *_host variants
 *     copy in, launch, copy out and synchronise.
 *   - there is NO CPU fallback: without a usable HIP device every create
 *     call fails with MI_ENODEV.
 */
#ifndef MSMI355X_H
#define MSMI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)  /* bad argument */
#define MI_ENODEV (-2)  /* no HIP device / HIP runtime error */
#define MI_ENOMEM (-3)
#define MI_ENOTSUP (-4) /* configuration outside what the reference path uses */

#define MSMI355X_ABI_VERSION 3

int mi_abi_version(void);
const char *mi_last_error(void);

/* ------------------------------------------------------------- context */
typedef struct mi_ctx mi_ctx;
/* HIP devices visible to the process (0 when there is none or the runtime cannot be initialised).  Every object
 * belongs to the device of the context it was created on and every entry point makes that device current for the
 * calling thread first, so one process -- one thread or many -- can drive contexts on all GPUs of a node. */
int mi_device_count(void);
/* hip_stream: an existing hipStream_t to launch on (e.g. the caller's
 * framework stream), or NULL to create a private non-blocking stream. */
int mi_ctx_create(int device, void *hip_stream, mi_ctx **out);
void mi_ctx_destroy(mi_ctx *ctx);
int mi_ctx_sync(mi_ctx *ctx);
/* Loads every kernel's code object on the context's device now.  HIP loads a code object with the first launch of one of its
 * kernels, behind a process-wide lock: a media server's first ticks -- every ticker thread's first launches -- otherwise wait for
 * it (0.1 - 0.3 s measured with 16 tickers).  The plugin calls it from libmsmi355xfilters_init for every device it may use; the
 * reference has no counterpart (its filters' code is mapped with the library, src/base/msfactory.c:546). */
int mi_warmup(mi_ctx *ctx);
void *mi_ctx_stream(mi_ctx *ctx);
int mi_ctx_device(mi_ctx *ctx);
/* device properties the bench reports: CU count, HBM bytes, name */
int mi_ctx_props(mi_ctx *ctx, int *cu_count, size_t *hbm_bytes, char *name, int name_cap);

void *mi_dev_alloc(mi_ctx *ctx, size_t bytes);   /* NULL on failure */
void mi_dev_free(mi_ctx *ctx, void *d_ptr);
void *mi_host_alloc(mi_ctx *ctx, size_t bytes);  /* pinned host memory */
void mi_host_free(mi_ctx *ctx, void *h_ptr);    /* ctx may be NULL (a buffer that outlived its context: blocks still held downstream) */
int mi_copy_h2d(mi_ctx *ctx, void *d_dst, const void *h_src, size_t bytes); /* async on ctx stream */
int mi_copy_d2h(mi_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* async on ctx stream */
/* the same for host memory from mi_host_alloc, as a kernel of this library on the context's stream (a launch, never a call
 * into the runtime's copy path: what the plugin's tick path uses). */
int mi_copy_h2d_pinned(mi_ctx *ctx, void *d_dst, const void *h_pinned_src, size_t bytes);
int mi_copy_d2h_pinned(mi_ctx *ctx, void *h_pinned_dst, const void *d_src, size_t bytes);
int mi_memset(mi_ctx *ctx, void *d_dst, int value, size_t bytes);

/* hipGraph capture of everything the mi_* calls enqueue on the context stream
 * between begin and end (a tick's worth of launches, or K ticks): replaying the
 * graph removes the per-launch host cost from launch-bound inner loops.  The
 * captured calls must not synchronise or allocate (steady-state process calls
 * do not). */
typedef struct mi_graph mi_graph;
int mi_ctx_capture_begin(mi_ctx *ctx);
int mi_ctx_capture_end(mi_ctx *ctx, mi_graph **out);
int mi_graph_launch(mi_graph *g);
void mi_graph_destroy(mi_graph *g);

/* HIP-event stopwatch on the context stream (what bench.py's roofline uses) */
int mi_timer_start(mi_ctx *ctx);
int mi_timer_stop(mi_ctx *ctx, float *elapsed_ms); /* synchronises the stop event */

/* ----------------------------------------------------------- resampler */
/* A batch of `nstreams` mono resamplers with one (in_rate, out_rate, quality).
 * quality: 3 (SPEEX_RESAMPLER_QUALITY_VOIP, the only one msresample.c:104 uses
 * on x86) or 4.  State per stream: filt_len-1 input samples of history and
 * the fractional read position, zero-initialised like speex_resampler_init. */
typedef struct mi_resampler mi_resampler;
int mi_resampler_create(mi_ctx *ctx, int nstreams, uint32_t in_rate, uint32_t out_rate, int quality,
                        mi_resampler **out);
void mi_resampler_destroy(mi_resampler *r);
int mi_resampler_reset(mi_resampler *r, int first, int count);
/* one stream's running state -- (last_sample, frac) and the history of input samples: what a speex resampler handle carries from
 * call to call (msresample.c:117-120 keeps its handle across a detach) -- to and from host memory; mi_resampler_state_bytes(r)
 * bytes, valid between resamplers of the same rates and quality.  Both wait for the context's stream. */
int mi_resampler_state_bytes(const mi_resampler *r);
int mi_resampler_get_state(mi_resampler *r, int stream, void *h_state, size_t cap);
int mi_resampler_set_state(mi_resampler *r, int stream, const void *h_state, size_t bytes);
/* ... of streams [first, first + count), mi_resampler_state_bytes() each, back to back: one round trip for a conference's members */
int mi_resampler_get_states(mi_resampler *r, int first, int count, void *h_states, size_t cap);
int mi_resampler_set_states(mi_resampler *r, int first, int count, const void *h_states, size_t bytes);
/* msresample.c:151-152: in_len*out_rate/in_rate + 1 */
int mi_resampler_out_capacity(const mi_resampler *r, int in_len);
/* filter facts for tests/DESIGN: taps per phase, phases (den_rate), num_rate,
 * 1 if the direct polyphase table is in use, 0 for the interpolated table */
int mi_resampler_info(const mi_resampler *r, int *filt_len, int *den_rate, int *num_rate, int *direct);
int mi_resampler_get_table(const mi_resampler *r, float *h_dst, int cap); /* returns length */
/* One input block of in_len samples per stream (msresample.c:150-177 body).
 * d_in  [nstreams][in_stride]  int16
 * d_out [nstreams][out_stride] int16, out_stride >= mi_resampler_out_capacity
 * d_out_len [nstreams] int32 produced counts, may be NULL */
int mi_resampler_process(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                         int out_stride, int32_t *d_out_len);
int mi_resampler_process_host(mi_resampler *r, const int16_t *h_in, int in_len, int in_stride,
                              int16_t *h_out, int out_stride, int32_t *h_out_len);
/* same, but only streams with d_run[s] != 0 consume a block this tick (idle streams keep
 * their state and their output row is left untouched); d_run == NULL runs every stream */
int mi_resampler_process_masked(mi_resampler *r, const int16_t *d_in, int in_len, int in_stride, int16_t *d_out,
                                int out_stride, int32_t *d_out_len, const uint8_t *d_run);

/* --------------------------------------------------------------- mixer */
/* `nconf` MSAudioMixer instances with up to `max_members` (<= 50,
 * audiomixer.c:29) linked pins each and `nsamples` samples per tick
 * (audiomixer.c:190 bytespertick/2). */
typedef struct mi_mixer mi_mixer;
#define MI_MIXER_MAX_CHANNELS 50
#define MI_MIX_LINKED 1u   /* pin is connected (f->inputs[i] != NULL) */
#define MI_MIX_ACTIVE 2u   /* MS_AUDIO_MIXER_SET_ACTIVE, audiomixer.c:384-393 */
#define MI_MIX_OUTPUT 4u   /* MS_AUDIO_MIXER_ENABLE_OUTPUT, audiomixer.c:395-408 */
int mi_mixer_create(mi_ctx *ctx, int nconf, int max_members, int nsamples, mi_mixer **out);
void mi_mixer_destroy(mi_mixer *m);
/* h_flags / h_gain [nconf][max_members]; NULL keeps the current values.
 * Defaults: LINKED|ACTIVE|OUTPUT, gain 1.0 (channel_init audiomixer.c:65-71). */
int mi_mixer_set_controls(mi_mixer *m, const uint8_t *h_flags, const float *h_gain);
/* One tick.  d_in [nconf][max_members][nsamples]; d_has_data [nconf][max_members]
 * (0 = the bufferizer read came up short -> zeros, audiomixer.c:88), NULL = all 1.
 * conf_mode 1 (MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE): d_out [nconf][max_members][nsamples],
 * rows of unlinked / output-disabled pins are not written.
 * conf_mode 0: d_out [nconf][nsamples] (the one block dupb'd to every output). */
int mi_mixer_process(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int conf_mode,
                     int16_t *d_out);
int mi_mixer_process_host(mi_mixer *m, const int16_t *h_in, const uint8_t *h_has_data, int conf_mode,
                          int16_t *h_out);
/* d_run [nconf]: conferences with 0 are skipped this tick (bypass mode / nothing to mix,
 * audiomixer.c:244-286); d_conf_mode [nconf] per-conference mode, NULL -> `conf_mode` for all.
 * d_out is always [nconf][max_members][nsamples]; a non-conference mixer writes row 0. */
int mi_mixer_process_masked(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int conf_mode,
                            const uint8_t *d_conf_mode, int16_t *d_out, const uint8_t *d_run);
/* Split form for a conference whose members live on several GPUs:
 * partial int32 sums of the LOCAL members, an int32 all-reduce by the caller
 * (RCCL), then the local outputs from the global sum. */
int mi_mixer_partial_sum(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, int32_t *d_sum);
int mi_mixer_finalize(mi_mixer *m, const int16_t *d_in, const uint8_t *d_has_data, const int32_t *d_sum,
                      int conf_mode, int16_t *d_out);

/* ------------------------------------------------------------ exchange */
/* The one cross-GPU step of the path: the int32 all-reduce between mi_mixer_partial_sum and mi_mixer_finalize when a
 * conference's members live on several GPUs (audiomixer.c:304-314 across devices), directly on RCCL (xGMI
 * inside a node).  One exchange rank per context; the ranks may be threads of one process or processes.
 *   rank 0:      mi_exchange_unique_id(id) and hands the 128 bytes to the others (any transport);
 *   every rank:  mi_exchange_create(ctx, nranks, rank, id, &x)   -- returns when all ranks have joined;
 *   every tick:  mi_mixer_partial_sum(.., d_sum); mi_exchange_allreduce_i32(x, d_sum, n); mi_mixer_finalize(.., d_sum, ..)
 * The collective is enqueued on the context's own stream: stream order puts it after the partial sums and before the
 * finalize, nothing blocks the host.  librccl is loaded on first use; any RCCL failure is MI_ENODEV with RCCL's message
 * in mi_last_error() -- there is no fallback transport. */
typedef struct mi_exchange mi_exchange;
#define MI_EXCHANGE_ID_BYTES 128
int mi_exchange_unique_id(void *id_out, size_t cap);
int mi_exchange_create(mi_ctx *ctx, int nranks, int rank, const void *unique_id, mi_exchange **out);
void mi_exchange_destroy(mi_exchange *x);
int mi_exchange_ranks(const mi_exchange *x, int *nranks, int *rank);
int mi_exchange_allreduce_i32(mi_exchange *x, int32_t *d_buf, size_t count); /* in place, sum over the ranks */

/* -------------------------------------------------------------- volume */
typedef struct mi_volume mi_volume;
/* user-settable fields of struct Volume (msvolume.c:48-86), set by the
 * MS_VOLUME_* methods (msvolume.c:516-536) */
typedef struct mi_volume_params {
	float static_gain;                                /* MS_VOLUME_SET_GAIN / _DB_GAIN */
	float vol_upramp, vol_fast_upramp, vol_downramp;  /* MS_VOLUME_SET_EA_SPEED */
	float ea_thres, ea_transmit_thres, force;         /* MS_VOLUME_SET_EA_* */
	int32_t sustain_time;                             /* MS_VOLUME_SET_EA_SUSTAIN */
	int32_t ng_cut_time;
	float ng_threshold, ng_floorgain;                 /* MS_VOLUME_SET_NOISE_GATE_* */
	int32_t agc_enabled, noise_gate_enabled, remove_dc;
	int32_t peer; /* index (same batch) of the MS_VOLUME_SET_PEER filter, -1 = none, MI_VOLUME_PEER_EXTERNAL = see mi_volume_set_peer_batch */
} mi_volume_params;
#define MI_VOLUME_PEER_EXTERNAL (-2)
/* running state (msvolume.c:49-53,:58-62,:79) -- read back every tick for
 * MS_VOLUME_GET / GET_LINEAR / GET_MIN / GET_MAX */
typedef struct mi_volume_state {
	float energy, level_pk, instant_energy, lt_speaker_en;
	float gain, target_gain, ng_gain;
	int32_t dc_offset, sustain_dur, ng_noise_dur, fast_upramp;
} mi_volume_state;
int mi_volume_create(mi_ctx *ctx, int nstreams, int sample_rate, mi_volume **out);
void mi_volume_destroy(mi_volume *v);
void mi_volume_default_params(mi_volume_params *p);          /* volume_init msvolume.c:88-118 */
int mi_volume_set_params(mi_volume *v, int first, int count, const mi_volume_params *h_params);
/* The echo limiter's peer in ANOTHER batch (msvolume.c:201-238 reads `((Volume *)v->peer->data)->energy`, whatever filter that is):
 * stream s of `v` whose params.peer is MI_VOLUME_PEER_EXTERNAL reads the smoothed energy of stream s of `peers`, as the last launch on
 * `peers` left it (launch order on the context's stream decides; both on one context).  NULL = none.  The plugin's fused call leg keeps
 * volsend in its chain's batch and meters volrecv in a batch beside it (leg_chain.inl).  Destroying `peers` first is allowed: `v` loses
 * the link, and a stream whose peer is EXTERNAL while no peer batch is set has NO peer (no limiter), as with -1. */
int mi_volume_set_peer_batch(mi_volume *v, mi_volume *peers);
int mi_volume_get_state(mi_volume *v, int first, int count, mi_volume_state *h_state); /* syncs */
/* the same read-back enqueued on the context's stream behind the launches so far (h_state: pinned, mi_host_alloc); valid
 * after the next mi_ctx_sync -- what a per-tick flush uses so that the meters cost no synchronisation of their own */
int mi_volume_get_state_async(mi_volume *v, int first, int count, mi_volume_state *h_state);
int mi_volume_set_state(mi_volume *v, int first, int count, const mi_volume_state *h_state);
/* MS_VOLUME_GET_MAX (linear): the maximum of the smoothed energy over the last second, recorded on the device by every
 * process call like ortp_extremum_record_max in update_energy (msvolume.c:115,:143-148,:404) -- a peak between two polls
 * is not missed.  A stream's own chunks are its clock (one chunk per tick).  Syncs.  reset_max starts the windows over
 * (a new filter). */
int mi_volume_get_max(mi_volume *v, int first, int count, float *h_max);
int mi_volume_reset_max(mi_volume *v, int first, int count);
/* One chunk per stream, in place: d_samples [nstreams][stride].
 * d_nsamples [nstreams] int32 per-stream chunk length (0 = stream idle this
 * tick), or NULL -> every stream processes `nsamples`. */
int mi_volume_process(mi_volume *v, int16_t *d_samples, int nsamples, int stride, const int32_t *d_nsamples);
int mi_volume_process_host(mi_volume *v, int16_t *h_samples, int nsamples, int stride,
                           const int32_t *h_nsamples);
/* The chunk comes out of a device FIFO (mi_fifo, below): stream s pops nsamples from f_src -- all or nothing, a stream
 * holding less gets silence, like mi_fifo_pop with zero_fill -- and the processed chunk is written to row s of d_out.
 * Equals mi_fifo_pop + mi_volume_process without the separate launch and copy.  Multiples of 8 samples throughout. */
struct mi_fifo;
int mi_volume_process_fifo(mi_volume *v, struct mi_fifo *f_src, int16_t *d_out, int nsamples, int stride);
/* flags: MI_VOLMIX_DRY_SKIPS (below) -- a stream whose queue holds less than the chunk is left alone: no meter update, no gain
 * ramp, its row of d_out is not written (MSVolume finds no whole 10 ms chunk in its bufferizer, msvolume.c:480-486) */
int mi_volume_process_fifo_flags(mi_volume *v, struct mi_fifo *f_src, int16_t *d_out, int nsamples, int stride, unsigned flags);
/* the same for streams [first, first + count) only (rows of d_out are still indexed by stream) */
int mi_volume_process_fifo_range(mi_volume *v, struct mi_fifo *f_src, int16_t *d_out, int nsamples, int stride, int first, int count);
/* MSVolume + MSAudioMixer of whole conferences in ONE launch (the chain's last two filters): pin k of conference c is
 * stream first_stream + c * max_members + k of the volume batch; its chunk of `nsamples` (the mixer's tick) is popped from
 * f_src as in mi_volume_process_fifo, metered and levelled, and the conference is mixed in conference mode from the
 * levelled chunks: d_out [nconf][max_members][nsamples] as mi_mixer_process(.., conf_mode 1, ..) writes it.  The levelled
 * audio never leaves the chip: a leg's tick crosses HBM twice instead of four times.  Results equal
 * mi_volume_process_fifo + mi_mixer_process bit for bit (samples, meter state, FIFO positions).  Ticks and FIFO capacities
 * in multiples of 8 samples. */
struct mi_mixer;
int mi_mixer_process_volume_fifo(struct mi_mixer *m, mi_volume *v, int first_stream, struct mi_fifo *f_src, int16_t *d_out);
/* flags: MI_VOLMIX_DRY_SKIPS -- a pin whose queue holds less than a tick is not metered at all (its meter state, gain ramp
 * and one-second window stay as they are) and contributes silence: what the reference's chain does when MSVolume finds no
 * whole 10 ms chunk in its bufferizer (msvolume.c:480-486: the loop does not run) and the mixer then reads nothing from
 * that pin (audiomixer.c:88).  Without the flag the dry pin is metered on a tick of silence (mi_fifo_pop with zero_fill).
 * The plugin's fused call-leg chain (mediastreamer2_amd/host/filters/leg_chain.inl) sets it. */
#define MI_VOLMIX_DRY_SKIPS 1u
/* d_run (nullable) [nconf]: conferences with 0 do not tick in this launch -- nothing is popped, metered or written for them
 * (a mixer that is not due: mixer_check_bypass found nobody contributing, audiomixer.c:244-286; a conference slot not in use) */
int mi_mixer_process_volume_fifo_flags(struct mi_mixer *m, mi_volume *v, int first_stream, struct mi_fifo *f_src, int16_t *d_out, unsigned flags,
                                       const uint8_t *d_run);

/* ----------------------------------------------------------- equalizer */
typedef struct mi_equalizer mi_equalizer;
int mi_equalizer_create(mi_ctx *ctx, int nstreams, int sample_rate, mi_equalizer **out);
void mi_equalizer_destroy(mi_equalizer *e);
int mi_equalizer_fir_len(const mi_equalizer *e); /* 128 / 256 / 512, equalizer.c:60-66 */
/* MS_EQUALIZER_SET_GAIN on one stream (equalizer.c:147-172); marks taps stale */
int mi_equalizer_set_gain(mi_equalizer *e, int stream, float freq_hz, float gain, float width_hz);
/* MS_FILTER_SET_SAMPLE_RATE semantics for one stream's gains: flatten (A14) */
int mi_equalizer_flatten(mi_equalizer *e, int stream);
int mi_equalizer_set_active(mi_equalizer *e, int stream, int active); /* MS_EQUALIZER_SET_ACTIVE */
/* Designs the taps of every stream whose gains changed and sends them up NOW (the launches do it themselves when they find stale streams: a
 * host FFT per stream).  For whoever sets many streams' gains at once and wants that cost on the thread that does it -- the plugin calls it
 * where a leg joins a bank, on the attaching thread.  Waits for the stream. */
int mi_equalizer_prepare(mi_equalizer *e);
/* MS_EQUALIZER_DUMP_STATE (equalizer.c:317-328): nfft/2 floats */
int mi_equalizer_dump(mi_equalizer *e, int stream, float *h_dst, int cap);
int mi_equalizer_get_taps(mi_equalizer *e, int stream, float *h_dst, int cap); /* designs if stale */
int mi_equalizer_set_taps(mi_equalizer *e, int stream, const float *h_taps, int n);
/* The FIR's memory (ms_fir_mem16's `mem`, equalizer.c:256-268: the last fir_len - 1 input samples; it lives as long as the filter):
 * fir_len int16 to and from host memory, so that a stream can move between batches and carry on sample for sample (the plugin's fused
 * call leg takes its mic_equalizer along).  h_hist == NULL on set: cleared, as a new filter's.  Syncs. */
int mi_equalizer_get_history(mi_equalizer *e, int stream, int16_t *h_hist, int n);
int mi_equalizer_set_history(mi_equalizer *e, int stream, const int16_t *h_hist, int n);
/* in place; designs + uploads stale taps first (equalizer.c:265) */
int mi_equalizer_process(mi_equalizer *e, int16_t *d_samples, int nsamples, int stride);
int mi_equalizer_process_host(mi_equalizer *e, int16_t *h_samples, int nsamples, int stride);
/* d_nsamples [nstreams]: per-stream block length this tick (0 = idle), NULL -> nsamples for all */
int mi_equalizer_process_masked(mi_equalizer *e, int16_t *d_samples, int nsamples, int stride,
                                const int32_t *d_nsamples);

/* ----------------------------------------------------------------- AEC */
typedef struct mi_aec mi_aec;
/* speex_ec_preprocess sizing, speexec.c:171-180,:194-203 */
int mi_aec_framesize(int framesize_at_8000, int sample_rate);
int mi_aec_create(mi_ctx *ctx, int nstreams, int sample_rate, int frame_size, int filter_length,
                  mi_aec **out);
void mi_aec_destroy(mi_aec *a);
int mi_aec_reset(mi_aec *a, int first, int count);
/* One frame (frame_size samples) per stream: d_mic = near-end + echo
 * (speexec.c `echo`), d_ref = delayed far-end (`ref`), d_out = cleaned.
 * d_run [nstreams] u8: 0 = this stream has no full frame this tick (A23), NULL = all run.
 * flags bit0: also run the residual-echo/denoise post-filter (speex_preprocess_run). */
#define MI_AEC_POSTFILTER 1u
int mi_aec_process(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride,
                   const uint8_t *d_run, unsigned flags);
/* The frames of one TICK in one launch (the form the chained path uses): row s of d_mic / d_ref / d_out holds up to
 * max_frames frames back to back (stride >= max_frames * frame_size), d_count[s] (0 .. max_frames) says how many the
 * stream has ready -- at 48 kHz the filter's 256-sample frames make that one or two per 10 ms tick (speexec.c:256 runs
 * its while loop that often).  Results equal d_count[s] consecutive mi_aec_process calls; the per-stream state crosses
 * HBM once per tick instead of once per frame and the foreground filter is streamed once for both frames. */
#define MI_AEC_MAX_TICK_FRAMES 2
int mi_aec_process_frames(mi_aec *a, const int16_t *d_mic, const int16_t *d_ref, int16_t *d_out, int stride,
                          const uint8_t *d_count, int max_frames, unsigned flags);
/* The same with the FIFOs folded in (what mi_session runs): stream s's new microphone block (row s of d_mic_tick,
 * tick_len samples) and far-end block are appended to f_mic / f_ref, every whole frame f_mic then holds (<= max_frames)
 * is cancelled against the frame f_ref supplies -- silence where it cannot (speexec.c:261-272) --, and the cleaned frames
 * are appended to f_out: ms_bufferizer_put x 2, the while loop of speexec.c:256-305 and the queue put of :303, one
 * launch.  Results equal mi_fifo_push x 2, mi_fifo_pop_frames x 2, mi_aec_process_frames, mi_fifo_push_frames.
 * FIFO capacities must be multiples of the frame size.  d_ref_len (nullable): per-stream length of the far-end block
 * this tick, 0 = the far end delivered nothing (then, or later, the canceller runs on injected silence); NULL = tick_len
 * for all.  d_count_out (nullable): frames each stream ran. */
typedef struct mi_fifo mi_fifo;
int mi_aec_process_fifos(mi_aec *a, mi_fifo *f_mic, const int16_t *d_mic_tick, int mic_stride, mi_fifo *f_ref,
                         const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, int tick_len, mi_fifo *f_out,
                         int max_frames, unsigned flags, uint8_t *d_count_out);
/* mi_aec_process_fifos with the leg's MSResample folded into the same launch (the chain's first two filters as one kernel):
 * row s of d_mic_in holds in_len samples at the resampler's input rate; the wavefront that serves leg s up-samples them
 * itself -- the resampler's own tile FIR on its own history and table: the samples queued are bit for bit those of
 * mi_resampler_process, whose state advances as if it had been called -- and carries on as mi_aec_process_fifos with a
 * block of in_len * (out_rate / in_rate) samples.  The up-sampled block touches HBM once (into the FIFO) instead of three
 * times.  For integer up-sampling ratios at quality 3 with in_len * ratio / 8 <= 64 (16k -> 48k, 8k -> 48k, 8k -> 16k ticks);
 * MI_ENOTSUP otherwise: call mi_resampler_process and mi_aec_process_fifos then. */
int mi_aec_process_fifos_resampled(mi_aec *a, mi_resampler *rs, const int16_t *d_mic_in, int in_len, int in_stride, mi_fifo *f_mic,
                                   mi_fifo *f_ref, const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, mi_fifo *f_out,
                                   int max_frames, unsigned flags, uint8_t *d_count_out);
/* Both forms with a per-leg gate on the microphone block: d_mic_gate[s] == 0 = leg s is handed NO microphone block by this
 * launch (nothing is queued for it, the folded resampler's state stays as it is; frames its queue still holds run all the
 * same); NULL = every leg gets one.  What a host runtime needs whose legs do not all deliver a block every tick (packets
 * of 20 ms: two launches in one tick and none in the next; slots of a bank that are not in use). */
int mi_aec_process_fifos_masked(mi_aec *a, mi_fifo *f_mic, const int16_t *d_mic_tick, int mic_stride, mi_fifo *f_ref,
                                const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, int tick_len, mi_fifo *f_out,
                                int max_frames, unsigned flags, uint8_t *d_count_out, const uint8_t *d_mic_gate);
int mi_aec_process_fifos_resampled_masked(mi_aec *a, mi_resampler *rs, const int16_t *d_mic_in, int in_len, int in_stride, mi_fifo *f_mic,
                                          mi_fifo *f_ref, const int16_t *d_ref_tick, int ref_stride, const int32_t *d_ref_len, mi_fifo *f_out,
                                          int max_frames, unsigned flags, uint8_t *d_count_out, const uint8_t *d_mic_gate);
/* Spreading the load of mi_aec_process_fifos over the ticks.  Ticks of tick_len samples against frames of frame_size
 * leave a leg's microphone FIFO at a level that cycles through the multiples of gcd(tick_len, frame_size) -- at 48 kHz
 * (480 / 256) eight levels -- and a leg has one frame less to cancel in the tick it passes level 0.  Legs that start
 * together pass it together (seven heavy ticks, one light).  mi_aec_stagger_fifos gives streams [first, first + count)
 * -- freshly created or reset, both FIFOs empty or holding only a delay line -- a lead of unit * phase(stream) samples of
 * silence in BOTH queues (so the echo path the canceller sees is unchanged; the leg's audio is `lead` samples later):
 * every tick then carries the same share of light legs.  phase(stream) = mi_fifo_phase_of(stream, phases), a bijective
 * hash of the slot index, so any regular arrangement of slots still gets all phases.  The output FIFO of a staggered leg
 * can stand one frame fuller: size it tick_len * 2 + frame_size * 3 or more.  (The launch itself serves the legs in an
 * order sorted by the frames they have -- long ones first, evenly over the XCDs -- whatever their phases are; that needs
 * no call.)  mi_aec_stagger_info: the lead's unit and the number of phases for this frame size and tick. */
int mi_aec_stagger_info(const mi_aec *a, int tick_len, int *unit, int *phases);
int mi_aec_stagger_fifos(mi_aec *a, mi_fifo *f_mic, mi_fifo *f_ref, int tick_len, int first, int count);
int mi_aec_process_host(mi_aec *a, const int16_t *h_mic, const int16_t *h_ref, int16_t *h_out, int stride,
                        const uint8_t *h_run, unsigned flags);
/* state bytes per stream (for DESIGN/roofline accounting) */
size_t mi_aec_state_bytes(const mi_aec *a);
/* One stream's whole state as a host blob, and back: what fetch_config / apply_config (src/audiofilters/speexec.c:119-167)
 * do with SPEEX_ECHO_GET_BLOB / SET_BLOB of the reference's speex fork, so a converged canceller survives the end of a
 * call (MS_ECHO_CANCELLER_GET/SET_STATE_STRING).  The fork's blob format is not published, so the two are NOT
 * interchangeable: a string saved by the reference's MSSpeexEC is refused here with a clear message (and vice versa the
 * reference refuses ours: speex_echo_state_blob_new_from_memory fails on it) and the canceller starts from scratch, as
 * after any failed restore (speexec.c:137-139).  Format, version MI_AEC_BLOB_VERSION, little-endian:
 *   char magic[4] = "MIEC"; uint32 version, rate, frame_size F, blocks M, N = 2F, small_state_floats, scalar_bytes;
 *   float X[(M+1) N] (ring, bin-interleaved), W[M N], foreground[M N], small_state[19 F + 192]; scalar record.
 * Import checks tag, version, rate / frame / tail and the size, and refuses anything else; a restored stream continues
 * bit for bit. */
#define MI_AEC_BLOB_VERSION 3u
size_t mi_aec_blob_bytes(const mi_aec *a);
int mi_aec_export_state(mi_aec *a, int stream, void *h_blob, size_t cap);
int mi_aec_import_state(mi_aec *a, int stream, const void *h_blob, size_t size);
/* The state of `count` streams of `src` from src_first on, copied on the device into `dst` from dst_first on (same rate /
 * frame / tail, same device; src == dst allowed for disjoint ranges): a canceller converged for an endpoint serves as the
 * starting point of other legs of that endpoint without the host round trip of export / import.  Ordered after what was
 * enqueued on src's stream; asynchronous on dst's. */
int mi_aec_copy_state(mi_aec *dst, int dst_first, const mi_aec *src, int src_first, int count);
/* debug/parity read-back of one stream's float arrays: "W","foreground","X","power" ...; "counters": foreground updates,
 * background resets, state resets, frames cancelled since the stream's last reset */
int mi_aec_get(mi_aec *a, int stream, const char *what, float *h_dst, int cap);

/* -------------------------------------------------------------- scaler */
typedef struct mi_scaler mi_scaler;
#define MI_PIX_I420 0  /* MS_YUV420P, layout of ms_yuv_buf_init msvideo.c:85-99 */
#define MI_PIX_RGB24 1 /* MS_RGB24: R,G,B bytes, stride 3*w */
/* MSScalerDesc.create_context (msvideo.h:474-475); src is always I420 */
int mi_scaler_create(mi_ctx *ctx, int src_w, int src_h, int dst_w, int dst_h, int dst_fmt, mi_scaler **out);
void mi_scaler_destroy(mi_scaler *s);
size_t mi_scaler_src_bytes(const mi_scaler *s);
size_t mi_scaler_dst_bytes(const mi_scaler *s);
/* MSScalerDesc.context_process for `nframes` frames of a batch:
 * frame i at d_src + i*src_pitch, result at d_dst + i*dst_pitch. */
int mi_scaler_process(mi_scaler *s, int nframes, const uint8_t *d_src, size_t src_pitch, uint8_t *d_dst,
                      size_t dst_pitch);
int mi_scaler_process_host(mi_scaler *s, int nframes, const uint8_t *h_src, size_t src_pitch, uint8_t *h_dst,
                           size_t dst_pitch);

/* MSScalerDesc.context_process with the reference's own argument shape (msvideo.h:476): three plane
 * pointers + strides on the HOST, one frame, synchronous.  dst[1], dst[2] are ignored for MI_PIX_RGB24.
 * Strides may exceed the width; planes need not be contiguous. */
int mi_scaler_process_planes_host(mi_scaler *s, const uint8_t *const src[3], const int src_strides[3],
                                  uint8_t *const dst[3], const int dst_strides[3]);

/* The scaler fed from HOST buffers with the copies overlapped -- BASELINE config 5 (2048 x 1080p30 over 8 GPUs = 7 680 frames
 * per second and GPU: 23.9 GB/s up, 21.2 GB/s down; the kernel itself does ~800 k frames/s, so the copies ARE the path).
 * Batches of up to batch_frames frames travel through a ring of `depth` (1..8) pinned buffer pairs: upload, kernel and
 * download run on three HIP streams ordered by events, so batch k+1 goes up while batch k is scaled and batch k-1 comes
 * down (what mi_session does for the audio chain).  MSSizeConv's frames of one tick (sizeconv.c:133-181) are one or a few
 * such batches; a capture / decode stage writes straight into the staging buffer (no intermediate copy).
 *   acquire: the pinned staging of the NEXT batch, frame i at h_src + i * src_pitch (layout of ms_yuv_buf_init msvideo.c:85-99);
 *            fails when `depth` batches are in flight (collect first);
 *   submit:  nframes of it are uploaded, scaled, downloaded; returns at once;
 *   collect: the OLDEST batch in flight: waits for its download; frame i at h_dst + i * dst_pitch (valid until `depth` more
 *            batches have been submitted). */
typedef struct mi_scaler_pipe mi_scaler_pipe;
int mi_scaler_pipe_create(mi_scaler *s, int batch_frames, int depth, mi_scaler_pipe **out);
void mi_scaler_pipe_destroy(mi_scaler_pipe *p);
int mi_scaler_pipe_acquire(mi_scaler_pipe *p, uint8_t **h_src, size_t *src_pitch);
int mi_scaler_pipe_submit(mi_scaler_pipe *p, int nframes);
int mi_scaler_pipe_collect(mi_scaler_pipe *p, const uint8_t **h_dst, size_t *dst_pitch, int *nframes);
int mi_scaler_pipe_in_flight(const mi_scaler_pipe *p);

/* ---------------------------------------------------------------- fifo */
/* MSBufferizer (include/mediastreamer2/msqueue.h:131-134, src/base/msqueue.c:70-113) for a batch of streams, resident
 * on the device: lets filters with different block sizes be chained without a host round trip (480-sample ticks
 * from the resampler -> 256-sample frames for the echo canceller, speexec.c:252-257 -> ticks for the mixer). */
int mi_fifo_create(mi_ctx *ctx, int nstreams, int capacity_samples, mi_fifo **out);
void mi_fifo_destroy(mi_fifo *f);
/* ms_bufferizer_put for every stream: row s of d_in ([nstreams][stride]); d_count[s] samples (NULL = nsamples each;
 * 0 = nothing for that stream).  A block that does not fit is refused and counted (mi_fifo_overflows). */
int mi_fifo_push(mi_fifo *f, const int16_t *d_in, int nsamples, int stride, const int32_t *d_count);
/* the same with a byte mask: only streams with d_gate[s] != 0 push (NULL = all) -- takes the `ok` mask of a pop or the
 * `run` mask of mi_aec_process as is */
int mi_fifo_push_gated(mi_fifo *f, const int16_t *d_in, int nsamples, int stride, const uint8_t *d_gate);
/* ms_bufferizer_read, all-or-nothing (msqueue.c:83): streams holding >= frame samples (and with d_gate[s] != 0 when a
 * gate is given) get them in row s of d_out and d_ok[s] = 1; the others keep their samples, d_ok[s] = 0 and, if
 * zero_fill, a row of zeros (the silence the filters inject: speexec.c:261-272, audiomixer.c:88). */
int mi_fifo_pop(mi_fifo *f, int frame, int16_t *d_out, int stride, uint8_t *d_ok, const uint8_t *d_gate, int zero_fill);
/* A whole tick's frames in one launch (the `while` of speex_ec_process, speexec.c:256): up to max_frames frames of
 * `frame` samples, back to back in row s of d_out (stride >= frame * max_frames).
 *  - d_nframes_wanted == NULL: every stream delivers the whole frames it holds (<= max_frames); d_nframes_out[s] says
 *    how many -- the per-stream count mi_aec_process_frames takes;
 *  - d_nframes_wanted != NULL: stream s is asked for that many frames; frames it cannot supply are zero-filled when
 *    zero_fill is set (speexec.c:261-272), d_nframes_out (nullable) reports the frames really popped. */
int mi_fifo_pop_frames(mi_fifo *f, int frame, int max_frames, int16_t *d_out, int stride, uint8_t *d_nframes_out,
                       const uint8_t *d_nframes_wanted, int zero_fill);
/* d_nframes[s] * frame samples of row s are appended (0 = nothing for that stream) */
int mi_fifo_push_frames(mi_fifo *f, const int16_t *d_in, int frame, int max_frames, int stride, const uint8_t *d_nframes);
/* unit * mi_fifo_phase_of(s, phases) samples of silence appended to streams [first, first + count) (see mi_aec_stagger_fifos) */
int mi_fifo_push_lead(mi_fifo *f, int first, int count, int unit, int phases);
int mi_fifo_phase_of(int stream, int phases); /* 0 .. phases-1 */
/* d_count[s] samples of silence appended to stream s (0 or less = nothing): the frame of zeros MSSpeexEC puts into its delay
 * line when the far end runs short (speexec.c:261-272) and the delay line's initial fill (:205-208), decided per leg on the host */
int mi_fifo_push_silence(mi_fifo *f, const int32_t *d_count);
int mi_fifo_levels(mi_fifo *f, int32_t *d_levels); /* ms_bufferizer_get_avail, in samples */
int mi_fifo_overflows(mi_fifo *f, int32_t *h_count);
/* debug / parity read-back (like mi_aec_get): the rings as they lie, [nstreams][capacity], and per stream the read position
 * and the level -- the samples a fused launch queued can be checked without popping them.  Any pointer may be NULL.  Syncs. */
int mi_fifo_snapshot(mi_fifo *f, int16_t *h_rings, int32_t *h_head, int32_t *h_level);
/* The queues of streams [first, first + count) as a host sees MSBufferizers (stream k's h_level[k] samples, oldest first, at h_samples +
 * k * stride; stride >= the capacity) -- and back, every stream of the range replaced.  tail_at_end: the queue ends on the ring's end, as
 * mi_fifo_reset_range_at + pushes leave it (levels then are multiples of 8).  One round trip for a whole conference that is re-plumbed
 * (src/voip/audioconference.c:322-374: what its members' bufferizers hold outlives the detach).  Both wait for the stream. */
int mi_fifo_export_range(mi_fifo *f, int first, int count, int16_t *h_samples, int stride, int32_t *h_level);
int mi_fifo_import_range(mi_fifo *f, int first, int count, const int16_t *h_samples, int stride, const int32_t *h_level, int tail_at_end);
int mi_fifo_reset(mi_fifo *f);
int mi_fifo_reset_range(mi_fifo *f, int first, int count); /* empty the FIFOs of streams [first, first+count) */
/* ... empty, with the read position at ring offset `head` (a multiple of 8, < capacity).  The canceller's launches append whole
 * frames to their output FIFO at a tail they take to be frame-aligned (mi_aec_process_fifos*): a queue that is to START with r
 * samples short of a frame -- what an MSVolume's bufferizer held when its graph was detached (msvolume.c keeps it; the plugin's
 * fused leg hands it over) -- is emptied at head = capacity - r and given those r samples, which leaves its tail at offset 0. */
int mi_fifo_reset_range_at(mi_fifo *f, int first, int count, int head);

/* ------------------------------------- codecs, channel adapter, flow control */
/* The per-stream stages either side of the hot path in an AudioStream graph (src/voip/audiostream.c:1798-1832;
 * SURVEY.md 8(f) rank 3).  Stateless conversions over `rows` rows of `len` samples; strides are in ELEMENTS of the
 * respective type; d_len (nullable) holds a per-row sample count <= len, samples beyond it are left untouched.
 * 16-byte aligned bases with codes_stride % 16 == 0 and pcm_stride % 8 == 0 take the vector path. */
#define MI_LAW_PCMA 0 /* MSAlawDec / MSAlawEnc: src/audiofilters/alaw.c:208-221, :56-90; Snack_* g711.c:113-166 */
#define MI_LAW_PCMU 1 /* MSUlawDec / MSUlawEnc: src/audiofilters/ulaw.c; Snack_* g711.c:200-255 */
int mi_g711_decode(mi_ctx *ctx, int law, const uint8_t *d_codes, size_t codes_stride, int16_t *d_pcm, size_t pcm_stride,
                   const int32_t *d_len, int len, size_t rows);
int mi_g711_encode(mi_ctx *ctx, int law, const int16_t *d_pcm, size_t pcm_stride, uint8_t *d_codes, size_t codes_stride,
                   const int32_t *d_len, int len, size_t rows);
/* MSL16Enc / MSL16Dec sample loop (src/audiofilters/l16.c:58-70,:86,:196): byte order of every sample; in place allowed */
int mi_l16_swap(mi_ctx *ctx, const int16_t *d_in, int16_t *d_out, size_t nsamples);
/* MSChannelAdapter (src/audiofilters/chanadapt.c): frames = samples per channel */
#define MI_CHAN_MONO_TO_STEREO 0     /* :110-113  d_out[2 * frames] */
#define MI_CHAN_STEREO_TO_MONO 1     /* :118-121  keeps the left sample; d_a[2 * frames] -> d_out[frames] */
#define MI_CHAN_TWO_MONO_TO_STEREO 2 /* :81-90    d_b nullable = that side is silent */
int mi_chan_adapt(mi_ctx *ctx, int mode, const int16_t *d_a, const int16_t *d_b, int16_t *d_out, size_t frames);

/* MSAudioFlowControl for a batch of streams: ms_audio_flow_controller_process (src/audiofilters/flowcontrol.c:107-152)
 * with discard_well_choosed_samples (:56-89) and compute_frame_power (:97-105); state = MSAudioFlowController
 * (include/mediastreamer2/flowcontrol.h:40-46) per stream on the device. */
typedef struct mi_flowctl mi_flowctl;
#define MI_FLOWCTL_BASIC 0 /* MSAudioFlowControlBasic */
#define MI_FLOWCTL_SOFT 1  /* MSAudioFlowControlSoft (default, silent_threshold 0.02: flowcontrol.c:37-41) */
int mi_flowctl_create(mi_ctx *ctx, int nstreams, int max_block /* samples, 3..2048 */, mi_flowctl **out);
void mi_flowctl_destroy(mi_flowctl *f);
int mi_flowctl_set_config(mi_flowctl *f, int first, int count, int strategy, float silent_threshold); /* MS_AUDIO_FLOW_CONTROL_SET_CONFIG */
/* MS_AUDIO_FLOW_CONTROL_DROP (:209-219) for every stream with a non-zero request; ignored, like there, by streams
 * that are still dropping.  Host arrays [nstreams], in samples (drop_ms * rate * nchannels / 1000). */
int mi_flowctl_request_drop(mi_flowctl *f, const uint32_t *h_samples_to_drop, const uint32_t *h_total_samples);
/* one block per stream (d_len[s] == 0: no block this round); d_out_len[s] = samples left, 0 = block dropped.
 * d_out may be d_in. */
int mi_flowctl_process(mi_flowctl *f, const int16_t *d_in, size_t in_stride, const int32_t *d_len, int len, int16_t *d_out,
                       size_t out_stride, int32_t *d_out_len);
int mi_flowctl_get_state(mi_flowctl *f, int stream, uint32_t out4[4]); /* target, total, pos, dropped */
int mi_flowctl_reset(mi_flowctl *f, int first, int count);             /* ms_audio_flow_controller_reset :30-35 */

/* MSGenericPLC for a batch of streams: generic_plc_process (src/audiofilters/msgenericplc.c:59-167) over plc_context_t
 * (src/audiofilters/genericplc.c:29-241).  Mono, 16-bit; any rate whose nb = rate/20 (rounded down to 100) has no prime
 * factor above 17, kiss_fft's own limit (8 .. 48 kHz incl. 22.05 / 44.1 kHz: radix 11 through the generic butterfly). */
typedef struct mi_plc mi_plc;
#define MI_PLC_NONE 0       /* no event for the stream this round */
#define MI_PLC_RECEIVED 1   /* a block of d_len[s] samples arrived: edited in place (delayed 5 ms, cross-faded after a loss) :63-116 */
#define MI_PLC_CONCEAL 2    /* nothing arrived: d_len[s] samples are generated into the row :150-156, genericplc.c:123-197 */
#define MI_PLC_CNG_RESUME 4 /* with RECEIVED: the stream was in comfort noise (silence without bcg729) :76-89 */
int mi_plc_create(mi_ctx *ctx, int nstreams, int rate, int max_block, mi_plc **out);
void mi_plc_destroy(mi_plc *p);
int mi_plc_reset(mi_plc *p, int first, int count);
/* d_blocks [nstreams][stride] int16 in/out, d_len [nstreams] int32, d_mode [nstreams] uint8 (MI_PLC_*) */
int mi_plc_process(mi_plc *p, int16_t *d_blocks, size_t stride, const int32_t *d_len, const uint8_t *d_mode);
int mi_plc_info(mi_plc *p, int stream, int32_t out3[3]); /* plc_buffer_samples_nb, plc_index, plc_samples_used */

/* ------------------------------------------------------------- session */
/* The chained path of BASELINE.json's north_star for a batch of call legs, fed from host buffers, one 10 ms tick per
 * submit: MSResample in_rate->rate (msresample.c:122-179) -> FIFO -> MSSpeexEC + post-filter (speexec.c:223-305) ->
 * FIFO -> MSVolume with AGC (msvolume.c:471-514) -> MSAudioMixer in conference mode (audiomixer.c:288-346; stream
 * s = conference * members + member).  Uploads, kernels and downloads run on three HIP streams, up to three ticks
 * in flight; the kernel sequence is a hipGraph per buffer slot.  Outputs equal the same C ABI objects called one by
 * one (tests/test_gpu_pipeline.py). */
typedef struct mi_session mi_session;
typedef struct mi_session_config {
	int32_t nstreams;               /* multiple of members_per_conference */
	int32_t members_per_conference; /* <= MI_MIXER_MAX_CHANNELS */
	int32_t in_rate;                /* microphone rate; == rate skips the resampler */
	int32_t rate;                   /* processing / output rate */
	int32_t tail_ms;                /* MS_ECHO_CANCELLER_SET_TAIL_LENGTH */
	int32_t agc;                    /* MS_VOLUME_ENABLE_AGC */
	int32_t use_graphs;             /* 1: replay a hipGraph per tick, 0: launch the kernels one by one */
	/* the legs as a SIP trunk delivers them (all 0 = 16-bit PCM in, 16-bit PCM at `rate` out, reference from the host) */
	int32_t mic_codec;    /* MI_SESSION_PCM16 | MI_SESSION_PCMA | MI_SESSION_PCMU: G.711 code words at in_rate (MSAlawDec / MSUlawDec) */
	int32_t out_rate;     /* 0 = rate; else each leg's mix is resampled rate -> out_rate (MSResample) before it leaves */
	int32_t out_codec;    /* as mic_codec: the output leaves as G.711 code words (MSAlawEnc / MSUlawEnc) */
	int32_t ref_loopback; /* 1: a leg's far-end reference is the mix this session sent it on the previous tick: no upload */
	int32_t ref_delay_ms; /* MS_ECHO_CANCELLER_SET_DELAY: the reference FIFO starts with this much silence (speexec.c:205-208) */
	int32_t plc;          /* 1: MSGenericPLC behind the decoder (msgenericplc.c): legs flagged lost for a tick are concealed */
	int32_t stagger;      /* 1 (default): every leg starts -- at creation, on reset / add_member -- with the re-framing lead of
	                       * mi_aec_stagger_fifos (0 .. 7/8 of a frame of silence in its microphone and reference queues), so that
	                       * legs which start together do not all have their light tick together; 0: legs start empty */
} mi_session_config;
#define MI_SESSION_PCM16 0
#define MI_SESSION_PCMA 1
#define MI_SESSION_PCMU 2
void mi_session_default_config(mi_session_config *c);
int mi_session_create(mi_ctx *ctx, const mi_session_config *cfg, mi_session **out);
void mi_session_destroy(mi_session *s);
int mi_session_tick_samples(const mi_session *s, int *in_samples, int *out_samples);
/* bytes per stream and tick of the three host buffers (mic, reference, output); the reference is 0 with ref_loopback */
int mi_session_tick_bytes(const mi_session *s, int *mic_bytes, int *ref_bytes, int *out_bytes);
/* pinned staging of the NEXT tick, to be filled in place: mic [nstreams][in_rate/100] int16 (or uint8 code words with
 * mic_codec), far-end reference [nstreams][rate/100] int16 (*h_ref = NULL with ref_loopback) */
int mi_session_acquire(mi_session *s, int16_t **h_mic, int16_t **h_ref);
/* with cfg.plc: the per-leg event bytes of the tick being filled, [nstreams], preset to MI_PLC_RECEIVED; set a leg to
 * MI_PLC_CONCEAL when its packet did not arrive (its mic row is then ignored).  Valid between acquire and submit. */
int mi_session_events(mi_session *s, uint8_t **h_events);
int mi_session_submit(mi_session *s);
/* the OLDEST tick in flight: waits for its download, returns the pinned output [nstreams][(out_rate or rate)/100]
 * int16, or uint8 code words with out_codec (valid until three more ticks have been submitted) */
int mi_session_collect(mi_session *s, const int16_t **h_out);
int mi_session_in_flight(const mi_session *s);
/* conference control plane: per-stream mixer flags (MI_MIX_LINKED | MI_MIX_ACTIVE | MI_MIX_OUTPUT: mute = clear ACTIVE,
 * MS_AUDIO_MIXER_SET_ACTIVE audiomixer.c:384-393; listen-only / no return = MS_AUDIO_MIXER_ENABLE_OUTPUT :404-414) and
 * input gains (MS_AUDIO_MIXER_SET_INPUT_GAIN :372-382), either may be NULL; arrays of [nstreams].  Takes effect for the
 * ticks submitted afterwards (waits for the ones in flight). */
int mi_session_set_controls(mi_session *s, const uint8_t *h_flags, const float *h_gain);
/* a call leg was replaced: streams [first, first+count) start over (resampler history, canceller, meter, FIFOs) as
 * newly created filters would; the other streams are untouched */
int mi_session_reset_streams(mi_session *s, int first, int count);
/* MS_VOLUME_GET_LINEAR of every stream (msvolume.c:129-134): what an active-speaker detector polls */
int mi_session_get_levels(mi_session *s, float *h_linear);
/* MSAudioConference membership (src/voip/audioconference.c:322-374).  A session is created full (every stream a member);
 * remove_member unplumbs a stream's mixer pin (it neither contributes nor hears; its output row is no longer written),
 * add_member plumbs it again for a NEW endpoint: that stream's resampler history, canceller, meter and FIFOs start
 * over, the other members keep theirs (the reference re-attaches the conference graph around both calls; the filters of
 * the remaining members survive that).  Both wait for the ticks in flight. */
int mi_session_add_member(mi_session *s, int stream);
int mi_session_remove_member(mi_session *s, int stream);
int mi_session_member_count(const mi_session *s, int conference); /* plumbed pins, or MI_EINVAL */
/* The active-speaker election of a conference in mixer mode (audioconference.c:436-452): per conference the plumbed,
 * unmuted member with the largest MS_VOLUME_GET_MAX (maximum of the smoothed energy over a one-second window, dBm0)
 * above -30 dB; h_winner[conf] = its stream index or -1, h_max_db[conf] (nullable) its level.  The windows are kept on
 * the device and fed by every tick (mi_volume_get_max), so the result does not depend on how often the application
 * polls; now_ms is not needed for that any more and is ignored (the ticks are the clock, 10 ms each). */
int mi_session_active_speakers(mi_session *s, uint64_t now_ms, int32_t *h_winner, float *h_max_db);

/* ------------------------------------------------------------- pixconv */
/* Packed formats -> I420, what pixconv_process (src/videofilters/pixconv.c:62-94) obtains from
 * ms_scaler_process with the libyuv implementation (yuv_scale src/voip/msvideo.c:542-581).
 * Formats are named by MEMORY byte order. */
typedef struct mi_pixconv mi_pixconv;
#define MI_PIX_YUY2 2      /* MS_YUY2, MS_YUYV: Y0 U Y1 V      (YUY2ToI420,  msvideo.c:553) */
#define MI_PIX_UYVY 3      /* MS_UYVY:          U Y0 V Y1      (UYVYToI420,  :558) */
#define MI_PIX_BGR24 4     /* MS_RGB24:         B G R, full-range output (RGB24ToJ420, :562) */
#define MI_PIX_RGB24_RAW 5 /* MS_RGB24_REV:     R G B          (RAWToI420,   :566) */
#define MI_PIX_BGRA32 6    /* MS_RGBA32_REV:    B G R A        (ARGBToI420,  :570) */
/* flip_vertical: read the source bottom-up (pixconv.c:78-81 does it for MS_RGB24_REV with a negative
 * stride).  Width must be even.  Source rows are packed (stride = w * bytes per pixel, as
 * ms_picture_init_from_mblk_with_size msvideo.c:121-160 sets them). */
int mi_pixconv_create(mi_ctx *ctx, int w, int h, int src_fmt, int flip_vertical, mi_pixconv **out);
void mi_pixconv_destroy(mi_pixconv *p);
size_t mi_pixconv_src_bytes(const mi_pixconv *p);
size_t mi_pixconv_dst_bytes(const mi_pixconv *p);
int mi_pixconv_process(mi_pixconv *p, int nframes, const uint8_t *d_src, size_t src_pitch, uint8_t *d_dst,
                       size_t dst_pitch);
int mi_pixconv_process_host(mi_pixconv *p, int nframes, const uint8_t *h_src, size_t src_pitch, uint8_t *h_dst,
                            size_t dst_pitch);

#ifdef __cplusplus
}
#endif
#endif /* MSMI355X_H */
