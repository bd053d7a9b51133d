/*
 * ms2_oracle.h -- CPU oracle for the mediastreamer2 DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mediastreamer2_amd/,
 * include/) may include, link or call this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY UNPINNED: the reference (mediastreamer2 5.5.0 under /root/reference)
 * cannot be built in this image -- every source file on the path includes
 * bctoolbox / oRTP headers (e.g. src/utils/kiss_fft.c:26, include/mediastreamer2/
 * mscommon.h) that are absent, the resampler / AEC / scaler arithmetic lives in
 * un-vendored libspeexdsp / libyuv / swscale, and the reference's own tests
 * hold no numeric golden vectors for these filters.  Each
 * function below is a plain-C restatement of the cited reference lines (or of
 * the published third-party algorithm where noted), checked by known-answer
 * tests and analytic properties in tests/, not against reference outputs.
 *
 * All citations are relative to /root/reference.
 */
#ifndef MS2_ORACLE_H
#define MS2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ mixer */
/* Restates src/audiofilters/audiomixer.c:33-51 (accumulate/saturate/apply_gain),
 * :78-90 (channel_process_in), :113-130 (channel_process_out), :210-217
 * (make_output), :301-344 (mixer_process core) for ONE conference, ONE tick.
 *
 * in        [nmembers][nsamples] int16, the tick each channel's bufferizer
 *           delivered; has_data[m]==0 means the read came up short and the
 *           channel contributes zeros (audiomixer.c:88).
 * gain/active/out_enabled  per channel controls (audiomixer.c:57-63).
 * conf_mode 1: out[m] = sat(sum - own gained input) for active channels,
 *              sat(sum) for inactive ones; rows of disabled outputs are left
 *              untouched.  0: out row 0 = sat(sum) (single shared block).
 * sum_out   optional [nsamples] int32 copy of the tick's sum. */
void orc_mixer_tick(const int16_t *in, const uint8_t *has_data, const float *gain,
                    const uint8_t *active, const uint8_t *out_enabled, int nmembers,
                    int nsamples, int conf_mode, int16_t *out, int32_t *sum_out);

/* ----------------------------------------------------------------- volume */
/* Mirrors struct Volume of src/audiofilters/msvolume.c:48-86 (DSP fields). */
typedef struct OrcVolume {
	float energy, level_pk, instant_energy, lt_speaker_en;
	float gain, static_gain;
	int dc_offset;
	float vol_upramp, vol_fast_upramp, vol_downramp;
	float ea_thres, ea_transmit_thres, force, target_gain;
	int sustain_time, sustain_dur;
	int sample_rate, nsamples;
	int ng_cut_time, ng_noise_dur;
	float ng_threshold, ng_floorgain, ng_gain;
	int agc_enabled, noise_gate_enabled, remove_dc, fast_upramp;
	int has_peer;
} OrcVolume;

void orc_volume_init(OrcVolume *v);                 /* msvolume.c:88-118 */
void orc_volume_set_rate(OrcVolume *v, int rate);   /* :150-154 + :450 */
void orc_volume_set_gain(OrcVolume *v, float g);    /* :270-276 */
void orc_volume_set_db_gain(OrcVolume *v, float d); /* :262-268 */
void orc_volume_enable_noise_gate(OrcVolume *v, int on); /* :352-359 */
/* One chunk through update_energy / echo limiter / AGC / noise gate /
 * apply_gain exactly as the loop bodies of volume_process (msvolume.c:480-513).
 * peer_energy is the peer MSVolume's `energy` (msvolume.c:206-207), ignored
 * unless v->has_peer.  samples are modified in place. */
void orc_volume_chunk(OrcVolume *v, int16_t *samples, int n, float peer_energy);

/* -------------------------------------------------------------- resampler */
/* Restates the libspeexdsp resampler (third-party, un-vendored; called from
 * src/audiofilters/msresample.c:114,157; version unpinned: CMakeLists.txt:207)
 * float build, speex_resampler_init + speex_resampler_process_int, following
 * the published algorithm (resample.c of speexdsp 1.2.x).  Mono. */
typedef struct OrcResampler OrcResampler;
OrcResampler *orc_resampler_new(uint32_t in_rate, uint32_t out_rate, int quality);
void orc_resampler_free(OrcResampler *r);
/* like speex_resampler_process_int: *in_len / *out_len are updated to the
 * consumed / produced counts. */
void orc_resampler_process(OrcResampler *r, const int16_t *in, uint32_t *in_len, int16_t *out,
                           uint32_t *out_len);
int orc_resampler_filt_len(const OrcResampler *r);
int orc_resampler_den_rate(const OrcResampler *r);
int orc_resampler_num_rate(const OrcResampler *r);
int orc_resampler_is_direct(const OrcResampler *r);
/* copy of the direct sinc table [den_rate][filt_len] (float) or, in
 * interpolated mode, the oversampled table [filt_len*oversample+8] */
int orc_resampler_table(const OrcResampler *r, float *dst, int cap);
/* msresample.c:150-177 framing for one mblk: outlen capacity = inlen*out/in+1 */
uint32_t orc_msresample_outcap(uint32_t inlen, uint32_t in_rate, uint32_t out_rate);

/* -------------------------------------------------------------- equalizer */
/* kiss_fft real transforms, float build: src/utils/kiss_fft.c:38-149 (bfly2/4),
 * :320-408 (kf_work), :412-475 (factor/alloc), src/utils/kiss_fftr.c:40-81,
 * :175-259 (kiss_fftr2), :261-296 (kiss_fftri2); wrappers ms_fft/ms_ifft
 * src/utils/dsptools.c:333-376. nfft must be even; radix 4, 2, 3, 5 and generic (<= 17) stages (kiss_fft.c:150-290). */
typedef struct OrcFft OrcFft; /* ms_fft_init handle */
OrcFft *orc_fft_new(int nfft);
void orc_fft_free(OrcFft *t);
void orc_fft_forward(OrcFft *t, const float *in, float *out); /* scaled 1/N */
void orc_fft_inverse(OrcFft *t, const float *in, float *out); /* unscaled */
void orc_ms_fft(int nfft, const float *in, float *out);  /* forward, scaled 1/N */
void orc_ms_ifft(int nfft, const float *in, float *out); /* inverse, unscaled */

typedef struct OrcEqualizer {
	int rate, nfft, fir_len;
	float *fft_cpx, *fir, *mem;
	int needs_update, active;
} OrcEqualizer;
OrcEqualizer *orc_equalizer_new(int rate);                /* equalizer.c:81-86 */
void orc_equalizer_free(OrcEqualizer *s);
void orc_equalizer_set_rate(OrcEqualizer *s, int rate);   /* :57-79 */
void orc_equalizer_set_gain(OrcEqualizer *s, int freq_0, float gain, int freq_bw); /* :147-172 */
void orc_equalizer_design(OrcEqualizer *s);               /* :215-237 */
/* equalizer_state_run (:263-269) + ms_fir_mem16 (dsptools.c:253-268), in place.
 * float->int16 out-of-range is UB in the reference (equalizer.c:251-255); the
 * oracle saturates to [-32768,32767] (documented divergence). */
void orc_equalizer_run(OrcEqualizer *s, int16_t *samples, int nsamples);
void orc_fir_mem16(const float *x, const float *num, float *y, int N, int ord, float *mem);

/* ------------------------------------------------------------------ video */
/* I420 bilinear down-scale, per plane, following libyuv's C reference path
 * (third-party, un-vendored; called from src/voip/msvideo.c:548; unpinned):
 * ScaleSlope / ScalePlaneBilinearDown / InterpolateRow_C / ScaleFilterCols_C. */
void orc_scale_plane_bilinear(const uint8_t *src, int src_stride, int sw, int sh, uint8_t *dst,
                              int dst_stride, int dw, int dh);
/* frame layout of ms_yuv_buf_init (src/voip/msvideo.c:85-99): Y w*h, U,V
 * (w/2)*(h/2) contiguous, odd h rounded up. */
void orc_i420_scale(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
/* BT.601 limited range, the in-tree Q13 constants of src/voip/scaler_arm.S:54-63
 * (9535,13074,6660,3203,16531) == src/yuv2rgb.fs coefficients; RGB24 byte
 * order R,G,B; chroma replicated over each 2x2 block. */
void orc_i420_to_rgb24(const uint8_t *src, int w, int h, uint8_t *rgb, int rgb_stride);
/* the fused pipeline the GPU kernel implements: scale then convert */
void orc_i420_scale_to_rgb24(const uint8_t *src, int sw, int sh, uint8_t *rgb, int dw, int dh);

/* Packed formats -> I420, what MSPixConv (src/videofilters/pixconv.c:62-94) gets from the libyuv
 * scaler implementation (yuv_scale, src/voip/msvideo.c:542-581).  Formats named by MEMORY order. */
#define ORC_PIX_YUY2 2      /* MS_YUY2 / MS_YUYV  -> YUY2ToI420  */
#define ORC_PIX_UYVY 3      /* MS_UYVY            -> UYVYToI420  */
#define ORC_PIX_BGR24 4     /* MS_RGB24           -> RGB24ToJ420 (full range) */
#define ORC_PIX_RGB24_RAW 5 /* MS_RGB24_REV       -> RAWToI420   */
#define ORC_PIX_BGRA32 6    /* MS_RGBA32_REV      -> ARGBToI420  */
/* src_stride may be negative (pixconv.c:78-81 flips RGB24_REV); w must be even.  0 on success. */
int orc_pixconv_to_i420(int fmt, const uint8_t *src, int src_stride, int w, int h, uint8_t *dst);

/* -------------------------------------------------------------------- AEC */
/* Restates libspeexdsp's MDF echo canceller (mdf.c, float build) and the
 * preprocessor residual-echo / denoise stage (preprocess.c) as called from
 * src/audiofilters/speexec.c:200-203,297-298.  Third-party, un-vendored. */
typedef struct OrcEcho OrcEcho;
OrcEcho *orc_echo_new(int frame_size, int filter_length, int sample_rate);
void orc_echo_free(OrcEcho *st);
void orc_echo_cancel(OrcEcho *st, const int16_t *rec, const int16_t *play, int16_t *out);
/* state introspection for parity tests: copies W (background) [M*N] floats */
int orc_echo_get(const OrcEcho *st, const char *what, float *dst, int cap);
int adjust_framesize_8000(int framesize_at_8000, int samplerate); /* speexec.c:171-180 */

typedef struct OrcPreproc OrcPreproc;
OrcPreproc *orc_preproc_new(int frame_size, int sample_rate, OrcEcho *echo);
void orc_preproc_free(OrcPreproc *st);
void orc_preproc_run(OrcPreproc *st, int16_t *x);

/* ------------------------------------------- codecs / channel adapter / flow control */
/* oracle/g711.c.  The four G.711 conversions ARE pinned: oracle/_ref/libg711_ref.so is the
 * reference's own src/audiofilters/g711.c compiled unmodified (oracle/build_ref.sh). */
uint8_t orc_lin2alaw(int16_t pcm);  /* Snack_Lin2Alaw  g711.c:113-141 */
int16_t orc_alaw2lin(uint8_t code); /* Snack_Alaw2Lin  g711.c:147-166 */
uint8_t orc_lin2ulaw(int16_t pcm);  /* Snack_Lin2Mulaw g711.c:200-231 */
int16_t orc_ulaw2lin(uint8_t code); /* Snack_Mulaw2Lin g711.c:242-255 */
void orc_g711_encode(int law, const int16_t *pcm, size_t n, uint8_t *codes); /* law 0 = PCMA, 1 = PCMU */
void orc_g711_decode(int law, const uint8_t *codes, size_t n, int16_t *pcm);
void orc_l16_swap(const int16_t *in, size_t n, int16_t *out); /* l16.c:58-70 */
void orc_chan_adapt(int mode, const int16_t *a, const int16_t *b, size_t n, int16_t *out); /* chanadapt.c:87-121 */

/* MSAudioFlowController, include/mediastreamer2/flowcontrol.h:30-46 */
typedef struct OrcFlowCtl {
	int strategy; /* 0 basic, 1 soft */
	float silent_threshold;
	uint32_t target_samples, total_samples, current_pos, current_dropped;
} OrcFlowCtl;
void orc_flowctl_init(OrcFlowCtl *c);
void orc_flowctl_set_target(OrcFlowCtl *c, uint32_t samples_to_drop, uint32_t total_samples);
size_t orc_flowctl_process(OrcFlowCtl *c, int16_t *samples, size_t n); /* flowcontrol.c:107-152 */

/* ------------------------------------------------------------ MSGenericPLC */
/* oracle/plc.c: src/audiofilters/genericplc.c + msgenericplc.c:59-167 + MSConcealerContext (mscommon.c:315-366) */
typedef struct OrcPlc OrcPlc; /* uses the OrcFft handles above: any even size whose factors are <= 17 */
OrcPlc *orc_plc_new(int rate);
void orc_plc_free(OrcPlc *c);
int orc_plc_info(const OrcPlc *c, int *nb, int *index, int *used);
void orc_plc_transition_mix(int16_t *inout, const int16_t *continuity, uint16_t n);
void orc_plc_update_history(OrcPlc *c, const int16_t *data, size_t n);
void orc_plc_update_continuity(OrcPlc *c, int16_t *data, size_t n);
void orc_plc_generate(OrcPlc *c, int16_t *data, uint16_t n);
void orc_plc_received(OrcPlc *c, int16_t *data, size_t n, int cng_resume); /* a block arrived (edited in place) */
void orc_plc_conceal(OrcPlc *c, int16_t *data, uint16_t n);                /* a tick without one */
typedef struct OrcConcealer {
	int64_t sample_time, plc_start_time;
	unsigned long total_number_for_plc;
	uint32_t max_plc_time;
} OrcConcealer;
void orc_concealer_init(OrcConcealer *o, uint32_t max_plc_time);
uint32_t orc_concealer_inc_sample_time(OrcConcealer *o, uint64_t now, uint32_t increment, int got_packet);
int orc_concealer_required(OrcConcealer *o, uint64_t now);

/* ------------------------------------------------- conference glue (oracle/conference.c)
 * OrtpExtremum as MSVolume uses it (msvolume.c:115-116,405-406; oRTP itself is not under /root/reference: parity unpinned)
 * and MSAudioConference's bookkeeping in mixer mode (src/voip/audioconference.c). */
#define ORC_MIXER_MAX_CHANNELS 50   /* audiomixer.c:29 */
#define ORC_VOLUME_DB_LOWEST (-120) /* msvolume.h */
#define ORC_VOLUMES_NOT_FOUND (-32768) /* AUDIOSTREAMVOLUMES_NOT_FOUND = INT16_MIN, mediastream.h:1131 */
typedef struct OrcExtremum {
	float current_extremum, last_stable;
	uint64_t extremum_time;
	int period;
} OrcExtremum;
void orc_extremum_init(OrcExtremum *e, int period);
void orc_extremum_reset(OrcExtremum *e);
int orc_extremum_record_min(OrcExtremum *e, uint64_t curtime, float value);
int orc_extremum_record_max(OrcExtremum *e, uint64_t curtime, float value);
float orc_extremum_get_current(const OrcExtremum *e);
float orc_volume_linear_to_dbm0(float linear); /* msvolume.c:565-568 */
typedef struct OrcConference {
	int nmembers, active_speaker; /* active speaker: its mixer pin, -1 = none elected yet */
	uint8_t plumbed[ORC_MIXER_MAX_CHANNELS], muted[ORC_MIXER_MAX_CHANNELS];
} OrcConference;
void orc_conference_init(OrcConference *c);                             /* audioconference.c:67-92 */
int orc_conference_add_member(OrcConference *c, int muted);              /* :198-207,322-345 -> the member's pin */
void orc_conference_remove_member(OrcConference *c, int pin);            /* :366-374 */
void orc_conference_mute_member(OrcConference *c, int pin, int muted);   /* :376-388 */
int orc_conference_get_size(const OrcConference *c);                     /* :390-392 */
int orc_conference_participant_volume(const OrcConference *c, int pin, float volume_db); /* :394-418 */
int orc_conference_process_events(OrcConference *c, const int *order, const float *max_db, int *winner_pin, float *winner_db); /* :419-464 */

/* ------------------------------------------------- recording metrics (oracle/audiodiff.c)
 * src/utils/audiodiff.c on WAV files (read with include/ms2_mediaio.h: sizes from the file length).
 * MSAudioDiffParams {max_shift_percent, chunk_size_ms} are passed as two ints. */
int orc_audio_diff(const char *ref_file, const char *matched_file, double *ret, int max_shift_percent, int chunk_size_ms); /* :578-651 */
int orc_audio_compare_silence_and_speech(const char *ref_file, const char *matched_file, double *ret, double *energy,
                                         int max_shift_percent, int chunk_size_ms, int start_time_short_ms,
                                         int stop_time_short_ms, int start_time_ms); /* :442-576 */
int orc_audio_energy(const char *file, double *energy); /* :653-682 */

#ifdef __cplusplus
}
#endif
#endif
