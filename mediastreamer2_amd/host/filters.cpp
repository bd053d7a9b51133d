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

#include "filters/resample.inl"
#include "filters/volume.inl"
#include "filters/equalizer.inl"
#include "filters/mixer.inl"
#include "filters/echo_canceller.inl"
#include "filters/video.inl"
#include "filters/codec.inl"
#include "filters/flow_control.inl"
#include "filters/generic_plc.inl"

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
