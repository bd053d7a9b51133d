/*
 * oracle/g711.c -- CPU oracle for the per-stream stages either side of the hot path
 * (SURVEY.md section 8(f) rank 3): G.711 A-law / mu-law, L16, channel adaptation and
 * the audio flow controller.
 *
 * TEST INFRASTRUCTURE ONLY (see ms2_oracle.h).
 *
 * PINNED for G.711: src/audiofilters/g711.c is the one reference source on the path that
 * includes nothing but its own header, so oracle/build_ref.sh compiles it unmodified into
 * oracle/_ref/libg711_ref.so and tests/test_oracle_cpu.py compares the four conversions
 * below with it over their whole domain (65 536 PCM values, 256 code words).
 * The L16 / channel-adapter / flow-controller restatements have no such build
 * (their sources need bctoolbox / oRTP headers): parity unpinned, as for the rest.
 *
 * Citations are relative to /root/reference.
 */
#include "ms2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* number of significant bits of v (0 for v == 0) */
static int bit_length(unsigned v) {
	int n = 0;
	while (v) {
		++n;
		v >>= 1;
	}
	return n;
}

/* Snack_Lin2Alaw, src/audiofilters/g711.c:113-141.  13-bit magnitude (pcm >> 3, negatives
 * mapped through -x-1), segment = first table end 0x1F,0x3F,..,0xFFF that holds it (:50,:80-87),
 * i.e. bit_length - 5 floored at 0; four mantissa bits below the leading one (shift 1 in the
 * two linear segments); sign bit set for non-negative input; even bits inverted (0x55). */
uint8_t orc_lin2alaw(int16_t pcm) {
	int v = pcm >> 3;
	const int toggle = v >= 0 ? 0xD5 : 0x55;
	if (v < 0) v = -v - 1;
	int seg = bit_length((unsigned)v) - 5;
	if (seg < 0) seg = 0;
	if (seg >= 8) return (uint8_t)(0x7F ^ toggle); /* unreachable for 16-bit input; kept for the table's meaning */
	const int mant = (v >> (seg < 2 ? 1 : seg)) & 0xF;
	return (uint8_t)(((seg << 4) | mant) ^ toggle);
}

/* Snack_Alaw2Lin, g711.c:147-166 */
int16_t orc_alaw2lin(uint8_t code) {
	const int a = code ^ 0x55;
	const int seg = (a >> 4) & 7;
	int mag = (a & 0xF) << 4;
	if (seg == 0) mag += 8;
	else mag = (mag + 0x108) << (seg - 1);
	return (int16_t)((a & 0x80) ? mag : -mag);
}

/* Snack_Lin2Mulaw, g711.c:200-231: 14-bit magnitude (pcm >> 2, then negated), clipped at 8159,
 * biased by 0x84 >> 2 = 33; segment from the ends 0x3F..0x1FFF (:51) = bit_length - 6; the
 * clipped maximum 8192 falls past the last segment and yields 0x7F ^ mask (:221-222). */
uint8_t orc_lin2ulaw(int16_t pcm) {
	int v = pcm >> 2;
	const int toggle = v < 0 ? 0x7F : 0xFF;
	if (v < 0) v = -v;
	if (v > 8159) v = 8159;
	v += 33;
	int seg = bit_length((unsigned)v) - 6;
	if (seg < 0) seg = 0;
	if (seg >= 8) return (uint8_t)(0x7F ^ toggle);
	return (uint8_t)(((seg << 4) | ((v >> (seg + 1)) & 0xF)) ^ toggle);
}

/* Snack_Mulaw2Lin, g711.c:242-255 */
int16_t orc_ulaw2lin(uint8_t code) {
	const int u = (~code) & 0xFF;
	const int mag = (((u & 0xF) << 3) + 0x84) << ((u >> 4) & 7);
	return (int16_t)((u & 0x80) ? 0x84 - mag : mag - 0x84);
}

/* One block through MSAlawEnc / MSUlawEnc's sample loop (alaw.c:77-82, ulaw.c:78-82) or
 * MSAlawDec / MSUlawDec's (alaw.c:213-217, ulaw.c same).  law: 0 = A-law (PCMA), 1 = mu-law (PCMU). */
void orc_g711_encode(int law, const int16_t *pcm, size_t n, uint8_t *codes) {
	for (size_t i = 0; i < n; ++i) codes[i] = law ? orc_lin2ulaw(pcm[i]) : orc_lin2alaw(pcm[i]);
}
void orc_g711_decode(int law, const uint8_t *codes, size_t n, int16_t *pcm) {
	for (size_t i = 0; i < n; ++i) pcm[i] = law ? orc_ulaw2lin(codes[i]) : orc_alaw2lin(codes[i]);
}

/* MSL16Enc / MSL16Dec: host <-> network byte order of every sample (l16.c:58-70, :86, :196);
 * the same swap both ways on a little-endian host. */
void orc_l16_swap(const int16_t *in, size_t n, int16_t *out) {
	for (size_t i = 0; i < n; ++i) {
		const uint16_t v = (uint16_t)in[i];
		out[i] = (int16_t)(uint16_t)((v << 8) | (v >> 8));
	}
}

/* MSChannelAdapter's three sample loops (chanadapt.c):
 *   mode 0  mono -> stereo, each sample written twice            (:110-113)
 *   mode 1  stereo -> mono, the LEFT sample kept, right dropped  (:118-121)
 *   mode 2  two mono inputs -> one interleaved stereo block      (:87-90); b may be NULL = silence (:81-82)
 * n = number of input frames (samples per channel). */
void orc_chan_adapt(int mode, const int16_t *a, const int16_t *b, size_t n, int16_t *out) {
	for (size_t i = 0; i < n; ++i) {
		if (mode == 0) out[2 * i] = out[2 * i + 1] = a[i];
		else if (mode == 1) out[i] = a[2 * i];
		else {
			out[2 * i] = a ? a[i] : 0;
			out[2 * i + 1] = b ? b[i] : 0;
		}
	}
}

/* ------------------------------------------------------------ audio flow controller */
/* ms_audio_flow_controller_init, flowcontrol.c:37-41 */
void orc_flowctl_init(OrcFlowCtl *c) {
	memset(c, 0, sizeof(*c));
	c->strategy = 1; /* MSAudioFlowControlSoft */
	c->silent_threshold = 0.02f;
}

/* ms_audio_flow_controller_set_target, flowcontrol.c:49-54 */
void orc_flowctl_set_target(OrcFlowCtl *c, uint32_t samples_to_drop, uint32_t total_samples) {
	c->target_samples = samples_to_drop;
	c->total_samples = total_samples;
	c->current_pos = 0;
	c->current_dropped = 0;
}

/* discard_well_choosed_samples, flowcontrol.c:56-89 (three-sample criterion): todrop times, find the
 * LAST position whose two neighbouring differences sum to the minimum (<= comparison, start value
 * 32768) and delete the middle sample of that triple. */
static size_t drop_smooth_samples(int16_t *s, size_t n, uint32_t todrop) {
	while (todrop--) {
		int best = 32768;
		size_t pos = 0;
		for (size_t i = 0; i + 2 < n; ++i) {
			const int d = abs((int)s[i] - (int)s[i + 1]) + abs((int)s[i + 1] - (int)s[i + 2]);
			if (d <= best) {
				best = d;
				pos = i;
			}
		}
		memmove(s + pos + 1, s + pos + 2, (n - pos - 2) * sizeof(int16_t));
		--n;
	}
	return n;
}

/* compute_frame_power, flowcontrol.c:97-105: float accumulation in sample order */
static float frame_power(const int16_t *s, uint32_t n) {
	float acc = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const int v = s[i];
		acc += (float)(v * v);
	}
	return sqrtf(acc / (float)n) / (32768 * 0.7f);
}

/* ms_audio_flow_controller_process, flowcontrol.c:107-152, for one block of n samples edited in
 * place.  Returns the number of samples left (0 = the block is dropped entirely). */
size_t orc_flowctl_process(OrcFlowCtl *c, int16_t *s, size_t n) {
	if (!(c->total_samples > 0 && c->target_samples > 0)) return n;
	const uint32_t nsamples = (uint32_t)n;
	size_t left = n;
	c->current_pos += nsamples;
	if (c->strategy == 0) { /* MSAudioFlowControlBasic :115-121 */
		if (c->current_dropped + nsamples <= c->target_samples) {
			c->current_dropped += nsamples;
			left = 0;
		}
	} else {
		const uint32_t th = (uint32_t)(((uint64_t)c->target_samples * (uint64_t)c->current_pos) / (uint64_t)c->total_samples);
		uint32_t todrop = th > c->current_dropped ? th - c->current_dropped : 0;
		if (todrop > 0) {
			if (nsamples <= c->target_samples && frame_power(s, nsamples) < c->silent_threshold) {
				left = 0; /* an almost silent frame goes entirely :127-133 */
				todrop = nsamples;
			} else if (todrop * 8 < nsamples) {
				left = drop_smooth_samples(s, n, todrop); /* :134-136 */
			} else {
				left = 0; /* too much to hide: the whole frame :137-142 */
				todrop = nsamples;
			}
			c->current_dropped += todrop;
		}
	}
	if (c->current_pos >= c->total_samples) c->target_samples = 0; /* :149 */
	return left;
}
