/* oracle/volume.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 * MSVolume's per-chunk DSP restated from
 * /root/reference/src/audiofilters/msvolume.c.  Must be compiled with
 * -ffp-contract=off: the reference's x86-64 build evaluates every float
 * expression unfused. */
#include "ms2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* msvolume.c:37-46 */
static const float k_max_e = (32768 * 0.7f);
static const float k_coef = 0.2f;
static const float k_vol_upramp = 0.4f;
static const float k_vol_downramp = 0.4f;
static const float k_en_weight = 4.0;
static const float k_noise_thres = 0.1f;
static const float k_transmit_thres = 4;
static const float k_min_ng_floorgain = 0.005f;
static const float k_agc_threshold = 0.5;

void orc_volume_init(OrcVolume *v) { /* msvolume.c:88-118 */
	memset(v, 0, sizeof(*v));
	v->static_gain = v->gain = v->target_gain = 1;
	v->vol_upramp = k_vol_upramp;
	v->vol_fast_upramp = k_vol_upramp * 3;
	v->vol_downramp = k_vol_downramp;
	v->ea_thres = k_noise_thres;
	v->ea_transmit_thres = k_transmit_thres;
	v->force = k_en_weight;
	v->sustain_time = 200;
	v->sample_rate = 8000;
	v->nsamples = 80;
	v->ng_cut_time = 400;
	v->ng_threshold = k_noise_thres;
	v->ng_floorgain = k_min_ng_floorgain;
	v->ng_gain = 1;
}

void orc_volume_set_rate(OrcVolume *v, int rate) {
	v->sample_rate = rate;                        /* :150-154 */
	v->nsamples = (int)(0.01 * (float)rate);      /* :450 (volume_preprocess) */
}

void orc_volume_set_gain(OrcVolume *v, float g) { v->gain = v->target_gain = v->static_gain = g; }

void orc_volume_set_db_gain(OrcVolume *v, float d) {
	v->gain = v->static_gain = (float)pow(10, d / 10); /* :265, power ratio */
}

void orc_volume_enable_noise_gate(OrcVolume *v, int on) {
	v->noise_gate_enabled = on;
	if (on) v->gain = v->target_gain = v->ng_floorgain;
}

static int16_t sat16(int val) { /* :382-384 */
	return (int16_t)((val > 32767) ? 32767 : ((val < -32767) ? -32767 : val));
}

/* :388-407 (extremum recording is host-side bookkeeping, not restated here) */
static void update_energy(OrcVolume *v, const int16_t *sig, int n) {
	float acc = 0, en;
	int pk = 0, i;
	for (i = 0; i < n; ++i) {
		int s = sig[i];
		int a;
		acc += s * s; /* int product -> float, sequential float32 accumulation */
		a = abs(s);
		if (a > pk) pk = a;
	}
	en = (float)((sqrt(acc / n) + 1) / k_max_e);
	v->energy = (en * k_coef) + v->energy * (1.0f - k_coef);
	v->level_pk = (float)pk / k_max_e;
	v->instant_energy = en;
}

/* :201-238 */
static float echo_avoider(OrcVolume *v, int n, float peer_energy) {
	float peer_e = peer_energy, peer_pk = peer_energy, ratio;
	if (peer_pk > v->lt_speaker_en) v->lt_speaker_en = peer_pk;
	else v->lt_speaker_en = (0.005f * peer_pk) + (0.995f * v->lt_speaker_en);
	ratio = (v->energy / (v->lt_speaker_en + v->ea_thres));
	if (peer_e > v->ea_thres) {
		if (ratio > v->ea_transmit_thres) {
			v->target_gain = v->static_gain;
			v->fast_upramp = 1;
		} else {
			v->target_gain = v->static_gain / (1 + (peer_e * v->force)); /* compute_gain :188-191 */
			v->sustain_dur = v->sustain_time;
		}
	} else {
		if (v->sustain_dur > 0) {
			v->sustain_dur -= (n * 1000) / v->sample_rate;
		} else {
			v->target_gain = v->static_gain;
			v->fast_upramp = 1;
		}
	}
	return v->target_gain;
}

/* :240-260 */
static void noise_gate(OrcVolume *v, float energy, int n) {
	float tgain = v->ng_floorgain;
	if (energy > v->ng_threshold) {
		v->ng_noise_dur = v->ng_cut_time;
		tgain = 1.0;
	} else if (v->ng_noise_dur > 0) {
		v->ng_noise_dur -= (n * 1000) / v->sample_rate;
		tgain = 1.0;
	}
	v->ng_gain = v->ng_gain * 0.75f + tgain * 0.25f;
}

/* :409-445 */
static void apply_gain(OrcVolume *v, int16_t *s, int n, float tgain) {
	float gain;
	int32_t intgain;
	int i;
	if (v->gain < tgain) {
		if (v->gain < v->ng_floorgain) v->gain = v->ng_floorgain;
		v->gain *= 1 + (v->fast_upramp ? v->vol_fast_upramp : v->vol_upramp);
		if (v->gain > tgain) v->gain = tgain;
	} else if (v->gain > tgain) {
		v->gain *= 1 - v->vol_downramp;
		if (v->gain < tgain) v->gain = tgain;
		v->fast_upramp = 0;
	}
	gain = v->gain * v->ng_gain;
	intgain = (int32_t)(gain * 4096);
	if (v->remove_dc) {
		int dc = 0;
		for (i = 0; i < n; ++i) {
			dc += s[i];
			s[i] = sat16(((s[i] - v->dc_offset) * intgain) / 4096);
		}
		v->dc_offset = (v->dc_offset * 7 + dc * 2 / (2 * n)) / 8; /* :439, divisor is the BYTE count */
	} else if (gain != 1) {
		for (i = 0; i < n; ++i) s[i] = sat16((s[i] * intgain) / 4096); /* C division: toward zero */
	}
}

/* bodies of the two loops of volume_process :480-513 */
void orc_volume_chunk(OrcVolume *v, int16_t *samples, int n, float peer_energy) {
	float target;
	update_energy(v, samples, n);
	target = v->static_gain;
	if (v->has_peer) target = echo_avoider(v, n, peer_energy);
	if (v->agc_enabled) target /= (k_agc_threshold + v->level_pk) / 1; /* volume_agc_process :172-184 */
	if (v->noise_gate_enabled) noise_gate(v, v->instant_energy, n);
	apply_gain(v, samples, n, target);
}
