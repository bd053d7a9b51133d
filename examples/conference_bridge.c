/*
 * examples/conference_bridge.c -- the main loop of a G.711 conference bridge on the C ABI (include/msmi355x.h): what a
 * media server runs instead of one MSTicker graph per call (src/voip/audioconference.c builds those in the reference).
 * Plain C99; links against libmsmi355x.so only.
 *
 *   cc -std=c99 -Iinclude examples/conference_bridge.c -Lmediastreamer2_amd -lmsmi355x -Wl,-rpath,$PWD/mediastreamer2_amd
 *
 * Every 10 ms: the RTP side hands over one PCMA payload of 80 bytes per leg (or marks the leg lost), and takes back the
 * 80 bytes to send to that leg: everybody else in its conference, echo-cancelled, levelled, mixed, down-sampled, encoded.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "msmi355x.h"

/* stand-ins for the RTP side of a real server */
static int rtp_receive(int leg, uint8_t payload[80]) {
	memset(payload, 0xD5, 80); /* A-law silence */
	return leg % 50 != 7;      /* one leg in fifty loses its packet this tick */
}
static void rtp_send(int leg, const uint8_t payload[80]) {
	(void)leg;
	(void)payload;
}

int main(void) {
	mi_ctx *ctx;
	mi_session *se;
	mi_session_config cfg;
	int legs = 64 * 32, tick, leg;

	if (mi_ctx_create(0, NULL, &ctx) != MI_OK) {
		fprintf(stderr, "no MI355X: %s\n", mi_last_error()); /* there is no CPU fallback */
		return 1;
	}
	mi_session_default_config(&cfg);
	cfg.nstreams = legs;
	cfg.members_per_conference = 32;
	cfg.in_rate = 8000, cfg.mic_codec = MI_SESSION_PCMA; /* MSAlawDec -> MSGenericPLC -> MSResample 8k->48k */
	cfg.plc = 1;
	cfg.rate = 48000, cfg.tail_ms = 128, cfg.agc = 1;     /* MSSpeexEC, MSVolume, MSAudioMixer at 48 kHz   */
	cfg.out_rate = 8000, cfg.out_codec = MI_SESSION_PCMA; /* MSResample 48k->8k -> MSAlawEnc               */
	cfg.ref_loopback = 1, cfg.ref_delay_ms = 40;          /* far-end reference = what the leg was sent      */
	if (mi_session_create(ctx, &cfg, &se) != MI_OK) {
		fprintf(stderr, "mi_session_create: %s\n", mi_last_error());
		return 1;
	}
	for (tick = 0; tick < 300; ++tick) { /* a real server paces this loop at 10 ms */
		int16_t *mic, *ref;
		uint8_t *events, *codes;
		const int16_t *out;
		if (mi_session_in_flight(se) == 3) { /* three ticks in flight: upload | kernels | download overlap */
			mi_session_collect(se, &out);
			for (leg = 0; leg < legs; ++leg) rtp_send(leg, (const uint8_t *)out + 80 * leg);
		}
		mi_session_acquire(se, &mic, &ref); /* pinned staging, filled in place; ref is NULL with ref_loopback */
		mi_session_events(se, &events);
		codes = (uint8_t *)mic;
		for (leg = 0; leg < legs; ++leg)
			if (!rtp_receive(leg, codes + 80 * leg)) events[leg] = MI_PLC_CONCEAL;
		if (mi_session_submit(se) != MI_OK) {
			fprintf(stderr, "mi_session_submit: %s\n", mi_last_error());
			return 1;
		}
	}
	while (mi_session_in_flight(se)) {
		const int16_t *out;
		mi_session_collect(se, &out);
	}
	mi_session_destroy(se);
	mi_ctx_destroy(ctx);
	puts("ok");
	return 0;
}
