/* oracle/conference.c -- TEST INFRASTRUCTURE (see ms2_oracle.h): MSAudioConference's bookkeeping in mixer mode, restated
 * from /root/reference/src/voip/audioconference.c, and the windowed extremum MSVolume feeds for it.
 *
 * OrtpExtremum lives in oRTP (ortp/utils.h, src/utils.c), a dependency that is NOT under /root/reference and whose version the
 * tree does not pin -- parity unpinned for it: restated from its published behaviour as the reference uses it
 * (msvolume.c:115-116 periods 1000 / 30000 ms, :405-406 one record per update_energy, :139,146 get_current, :467-468 reset). */
#include "ms2_oracle.h"

#include <math.h>
#include <string.h>

/* ---- OrtpExtremum: a value is the extremum of the window it was recorded in; a record that arrives more than `period` ms
 * after the window's first one closes the window (its extremum becomes last_stable) and opens the next with itself */
void orc_extremum_init(OrcExtremum *e, int period) {
	e->period = period;
	orc_extremum_reset(e);
}
void orc_extremum_reset(OrcExtremum *e) {
	e->current_extremum = e->last_stable = 0;
	e->extremum_time = (uint64_t)-1;
}
static int extremum_check_init(OrcExtremum *e, uint64_t curtime, float value) {
	if (e->extremum_time != (uint64_t)-1 && (int)(curtime - e->extremum_time) > e->period) {
		e->last_stable = e->current_extremum;
		e->extremum_time = (uint64_t)-1;
	}
	if (e->extremum_time == (uint64_t)-1) {
		e->current_extremum = value;
		e->extremum_time = curtime;
		return 1;
	}
	return 0;
}
int orc_extremum_record_min(OrcExtremum *e, uint64_t curtime, float value) {
	int ret = extremum_check_init(e, curtime, value);
	if (value < e->current_extremum) {
		e->current_extremum = value;
		ret = 1;
	}
	return ret;
}
int orc_extremum_record_max(OrcExtremum *e, uint64_t curtime, float value) {
	int ret = extremum_check_init(e, curtime, value);
	if (value > e->current_extremum) {
		e->current_extremum = value;
		ret = 1;
	}
	return ret;
}
float orc_extremum_get_current(const OrcExtremum *e) { return e->current_extremum; }

float orc_volume_linear_to_dbm0(float linear) { /* msvolume.c:565-568 */
	if (linear == 0) return ORC_VOLUME_DB_LOWEST;
	return (float)(10 * log10f(linear));
}

/* ---- MSAudioConference, mixer mode.  Members are known by the mixer pin plumb_to_conf gave them. */
void orc_conference_init(OrcConference *c) { /* audioconference.c:67-92 */
	memset(c, 0, sizeof(*c));
	c->active_speaker = -1;
}
static int find_free_pin(const OrcConference *c) { /* :198-207: the LOWEST pin nobody is linked to */
	int i;
	for (i = 0; i < ORC_MIXER_MAX_CHANNELS; ++i)
		if (!c->plumbed[i]) return i;
	return -1; /* the reference aborts here (ms_fatal) */
}
int orc_conference_add_member(OrcConference *c, int muted) { /* :322-345 (the ticker detach / attach around it is the caller's) */
	int pin = find_free_pin(c);
	if (pin < 0) return -1;
	c->plumbed[pin] = 1;
	c->nmembers++;
	orc_conference_mute_member(c, pin, muted);
	return pin;
}
void orc_conference_remove_member(OrcConference *c, int pin) { /* :366-374 */
	if (pin < 0 || pin >= ORC_MIXER_MAX_CHANNELS || !c->plumbed[pin]) return;
	c->plumbed[pin] = 0;
	c->nmembers--;
	/* (obj->active_speaker keeps pointing at the endpoint that left: the next election with a winner replaces it, :460-464) */
}
void orc_conference_mute_member(OrcConference *c, int pin, int muted) { /* :376-388: MS_AUDIO_MIXER_SET_ACTIVE !muted */
	if (pin < 0 || pin >= ORC_MIXER_MAX_CHANNELS) return;
	c->muted[pin] = (uint8_t)(muted != 0);
}
int orc_conference_get_size(const OrcConference *c) { return c->nmembers; } /* :390-392 */

/* :394-418 ms_audio_conference_get_participant_volume: muted -> lowest, else (int) of MS_VOLUME_GET */
int orc_conference_participant_volume(const OrcConference *c, int pin, float volume_db) {
	if (pin < 0 || pin >= ORC_MIXER_MAX_CHANNELS || !c->plumbed[pin]) return ORC_VOLUMES_NOT_FOUND;
	if (c->muted[pin]) return ORC_VOLUME_DB_LOWEST;
	return (int)volume_db;
}

/* :419-464 ms_audio_conference_process_events, mixer mode.  max_db[pin] = what MS_VOLUME_GET_MAX of that member's MSVolume
 * returned; members are visited in the order they joined (order[0..nmembers), the conference's list).  Returns 1 when the
 * active speaker changed (the callback's moment); *winner_pin / *winner_db describe this poll's winner (-1 / lowest: nobody). */
int orc_conference_process_events(OrcConference *c, const int *order, const float *max_db, int *winner_pin, float *winner_db) {
	static const float audio_threshold_min_db = -30.0f; /* :31 */
	float max_db_over_member = ORC_VOLUME_DB_LOWEST;
	int winner = -1, i, changed = 0;
	for (i = 0; i < c->nmembers; ++i) {
		int pin = order[i];
		if (c->muted[pin]) continue; /* :445 */
		if (max_db[pin] > audio_threshold_min_db && max_db[pin] > max_db_over_member) { /* :449: strict, so the first of equals wins */
			max_db_over_member = max_db[pin];
			winner = pin;
		}
	}
	if (c->active_speaker != winner && winner != -1) { /* :460-464: silence elects nobody and keeps the last speaker */
		c->active_speaker = winner;
		changed = 1;
	}
	if (winner_pin) *winner_pin = winner;
	if (winner_db) *winner_db = max_db_over_member;
	return changed;
}
