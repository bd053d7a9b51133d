/* oracle/mixer.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 * One conference-tick of MSAudioMixer, restated from
 * /root/reference/src/audiofilters/audiomixer.c. */
#include "ms2_oracle.h"
#include <stdlib.h>
#include <string.h>

/* audiomixer.c:40-44 -- symmetric clamp, never -32768 */
static int16_t sat16(int32_t s) {
	return (int16_t)(s > 32767 ? 32767 : (s < -32767 ? -32767 : s));
}

void orc_mixer_tick(const int16_t *in, const uint8_t *has_data, const float *gain,
                    const uint8_t *active, const uint8_t *out_enabled, int nmembers,
                    int nsamples, int conf_mode, int16_t *out, int32_t *sum_out) {
	int32_t *sum = (int32_t *)calloc((size_t)nsamples, sizeof(int32_t)); /* :301 memset */
	int16_t *contrib = (int16_t *)malloc((size_t)nmembers * nsamples * sizeof(int16_t));
	int m, i;

	/* :304-314 + channel_process_in :78-90 */
	for (m = 0; m < nmembers; ++m) {
		int16_t *c = contrib + (size_t)m * nsamples;
		if (!has_data[m]) { /* short read: whole tick is zeros (:88) */
			memset(c, 0, (size_t)nsamples * 2);
			continue;
		}
		memcpy(c, in + (size_t)m * nsamples, (size_t)nsamples * 2);
		if (active[m]) {
			if (gain[m] != 1.0f) { /* :82, apply_gain :46-51, in place on the stored copy */
				for (i = 0; i < nsamples; ++i) c[i] = sat16((int)(gain[m] * (float)c[i]));
			}
			for (i = 0; i < nsamples; ++i) sum[i] += c[i]; /* accumulate :33-38 */
		}
	}

	if (conf_mode == 0) {
		/* :321-334 one saturated block shared by every enabled output */
		for (i = 0; i < nsamples; ++i) out[i] = sat16(sum[i]);
	} else {
		/* :336-343 + channel_process_out :113-130 */
		for (m = 0; m < nmembers; ++m) {
			int16_t *o = out + (size_t)m * nsamples;
			const int16_t *c = contrib + (size_t)m * nsamples;
			if (!out_enabled[m]) continue;
			if (active[m]) {
				for (i = 0; i < nsamples; ++i) o[i] = sat16(sum[i] - (int32_t)c[i]);
			} else {
				for (i = 0; i < nsamples; ++i) o[i] = sat16(sum[i]);
			}
		}
	}
	if (sum_out) memcpy(sum_out, sum, (size_t)nsamples * sizeof(int32_t));
	free(contrib);
	free(sum);
}
