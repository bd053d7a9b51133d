/* oracle/audiodiff.c -- TEST INFRASTRUCTURE (see ms2_oracle.h).
 * The reference's recording-comparison metrics, restated in plain C on WAV files: what its testers grade an echo
 * canceller or a codec chain with (/root/reference/src/utils/audiodiff.c; users: tester/mediastreamer2_aec3_tester.c,
 * tools/msaudiocmp.c).
 *
 *   orc_audio_diff                          ms_audio_diff                            audiodiff.c:578-651
 *   orc_audio_compare_silence_and_speech    ms_audio_compare_silence_and_speech      :442-576
 *   orc_audio_energy                        ms_audio_energy                          :653-682
 *
 * Same quantities, same integer correlations (exact in int64), same float rounding of the normalised correlation
 * (xcorr is stored as float, :197), same first-maximum rule (strict '>', :205), file sizes from the file length
 * (:73-76).  Structure is this file's own: recordings are loaded whole (include/ms2_mediaio.h) and windows are taken
 * from memory; the two moving averages of the silence detector are running sums (exact: every term is a multiple of
 * 2^-15 and the sums stay below 2^12).  oracle/audiodiff.py is the numpy form of the same; tests compare the two. */
#include "ms2_oracle.h"
#include "../include/ms2_mediaio.h"

#include <math.h>

static int64_t dot16(const int16_t *a, const int16_t *b, int n, int step) {
	int64_t acc = 0;
	for (int i = 0; i < n; ++i) acc += (int64_t)a[i * step] * b[i * step];
	return acc;
}

/* compute_cross_correlation :184-216.  s2 holds (n1 + nshifts - 1) * step samples at least.
 * Returns the shift with the largest |<s1, s2[shift ...]>| (the first one), the normalised values in xc (float). */
static int xcorr_scan(const int16_t *s1, int n1, const int16_t *s2, float *xc, int nshifts, int step, int64_t *e1) {
	const int64_t norm1 = dot16(s1, s1, n1, step);
	int64_t norm2 = dot16(s2, s2, n1 - 1, step); /* window energy without its last sample; added per shift */
	int64_t best = 0;
	int arg = 0;
	for (int i = 0; i < nshifts; ++i) {
		const int64_t last = s2[step * (i + n1 - 1)];
		norm2 += last * last;
		int64_t num = dot16(s1, s2 + i * step, n1, step);
		const double den = sqrt((double)norm1 * (double)norm2);
		xc[i] = den > 0 ? (float)((double)num / den) : 1.0f;
		if (num < 0) num = -num;
		if (num > best) best = num, arg = i;
		norm2 -= (int64_t)s2[step * i] * s2[step * i];
	}
	if (e1) *e1 = norm1;
	return arg;
}

/* _ms_audio_diff_one_chunk :218-287: position of the best alignment (relative to the padding) and its similarity */
static int diff_one_chunk(const int16_t *s1, const int16_t *s2_padded, int nsamples, int max_shift, int nch, double *sim,
                          int64_t *e1) {
	const int nx = 2 * max_shift;
	int pos;
	if (nch == 2) {
		float *r = (float *)calloc((size_t)(nx > 0 ? nx : 1), sizeof(float));
		float *l = (float *)calloc((size_t)(nx > 0 ? nx : 1), sizeof(float));
		int64_t er = 0, el = 0;
		double best = 0;
		xcorr_scan(s1, nsamples, s2_padded, r, nx, 2, &er);
		xcorr_scan(s1 + 1, nsamples, s2_padded + 1, l, nx, 2, &el);
		pos = 0;
		for (int i = 0; i <= max_shift && i < nx; ++i) { /* :259-268: only the first half is searched */
			const float v = r[i] * r[i] + l[i] * l[i];
			if (v > best) best = v, pos = i;
		}
		*sim = sqrt(best / 2);
		if (e1) *e1 = (er + el) / 2;
		free(r);
		free(l);
		return pos - max_shift;
	}
	{
		float *xc = (float *)calloc((size_t)(nx > 0 ? nx : 1), sizeof(float));
		const int arg = xcorr_scan(s1, nsamples, s2_padded, xc, nx, 1, e1);
		*sim = nx > 0 ? xc[arg] : 0;
		free(xc);
		return arg - max_shift;
	}
}

/* _ms_audio_diff_chunked :289-343: energy-weighted similarity over chunks, discounted by the spread of the positions */
static int diff_chunked(const int16_t *s1, int n1, const int16_t *s2_padded, int max_shift, int chunk, int nch, int rate,
                        double *sim) {
	const int nchunks = (n1 + chunk) / chunk;
	int *posv = (int *)calloc((size_t)nchunks, sizeof(int));
	int64_t *env = (int64_t *)calloc((size_t)nchunks, sizeof(int64_t));
	double cum = 0, var = 0;
	int64_t cumpos = 0, tot = 0;
	int k = 0, pos;
	(void)rate;
	for (int at = 0; at < n1; at += chunk, ++k) {
		const int n = n1 - at < chunk ? n1 - at : chunk;
		double cs = 0;
		int64_t ce = 0;
		posv[k] = diff_one_chunk(s1 + at * nch, s2_padded + at * nch, n, max_shift, nch, &cs, &ce);
		env[k] = ce;
		cum += cs * (double)ce;
		cumpos += (int64_t)posv[k] * ce;
		tot += ce;
	}
	pos = tot ? (int)(cumpos / tot) : 0;
	for (int i = 0; i < k; ++i) {
		const double t = (posv[i] - pos) * ((double)env[i] / (double)tot);
		var += t * t;
	}
	var = sqrt(var) / (double)max_shift;
	*sim = (cum / (double)tot) * (1 - var);
	free(posv);
	free(env);
	return pos;
}

static int clamp_percent(int p) { return p < 1 ? 1 : (p > 100 ? 100 : p); }

/* zero_pad zeros, then `size` samples of w starting at `start` (file_info_read_short :103-121), zeros again: malloc'd */
static int16_t *window(const ms2_wav *w, int zero_pad, int start, int size) {
	const int nch = w->nchannels;
	int16_t *b = (int16_t *)calloc((size_t)(size + 2 * zero_pad) * (size_t)nch + 1, sizeof(int16_t));
	if (b && size > 0) memcpy(b + (size_t)zero_pad * nch, w->samples + (size_t)start * nch, sizeof(int16_t) * (size_t)size * nch);
	return b;
}

int orc_audio_diff(const char *ref_file, const char *matched_file, double *ret, int max_shift_percent, int chunk_size_ms) {
	ms2_wav a, b;
	int err = -1;
	*ret = 0;
	if (ms2_wav_read(ref_file, &a) != 0) return 0; /* sic: :593 returns 0 when the reference file cannot be opened */
	if (ms2_wav_read(matched_file, &b) != 0) {
		ms2_wav_free(&a);
		return -1;
	}
	if (a.rate == b.rate && a.nchannels == b.nchannels && a.nsamples > 0 && b.nsamples > 0) {
		const int nmin = a.nsamples < b.nsamples ? a.nsamples : b.nsamples;
		const int max_shift = nmin * clamp_percent(max_shift_percent) / 100;
		const int endpad = a.nsamples > b.nsamples ? a.nsamples - b.nsamples : 0;
		/* file_info_read(fi2, max_shift, endpad) :80-96: pads at both ends, the end padding counts as audio */
		int16_t *s2 = (int16_t *)calloc((size_t)(b.nsamples + 2 * max_shift + 2 * endpad) * (size_t)b.nchannels + 1, sizeof(int16_t));
		memcpy(s2 + (size_t)max_shift * b.nchannels, b.samples, sizeof(int16_t) * (size_t)b.nsamples * b.nchannels);
		if (chunk_size_ms == 0) diff_one_chunk(a.samples, s2, a.nsamples, max_shift, a.nchannels, ret, NULL);
		else diff_chunked(a.samples, a.nsamples, s2, max_shift, chunk_size_ms * a.rate / 1000, a.nchannels, a.rate, ret);
		free(s2);
		err = 0;
	}
	ms2_wav_free(&a);
	ms2_wav_free(&b);
	return err;
}

/* ms_audio_compute_energy_in_silence :349-407 (mono): mask of the reference's silences, energy of s2 there */
static double energy_in_silence(const int16_t *s1, const int16_t *s2, int n, unsigned char *mask) {
	/* |s| / 32768 averaged over +-200 samples < 0.001, then a majority vote over +-1400 samples */
	int64_t *pre = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
	int *cnt = (int *)calloc((size_t)n + 1, sizeof(int));
	unsigned char *raw = (unsigned char *)calloc((size_t)n + 1, 1);
	double en = 0;
	for (int i = 0; i < n; ++i) pre[i + 1] = pre[i] + (s1[i] < 0 ? -(int64_t)s1[i] : (int64_t)s1[i]);
	for (int i = 0; i < n; ++i) {
		const int lo = i - 200 < 0 ? 0 : i - 200, hi = i + 201 > n ? n : i + 201;
		/* sum/32768/k < 0.001, in the reference's doubles: the division by 32768 is exact, the one by k is not -- keep it */
		const double mean = ((double)(pre[hi] - pre[lo]) / 32768.) / (double)(hi - lo);
		raw[i] = mean < 0.001;
	}
	for (int i = 0; i < n; ++i) cnt[i + 1] = cnt[i] + raw[i];
	for (int i = 0; i < n; ++i) {
		const int lo = i - 1400 < 0 ? 0 : i - 1400, hi = i + 1401 > n ? n : i + 1401;
		mask[i] = !((double)(cnt[hi] - cnt[lo]) / (double)(hi - lo) < 0.5);
	}
	for (int i = 0; i < n; ++i)
		if (mask[i]) {
			const double s = (double)s2[i] / 32768.;
			en += s * s;
		}
	free(pre);
	free(cnt);
	free(raw);
	return en;
}

/* ms_audio_compute_similarity_in_speech :413-440 */
static double similarity_in_speech(const int16_t *s1, const int16_t *s2, int n, const unsigned char *mask) {
	int nspeech = 0, j = 0;
	double sim = 0;
	for (int i = 0; i < n; ++i) nspeech += !mask[i];
	{
		const int max_shift = (int)((double)nspeech / 100.);
		int16_t *a = (int16_t *)calloc((size_t)nspeech + 1, sizeof(int16_t));
		int16_t *b = (int16_t *)calloc((size_t)nspeech + 2 * (size_t)max_shift + 1, sizeof(int16_t));
		for (int i = 0; i < n; ++i)
			if (!mask[i]) a[j] = s1[i], b[j + max_shift] = s2[i], ++j;
		diff_one_chunk(a, b, nspeech, max_shift, 1, &sim, NULL);
		free(a);
		free(b);
	}
	return sim;
}

int orc_audio_compare_silence_and_speech(const char *ref_file, const char *matched_file, double *ret, double *energy,
                                         int max_shift_percent, int chunk_size_ms, int start_time_short_ms,
                                         int stop_time_short_ms, int start_time_ms) {
	ms2_wav a, b;
	int err = -1;
	*ret = 0;
	*energy = 0;
	if (ms2_wav_read(ref_file, &a) != 0) return -1;
	if (ms2_wav_read(matched_file, &b) != 0) {
		ms2_wav_free(&a);
		return -1;
	}
	if (a.rate == b.rate && a.nchannels == b.nchannels && a.nsamples > 0 && b.nsamples > 0) {
		const int tested = stop_time_short_ms - start_time_short_ms;
		if ((double)tested < (double)a.nsamples / (double)a.rate * 1000 && (double)tested < (double)b.nsamples / (double)b.rate * 1000) {
			/* (1) align on a short window: the reference side padded by max_shift, the matched one as it is (:514-530) */
			const int max_shift = tested * a.rate / 1000 * clamp_percent(max_shift_percent) / 100;
			const int start = (int)((double)start_time_short_ms / 1000. * (double)a.rate);
			const int size = (int)((double)tested / 1000. * (double)a.rate);
			if (start + size <= a.nsamples && start + size <= b.nsamples) {
				int16_t *wb = window(&b, 0, start, size), *wa = window(&a, max_shift, start, size);
				int pos;
				if (chunk_size_ms == 0) pos = diff_one_chunk(wb, wa, size, max_shift, a.nchannels, ret, NULL);
				else pos = diff_chunked(wb, size, wa, max_shift, chunk_size_ms * a.rate / 1000, a.nchannels, a.rate, ret);
				free(wa);
				free(wb);
				{
					/* (2) the recordings from start_time_ms on, shifted against each other by that position (:535-556) */
					const int pad_a = pos < 0 ? -pos : 0, pad_b = pos < 0 ? 0 : pos;
					const int s0 = (int)(start_time_ms / 1000. * (double)a.rate);
					const int na = a.nsamples - s0, nb = b.nsamples - (int)(start_time_ms / 1000. * (double)b.rate);
					int16_t *fa = window(&a, pad_a, s0, na), *fb = window(&b, pad_b, s0, nb);
					const int n = na < nb ? na : nb;
					unsigned char *mask = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
					/* (3) energy of the matched file where the reference is silent, similarity where it speaks (mono) */
					*energy = energy_in_silence(fa, fb, n, mask);
					*ret = similarity_in_speech(fa, fb, n, mask);
					free(mask);
					free(fa);
					free(fb);
					err = 0;
				}
			}
		}
	}
	ms2_wav_free(&a);
	ms2_wav_free(&b);
	return err;
}

int orc_audio_energy(const char *file, double *energy) {
	ms2_wav w;
	double en = 0;
	if (ms2_wav_read(file, &w) != 0) return 0; /* sic: :659 */
	if (w.nsamples == 0) {
		ms2_wav_free(&w);
		return -1;
	}
	for (int i = 0; i < w.nsamples; ++i) { /* the reference walks nsamples entries of the interleaved buffer, :674 */
		const double s = (double)w.samples[i] / 32768.;
		en += s * s;
	}
	*energy = en;
	ms2_wav_free(&w);
	return 0;
}
