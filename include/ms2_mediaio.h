/*
 * ms2_mediaio.h -- dependency-free file I/O for parity runs on "the same WAV / YUV inputs" (header-only, plain C99).
 *
 *   WAV   PCM16 reader that sizes the audio from the FILE length minus the header, never from the `data` chunk's length
 *         field, the way the reference's player and audiodiff do (src/audiofilters/msfileplayer.c:98-150 walks the chunks,
 *         src/utils/audiodiff.c:73-76 computes fsize - hsize; tester/sounds/hello8000.wav carries a bogus data length:
 *         SURVEY.md A27); a PCM16 writer with the 44-byte header of src/audiofilters/msfilerec.c.
 *   I420  raw frames, planes contiguous as ms_yuv_buf_init lays them out (src/voip/msvideo.c:85-99): Y w*h, then U and V
 *         (w/2)*(h/2) each, an odd height rounded up for the chroma rows.
 *
 * Used by examples/ and by the tests' checker (oracle/audiodiff.c).  Not used by the kernel library or the plugin.
 */
#ifndef MS2_MEDIAIO_H
#define MS2_MEDIAIO_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct ms2_wav {
	int rate, nchannels;
	int nsamples;     /* per channel */
	int16_t *samples; /* interleaved, nsamples * nchannels; malloc'd, ms2_wav_free() */
	int header_bytes;
} ms2_wav;

static inline uint32_t ms2_le32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t ms2_le16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* 0 on success.  Only PCM16 is accepted (what every recording of the reference's tester is). */
static inline int ms2_wav_read(const char *path, ms2_wav *w) {
	unsigned char h[16];
	long fsize, hsize;
	uint32_t fmtlen;
	int guard;
	FILE *f = fopen(path, "rb");
	memset(w, 0, sizeof(*w));
	if (!f) return -1;
	if (fseek(f, 0, SEEK_END) != 0 || (fsize = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) goto bad;
	if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) goto bad;
	if (fread(h, 1, 8, f) != 8 || memcmp(h, "fmt ", 4) != 0) goto bad; /* the reference expects 'fmt ' right behind the RIFF header too */
	fmtlen = ms2_le32(h + 4);
	if (fmtlen < 16) goto bad;
	if (fread(h, 1, 16, f) != 16) goto bad;
	if (ms2_le16(h) != 1 || ms2_le16(h + 14) != 16) goto bad; /* PCM, 16 bits */
	w->nchannels = ms2_le16(h + 2);
	w->rate = (int)ms2_le32(h + 4);
	if (w->nchannels < 1) goto bad;
	if (fmtlen > 16 && fseek(f, (long)(fmtlen - 16), SEEK_CUR) != 0) goto bad;
	hsize = 12 + 8 + (long)fmtlen;
	for (guard = 0; guard < 30; ++guard) { /* chunks until 'data' (msfileplayer.c:128-146) */
		if (fread(h, 1, 8, f) != 8) goto bad;
		hsize += 8;
		if (memcmp(h, "data", 4) == 0) break;
		if (fseek(f, (long)ms2_le32(h + 4), SEEK_CUR) != 0) goto bad;
		hsize += (long)ms2_le32(h + 4);
	}
	if (guard == 30) goto bad;
	w->header_bytes = (int)hsize;
	w->nsamples = (int)((fsize - hsize) / (long)(sizeof(int16_t) * (size_t)w->nchannels)); /* from the file size */
	w->samples = (int16_t *)malloc(sizeof(int16_t) * (size_t)(w->nsamples > 0 ? w->nsamples : 1) * (size_t)w->nchannels);
	if (!w->samples) goto bad;
	if (fread(w->samples, sizeof(int16_t) * (size_t)w->nchannels, (size_t)w->nsamples, f) != (size_t)w->nsamples) goto bad;
	fclose(f);
	return 0; /* little-endian hosts only, like the rest of this repository */
bad:
	fclose(f);
	free(w->samples);
	memset(w, 0, sizeof(*w));
	return -1;
}

static inline void ms2_wav_free(ms2_wav *w) {
	free(w->samples);
	memset(w, 0, sizeof(*w));
}

static inline void ms2_put32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)v, p[1] = (unsigned char)(v >> 8), p[2] = (unsigned char)(v >> 16), p[3] = (unsigned char)(v >> 24); }
static inline void ms2_put16(unsigned char *p, unsigned v) { p[0] = (unsigned char)v, p[1] = (unsigned char)(v >> 8); }

static inline int ms2_wav_write(const char *path, int rate, int nchannels, const int16_t *samples, int nsamples) {
	unsigned char h[44];
	const uint32_t bytes = (uint32_t)nsamples * (uint32_t)nchannels * 2u;
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	memcpy(h, "RIFF", 4);
	ms2_put32(h + 4, bytes + 36);
	memcpy(h + 8, "WAVEfmt ", 8);
	ms2_put32(h + 16, 16);
	ms2_put16(h + 20, 1);
	ms2_put16(h + 22, (unsigned)nchannels);
	ms2_put32(h + 24, (uint32_t)rate);
	ms2_put32(h + 28, (uint32_t)rate * (uint32_t)nchannels * 2u);
	ms2_put16(h + 32, (unsigned)nchannels * 2u);
	ms2_put16(h + 34, 16);
	memcpy(h + 36, "data", 4);
	ms2_put32(h + 40, bytes);
	if (fwrite(h, 1, 44, f) != 44 || fwrite(samples, 2, (size_t)nsamples * (size_t)nchannels, f) != (size_t)nsamples * (size_t)nchannels) {
		fclose(f);
		return -1;
	}
	return fclose(f) == 0 ? 0 : -1;
}

/* ---- raw I420 */
static inline size_t ms2_i420_frame_bytes(int w, int h) {
	const int h2 = h + (h & 1); /* msvideo.c:87,:159 */
	return (size_t)w * (size_t)h2 * 3 / 2;
}
/* frame `index` of a raw I420 file into dst (ms2_i420_frame_bytes); 0 on success, -1 past the end / on error */
static inline int ms2_i420_read_frame(FILE *f, int w, int h, long index, uint8_t *dst) {
	const size_t n = ms2_i420_frame_bytes(w, h);
	if (fseek(f, (long)n * index, SEEK_SET) != 0) return -1;
	return fread(dst, 1, n, f) == n ? 0 : -1;
}
static inline int ms2_i420_append_frame(FILE *f, int w, int h, const uint8_t *src) {
	const size_t n = ms2_i420_frame_bytes(w, h);
	return fwrite(src, 1, n, f) == n ? 0 : -1;
}

#endif /* MS2_MEDIAIO_H */
