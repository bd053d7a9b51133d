/*
 * oracle/pixconv.c -- TEST INFRASTRUCTURE ONLY (see ms2_oracle.h).  PARITY UNPINNED.
 *
 * Packed-format -> I420 conversions as MSPixConv reaches them through the libyuv
 * scaler implementation: yuv_scale() src/voip/msvideo.c:542-581 dispatches on the
 * SOURCE format only and calls YUY2ToI420 (:553), UYVYToI420 (:558),
 * RGB24ToJ420 (:562), RAWToI420 (:566), ARGBToI420 (:570).  libyuv is
 * un-vendored and unpinned (cmake/FindLibYUV.cmake); RGB24ToJ420 exists since
 * r1750 (2020), so this restates the portable C rows of that era
 * (source/convert.cc loops, source/row_common.cc: MAKEROWY / MAKEROWYJ with the
 * nested AVGB 2x2 average and +0x8080 chroma rounding, YUY2ToUVRow_C,
 * UYVYToUVRow_C).  libyuv names are memory order reversed: "RGB24" = B,G,R and
 * "ARGB" = B,G,R,A in memory; "RAW" = R,G,B.
 *
 * Destination: the frame layout of ms_yuv_buf_init (msvideo.c:85-99): Y stride w,
 * U/V stride w/2, contiguous, odd h rounded up for the plane sizes.
 */
#include "ms2_oracle.h"

#include <string.h>

#define AVGB(a, b) (((a) + (b) + 1) >> 1)

static int rgb_to_y(int r, int g, int b) { return (66 * r + 129 * g + 25 * b + 0x1080) >> 8; }
static int rgb_to_u(int r, int g, int b) { return (112 * b - 74 * g - 38 * r + 0x8080) >> 8; }
static int rgb_to_v(int r, int g, int b) { return (112 * r - 94 * g - 18 * b + 0x8080) >> 8; }
static int rgb_to_yj(int r, int g, int b) { return (77 * r + 150 * g + 29 * b + 128) >> 8; }
static int rgb_to_uj(int r, int g, int b) { return (127 * b - 84 * g - 43 * r + 0x8080) >> 8; }
static int rgb_to_vj(int r, int g, int b) { return (127 * r - 107 * g - 20 * b + 0x8080) >> 8; }

/* one row pair of a packed 4:2:2 source; y_off/u_off/v_off = byte positions inside the 4-byte macropixel */
static void row422(const uint8_t *s0, const uint8_t *s1, int w, int y0, int y1, int uo, int vo, uint8_t *dy0,
                   uint8_t *dy1, uint8_t *du, uint8_t *dv) {
	for (int x = 0; x < w; x += 2) {
		const uint8_t *p0 = s0 + 2 * x, *p1 = s1 + 2 * x;
		du[x / 2] = (uint8_t)((p0[uo] + p1[uo] + 1) >> 1);
		dv[x / 2] = (uint8_t)((p0[vo] + p1[vo] + 1) >> 1);
		dy0[x] = p0[y0];
		if (x + 1 < w) dy0[x + 1] = p0[y1];
		if (dy1) {
			dy1[x] = p1[y0];
			if (x + 1 < w) dy1[x + 1] = p1[y1];
		}
	}
}

static void rowrgb(const uint8_t *s0, const uint8_t *s1, int w, int bpp, int ro, int go, int bo, int jpeg,
                   uint8_t *dy0, uint8_t *dy1, uint8_t *du, uint8_t *dv) {
	for (int x = 0; x < w; ++x) {
		const uint8_t *p = s0 + bpp * x;
		dy0[x] = (uint8_t)(jpeg ? rgb_to_yj(p[ro], p[go], p[bo]) : rgb_to_y(p[ro], p[go], p[bo]));
		if (dy1) {
			const uint8_t *q = s1 + bpp * x;
			dy1[x] = (uint8_t)(jpeg ? rgb_to_yj(q[ro], q[go], q[bo]) : rgb_to_y(q[ro], q[go], q[bo]));
		}
	}
	for (int x = 0; x + 1 < w; x += 2) {
		const uint8_t *p0 = s0 + bpp * x, *p1 = s1 + bpp * x;
		const int ab = AVGB(AVGB(p0[bo], p1[bo]), AVGB(p0[bo + bpp], p1[bo + bpp]));
		const int ag = AVGB(AVGB(p0[go], p1[go]), AVGB(p0[go + bpp], p1[go + bpp]));
		const int ar = AVGB(AVGB(p0[ro], p1[ro]), AVGB(p0[ro + bpp], p1[ro + bpp]));
		du[x / 2] = (uint8_t)(jpeg ? rgb_to_uj(ar, ag, ab) : rgb_to_u(ar, ag, ab));
		dv[x / 2] = (uint8_t)(jpeg ? rgb_to_vj(ar, ag, ab) : rgb_to_v(ar, ag, ab));
	}
}

int orc_pixconv_to_i420(int fmt, const uint8_t *src, int src_stride, int w, int h, uint8_t *dst) {
	if (w < 2 || h < 1 || (w & 1)) return -1;
	const int h2 = h + (h & 1);
	uint8_t *dy = dst, *du = dst + (size_t)w * h2, *dv = du + (size_t)(w / 2) * (h2 / 2);
	for (int y = 0; y < h; y += 2) {
		const int last = (y + 1 >= h); /* odd height: the last row pairs with itself (stride 0 in libyuv) */
		const uint8_t *s0 = src + (ptrdiff_t)y * src_stride;
		const uint8_t *s1 = last ? s0 : s0 + src_stride;
		uint8_t *y0 = dy + (size_t)y * w, *y1 = last ? NULL : y0 + w;
		uint8_t *u = du + (size_t)(y / 2) * (w / 2), *v = dv + (size_t)(y / 2) * (w / 2);
		switch (fmt) {
			case ORC_PIX_YUY2: row422(s0, s1, w, 0, 2, 1, 3, y0, y1, u, v); break;
			case ORC_PIX_UYVY: row422(s0, s1, w, 1, 3, 0, 2, y0, y1, u, v); break;
			case ORC_PIX_BGR24: rowrgb(s0, s1, w, 3, 2, 1, 0, 1, y0, y1, u, v); break;  /* RGB24ToJ420 */
			case ORC_PIX_RGB24_RAW: rowrgb(s0, s1, w, 3, 0, 1, 2, 0, y0, y1, u, v); break; /* RAWToI420 */
			case ORC_PIX_BGRA32: rowrgb(s0, s1, w, 4, 2, 1, 0, 0, y0, y1, u, v); break;  /* ARGBToI420 */
			default: return -1;
		}
	}
	return 0;
}
