/* oracle/video.c -- TEST INFRASTRUCTURE (see ms2_oracle.h). Parity unpinned.
 *
 * The reference's scaler arithmetic is libyuv I420Scale(kFilterBilinear)
 * (/root/reference/src/voip/msvideo.c:542-581) or ffmpeg sws_scale (:672-681);
 * both are un-vendored, unpinned (cmake/FindLibYUV.cmake, CMakeLists.txt:374)
 * and absent here, and they do not agree with each other bit-for-bit.  This
 * file fixes ONE definition, following libyuv's portable C path
 * (scale_common.cc: ScaleSlope, ScaleFilterCols_C, InterpolateRow_C;
 * scale.cc: ScalePlaneBilinearDown/Up) in 16.16 fixed point:
 *   - down-scale: step = (src<<16)/dst, start = step/2 - 0.5 (pixel centres);
 *   - up-scale:   step = ((src<<16) - 0x10001)/(dst-1), start = 0;
 *   - rows blended with an 8-bit fraction, +128 rounding;
 *   - columns blended with the 16-bit fraction, +0x8000 rounding.
 * Colour: BT.601 limited range with the in-tree Q13 integer constants of
 * /root/reference/src/voip/scaler_arm.S:54-63 (= 1.164, 1.596, 0.813, 0.391,
 * 2.018 of /root/reference/src/yuv2rgb.fs), +0.5 rounding, clamp to 0..255.
 * Frame layout: /root/reference/src/voip/msvideo.c:85-99 (ms_yuv_buf_init).
 */
#include "ms2_oracle.h"
#include <stdlib.h>
#include <string.h>

static int fixed_div(int num, int div) { return (int)(((int64_t)num << 16) / div); }
static int fixed_div1(int num, int div) { return (int)((((int64_t)num << 16) - 0x00010001) / (div - 1)); }

static void slope(int src, int dst, int *x, int *dx) {
	*x = 0;
	*dx = 0;
	if (dst <= src) {
		*dx = fixed_div(src, dst);
		*x = (*dx >> 1) - 32768; /* centre the filter */
	} else if (src > 1 && dst > 1) {
		*dx = fixed_div1(src, dst);
		*x = 0;
	}
}

static void blend_rows(uint8_t *dst, const uint8_t *r0, const uint8_t *r1, int w, int yf) {
	int i;
	if (yf == 0) {
		memcpy(dst, r0, (size_t)w);
		return;
	}
	for (i = 0; i < w; ++i) dst[i] = (uint8_t)((r0[i] * (256 - yf) + r1[i] * yf + 128) >> 8);
}

static void filter_cols(uint8_t *dst, const uint8_t *src, int sw, int dw, int x, int dx) {
	int j;
	for (j = 0; j < dw; ++j) {
		int xi = x >> 16, f = x & 0xffff, xn, a, b;
		if (xi < 0) { /* only reachable for x < 0, not on the down-scale path */
			xi = 0;
			f = 0;
		}
		xn = xi + 1 < sw ? xi + 1 : sw - 1;
		a = src[xi];
		b = src[xn];
		dst[j] = (uint8_t)(a + ((f * (b - a) + 0x8000) >> 16));
		x += dx;
	}
}

void orc_scale_plane_bilinear(const uint8_t *src, int src_stride, int sw, int sh, uint8_t *dst,
                              int dst_stride, int dw, int dh) {
	int x, dx, y, dy, j;
	const int max_y = (sh - 1) << 16;
	uint8_t *row = (uint8_t *)malloc((size_t)sw);
	slope(sw, dw, &x, &dx);
	slope(sh, dh, &y, &dy);
	if (y > max_y) y = max_y;
	if (y < 0) y = 0;
	for (j = 0; j < dh; ++j) {
		int yi = y >> 16, yf = (y >> 8) & 255;
		const uint8_t *r0 = src + (size_t)yi * src_stride;
		const uint8_t *r1 = (yi + 1 < sh) ? r0 + src_stride : r0;
		blend_rows(row, r0, r1, sw, yf);
		filter_cols(dst + (size_t)j * dst_stride, row, sw, dw, x, dx);
		y += dy;
		if (y > max_y) y = max_y;
	}
	free(row);
}

void orc_i420_scale(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
	/* ms_yuv_buf_init: odd heights rounded up, chroma w/2 x h/2 */
	int sh2 = (sh & 1) ? sh + 1 : sh, dh2 = (dh & 1) ? dh + 1 : dh;
	int scw = sw / 2, sch = sh2 / 2, dcw = dw / 2, dch = dh2 / 2;
	const uint8_t *su = src + (size_t)sw * sh2, *sv = su + (size_t)scw * sch;
	uint8_t *du = dst + (size_t)dw * dh2, *dv = du + (size_t)dcw * dch;
	orc_scale_plane_bilinear(src, sw, sw, sh, dst, dw, dw, dh);
	orc_scale_plane_bilinear(su, scw, scw, sch, du, dcw, dcw, dch);
	orc_scale_plane_bilinear(sv, scw, scw, sch, dv, dcw, dcw, dch);
}

static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void orc_i420_to_rgb24(const uint8_t *src, int w, int h, uint8_t *rgb, int rgb_stride) {
	int h2 = (h & 1) ? h + 1 : h, cw = w / 2;
	const uint8_t *py = src, *pu = src + (size_t)w * h2, *pv = pu + (size_t)cw * (h2 / 2);
	int x, y;
	for (y = 0; y < h; ++y) {
		uint8_t *o = rgb + (size_t)y * rgb_stride;
		for (x = 0; x < w; ++x) {
			int cx = x >> 1 < cw ? x >> 1 : cw - 1;
			int c = py[(size_t)y * w + x] - 16;
			int d = pu[(size_t)(y >> 1) * cw + cx] - 128;
			int e = pv[(size_t)(y >> 1) * cw + cx] - 128;
			int yy = 9535 * c + 4096;
			o[3 * x + 0] = clamp8((yy + 13074 * e) >> 13);
			o[3 * x + 1] = clamp8((yy - 3203 * d - 6660 * e) >> 13);
			o[3 * x + 2] = clamp8((yy + 16531 * d) >> 13);
		}
	}
}

void orc_i420_scale_to_rgb24(const uint8_t *src, int sw, int sh, uint8_t *rgb, int dw, int dh) {
	int dh2 = (dh & 1) ? dh + 1 : dh;
	uint8_t *tmp = (uint8_t *)malloc((size_t)dw * dh2 + 2 * (size_t)(dw / 2) * (dh2 / 2));
	orc_i420_scale(src, sw, sh, tmp, dw, dh);
	orc_i420_to_rgb24(tmp, dw, dh, rgb, dw * 3);
	free(tmp);
}
