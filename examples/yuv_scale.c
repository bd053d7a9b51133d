/*
 * yuv_scale.c -- "the same YUV inputs" through the kernel library, in plain C: every frame of a raw I420 file scaled
 * with the MSScalerDesc replacement (include/mediastreamer2/msvideo.h:473-478, src/voip/msvideo.c:542-581).
 *
 *   yuv_scale in.yuv w h out dw dh [rgb]
 *
 * Writes raw I420 frames (ms_yuv_buf_init layout, src/voip/msvideo.c:85-99), or packed R,G,B rows with "rgb" (the
 * display-side conversion the north_star names: BT.601 limited range, src/voip/scaler_arm.S:54-63).
 * Build: gcc -std=c99 -Iinclude examples/yuv_scale.c -Lmediastreamer2_amd -lmsmi355x -o yuv_scale
 */
#include "ms2_mediaio.h"
#include "msmi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
	if (argc < 7) {
		fprintf(stderr, "usage: %s in.yuv w h out dw dh [rgb]\n", argv[0]);
		return 2;
	}
	const int w = atoi(argv[2]), h = atoi(argv[3]), dw = atoi(argv[5]), dh = atoi(argv[6]);
	const int rgb = argc > 7 && strcmp(argv[7], "rgb") == 0;
	FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[4], "wb");
	if (!in || !out) {
		fprintf(stderr, "cannot open the files\n");
		return 2;
	}
	mi_ctx *ctx = NULL;
	mi_scaler *sc = NULL;
	if (mi_ctx_create(0, NULL, &ctx) != MI_OK || mi_scaler_create(ctx, w, h, dw, dh, rgb ? MI_PIX_RGB24 : MI_PIX_I420, &sc) != MI_OK) {
		fprintf(stderr, "%s\n", mi_last_error());
		return 1;
	}
	const size_t sb = mi_scaler_src_bytes(sc), db = mi_scaler_dst_bytes(sc);
	if (sb != ms2_i420_frame_bytes(w, h)) {
		fprintf(stderr, "frame size mismatch\n");
		return 1;
	}
	uint8_t *src = (uint8_t *)malloc(sb), *dst = (uint8_t *)malloc(db);
	long k = 0;
	for (; ms2_i420_read_frame(in, w, h, k, src) == 0; ++k) {
		if (mi_scaler_process_host(sc, 1, src, sb, dst, db) != MI_OK) {
			fprintf(stderr, "%s\n", mi_last_error());
			return 1;
		}
		if (fwrite(dst, 1, db, out) != db) return 1;
	}
	fclose(in);
	fclose(out);
	mi_scaler_destroy(sc);
	mi_ctx_destroy(ctx);
	printf("ok %ld frames %dx%d -> %dx%d %s\n", k, w, h, dw, dh, rgb ? "RGB24" : "I420");
	return 0;
}
