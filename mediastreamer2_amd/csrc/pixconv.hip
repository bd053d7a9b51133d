// pixconv.hip -- batched MSPixConv: packed YUY2 / UYVY / RGB24 / RGBA32 frames -> I420, for gfx950.
//
// Replaces, for a whole batch of frames per launch, what pixconv_process
// (src/videofilters/pixconv.c:62-94) gets from ms_scaler_process with the libyuv scaler
// implementation: yuv_scale (src/voip/msvideo.c:542-581) dispatches on the SOURCE format and calls
// YUY2ToI420 / UYVYToI420 / RGB24ToJ420 / RAWToI420 / ARGBToI420 (libyuv, un-vendored, unpinned; the
// portable C rows of the r1750+ era: BT.601 limited-range Q8 luma (66,129,25)+0x1080, chroma on the
// nested-AVGB 2x2 average with +0x8080, the JPEG full-range set (77,150,29)+128 for RGB24ToJ420, and
// the rounded vertical chroma average of the 4:2:2 rows).  Integer only; parity is checked bit for bit in tests/test_gpu_pixconv.py.
//
// Mapping: byte streaming, HBM-bound.  One lane owns 8 pixels of one row PAIR: it loads the two
// 16/24/32-byte row segments with 8/16-byte loads, and stores 2 x 8 bytes of luma and 4 + 4 bytes of
// chroma, so every source byte is read once and every destination byte written once; consecutive lanes
// cover consecutive segments (coalesced).  Frames of a batch are blockIdx.y.  Output layout = ms_yuv_buf_init
// (msvideo.c:85-99).
#include "common.hpp"

namespace {

struct PcArgs {
	const uint8_t *src;
	uint8_t *dst;
	size_t src_pitch, dst_pitch; // bytes between frames
	int w, h, h2, fmt, flip, bpp, groups, npairs, fast;
};

__device__ __forceinline__ int avgb(int a, int b) { return (a + b + 1) >> 1; }

template <int N>
struct Seg { // N dwords of one row segment
	uint32_t r[N];
	__device__ __forceinline__ int byte(int i) const { return (int)((r[i >> 2] >> (8 * (i & 3))) & 0xffu); }
};

template <int N>
__device__ __forceinline__ void load_seg(Seg<N> &s, const uint8_t *row, int off, int row_bytes, bool fast) {
	if (fast) {
		if (N == 4) {
			const uint4 v = *reinterpret_cast<const uint4 *>(row + off);
			s.r[0] = v.x, s.r[1] = v.y, s.r[2] = v.z, s.r[3] = v.w;
		} else if (N == 6) {
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const uint2 v = *reinterpret_cast<const uint2 *>(row + off + 8 * k);
				s.r[2 * k] = v.x, s.r[2 * k + 1] = v.y;
			}
		} else {
#pragma unroll
			for (int k = 0; k < N / 4; ++k) {
				const uint4 v = *reinterpret_cast<const uint4 *>(row + off + 16 * k);
				s.r[4 * k] = v.x, s.r[4 * k + 1] = v.y, s.r[4 * k + 2] = v.z, s.r[4 * k + 3] = v.w;
			}
		}
	} else { // ragged / unaligned: byte loads, clamped inside the row
#pragma unroll
		for (int k = 0; k < N; ++k) {
			uint32_t v = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) v |= (uint32_t)row[min(off + 4 * k + b, row_bytes - 1)] << (8 * b);
			s.r[k] = v;
		}
	}
}

// BPP bytes per pixel; RO/GO/BO byte offsets inside a pixel; JPEG = full-range coefficient set
template <int BPP, int RO, int GO, int BO, bool JPEG>
__device__ __forceinline__ void conv_rgb(const Seg<2 * BPP> &a, const Seg<2 * BPP> &b, int y0[8], int y1[8], int u[4],
                                         int v[4]) {
#pragma unroll
	for (int x = 0; x < 8; ++x) {
		const int r0 = a.byte(BPP * x + RO), g0 = a.byte(BPP * x + GO), b0 = a.byte(BPP * x + BO);
		const int r1 = b.byte(BPP * x + RO), g1 = b.byte(BPP * x + GO), b1 = b.byte(BPP * x + BO);
		y0[x] = JPEG ? (77 * r0 + 150 * g0 + 29 * b0 + 128) >> 8 : (66 * r0 + 129 * g0 + 25 * b0 + 0x1080) >> 8;
		y1[x] = JPEG ? (77 * r1 + 150 * g1 + 29 * b1 + 128) >> 8 : (66 * r1 + 129 * g1 + 25 * b1 + 0x1080) >> 8;
	}
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const int x = 2 * c;
		const int ab = avgb(avgb(a.byte(BPP * x + BO), b.byte(BPP * x + BO)), avgb(a.byte(BPP * (x + 1) + BO), b.byte(BPP * (x + 1) + BO)));
		const int ag = avgb(avgb(a.byte(BPP * x + GO), b.byte(BPP * x + GO)), avgb(a.byte(BPP * (x + 1) + GO), b.byte(BPP * (x + 1) + GO)));
		const int ar = avgb(avgb(a.byte(BPP * x + RO), b.byte(BPP * x + RO)), avgb(a.byte(BPP * (x + 1) + RO), b.byte(BPP * (x + 1) + RO)));
		u[c] = JPEG ? (127 * ab - 84 * ag - 43 * ar + 0x8080) >> 8 : (112 * ab - 74 * ag - 38 * ar + 0x8080) >> 8;
		v[c] = JPEG ? (127 * ar - 107 * ag - 20 * ab + 0x8080) >> 8 : (112 * ar - 94 * ag - 18 * ab + 0x8080) >> 8;
	}
}

// 4:2:2 packed: YO0/YO1 luma byte offsets in the 4-byte macropixel, UO/VO chroma offsets
template <int YO0, int YO1, int UO, int VO>
__device__ __forceinline__ void conv_422(const Seg<4> &a, const Seg<4> &b, int y0[8], int y1[8], int u[4], int v[4]) {
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		y0[2 * c] = a.byte(4 * c + YO0), y0[2 * c + 1] = a.byte(4 * c + YO1);
		y1[2 * c] = b.byte(4 * c + YO0), y1[2 * c + 1] = b.byte(4 * c + YO1);
		u[c] = (a.byte(4 * c + UO) + b.byte(4 * c + UO) + 1) >> 1;
		v[c] = (a.byte(4 * c + VO) + b.byte(4 * c + VO) + 1) >> 1;
	}
}

__device__ __forceinline__ uint32_t pack4(const int *p) {
	return (uint32_t)(p[0] & 0xff) | ((uint32_t)(p[1] & 0xff) << 8) | ((uint32_t)(p[2] & 0xff) << 16) | ((uint32_t)(p[3] & 0xff) << 24);
}

template <int FMT>
__global__ __launch_bounds__(256) void pixconv_kernel(PcArgs a) {
	constexpr int BPP = (FMT == MI_PIX_YUY2 || FMT == MI_PIX_UYVY) ? 2 : (FMT == MI_PIX_BGRA32 ? 4 : 3);
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= a.groups * a.npairs) return;
	const int pair = idx / a.groups, g = idx - pair * a.groups;
	const int y = 2 * pair;
	const bool single = y + 1 >= a.h; // odd height: the last row pairs with itself
	const uint8_t *frame = a.src + (size_t)blockIdx.y * a.src_pitch;
	const int row_bytes = a.w * BPP;
	const int r0i = a.flip ? a.h - 1 - y : y;
	const int r1i = single ? r0i : (a.flip ? r0i - 1 : r0i + 1);
	const uint8_t *row0 = frame + (size_t)r0i * row_bytes, *row1 = frame + (size_t)r1i * row_bytes;
	const int px = 8 * g, npx = min(8, a.w - px);
	const bool fast = a.fast && npx == 8;

	Seg<2 * BPP> s0, s1;
	load_seg<2 * BPP>(s0, row0, px * BPP, row_bytes, fast);
	load_seg<2 * BPP>(s1, row1, px * BPP, row_bytes, fast);
	int y0[8], y1[8], u[4], v[4];
	if (FMT == MI_PIX_YUY2) conv_422<0, 2, 1, 3>(reinterpret_cast<const Seg<4> &>(s0), reinterpret_cast<const Seg<4> &>(s1), y0, y1, u, v);
	else if (FMT == MI_PIX_UYVY) conv_422<1, 3, 0, 2>(reinterpret_cast<const Seg<4> &>(s0), reinterpret_cast<const Seg<4> &>(s1), y0, y1, u, v);
	else if (FMT == MI_PIX_BGR24) conv_rgb<3, 2, 1, 0, true>(reinterpret_cast<const Seg<6> &>(s0), reinterpret_cast<const Seg<6> &>(s1), y0, y1, u, v);
	else if (FMT == MI_PIX_RGB24_RAW) conv_rgb<3, 0, 1, 2, false>(reinterpret_cast<const Seg<6> &>(s0), reinterpret_cast<const Seg<6> &>(s1), y0, y1, u, v);
	else conv_rgb<4, 2, 1, 0, false>(reinterpret_cast<const Seg<8> &>(s0), reinterpret_cast<const Seg<8> &>(s1), y0, y1, u, v);

	uint8_t *dframe = a.dst + (size_t)blockIdx.y * a.dst_pitch;
	uint8_t *dy0 = dframe + (size_t)y * a.w + px;
	const int cw = a.w / 2;
	uint8_t *du = dframe + (size_t)a.w * a.h2 + (size_t)pair * cw + 4 * g;
	uint8_t *dv = du + (size_t)cw * (a.h2 / 2);
	if (fast) {
		*reinterpret_cast<uint2 *>(dy0) = make_uint2(pack4(y0), pack4(y0 + 4));
		if (!single) *reinterpret_cast<uint2 *>(dy0 + a.w) = make_uint2(pack4(y1), pack4(y1 + 4));
		*reinterpret_cast<uint32_t *>(du) = pack4(u);
		*reinterpret_cast<uint32_t *>(dv) = pack4(v);
	} else {
		for (int x = 0; x < npx; ++x) {
			dy0[x] = (uint8_t)y0[x];
			if (!single) dy0[a.w + x] = (uint8_t)y1[x];
		}
		for (int c = 0; 2 * c < npx; ++c) du[c] = (uint8_t)u[c], dv[c] = (uint8_t)v[c];
	}
}

} // namespace

struct mi_pixconv {
	mi_ctx *ctx = nullptr;
	PcArgs a;
	size_t src_bytes = 0, dst_bytes = 0;
};

extern "C" {

int mi_pixconv_create(mi_ctx *ctx, int w, int h, int src_fmt, int flip_vertical, mi_pixconv **out) {
	MI_CHECK_ARG(ctx && out && w >= 2 && h >= 1);
	*out = nullptr;
	if (src_fmt < MI_PIX_YUY2 || src_fmt > MI_PIX_BGRA32) {
		mi::set_error("pixel format %d has no conversion to I420 (yuv_scale msvideo.c:574-576 warns and fails too)", src_fmt);
		return MI_ENOTSUP;
	}
	if ((w & 1) || w >= 32768 || h >= 32768) {
		mi::set_error("picture %dx%d: width must be even and both dimensions < 32768", w, h);
		return MI_ENOTSUP;
	}
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_pixconv *p = new mi_pixconv();
	p->ctx = ctx;
	PcArgs &a = p->a;
	memset(&a, 0, sizeof(a));
	a.w = w, a.h = h, a.h2 = h + (h & 1), a.fmt = src_fmt, a.flip = flip_vertical ? 1 : 0;
	a.bpp = (src_fmt == MI_PIX_YUY2 || src_fmt == MI_PIX_UYVY) ? 2 : (src_fmt == MI_PIX_BGRA32 ? 4 : 3);
	a.groups = mi::ceil_div(w, 8);
	a.npairs = a.h2 / 2;
	p->src_bytes = (size_t)w * h * a.bpp;
	p->dst_bytes = (size_t)w * a.h2 + 2 * (size_t)(w / 2) * (a.h2 / 2);
	*out = p;
	return MI_OK;
}

void mi_pixconv_destroy(mi_pixconv *p) { delete p; }
size_t mi_pixconv_src_bytes(const mi_pixconv *p) { return p ? p->src_bytes : 0; }
size_t mi_pixconv_dst_bytes(const mi_pixconv *p) { return p ? p->dst_bytes : 0; }

int mi_pixconv_process(mi_pixconv *p, int nframes, const uint8_t *d_src, size_t src_pitch, uint8_t *d_dst,
                       size_t dst_pitch) {
	MI_CHECK_ARG(p && d_src && d_dst && nframes > 0 && nframes <= 65535);
	MI_CHECK_ARG(src_pitch >= p->src_bytes && dst_pitch >= p->dst_bytes);
	if (p->ctx->activate() != MI_OK) return MI_ENODEV;
	PcArgs a = p->a;
	a.src = d_src;
	a.dst = d_dst;
	a.src_pitch = src_pitch;
	a.dst_pitch = dst_pitch;
	// aligned wide accesses need whole 8-pixel groups per row and 16-byte aligned frames
	a.fast = ((a.w & 7) == 0) && (((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst) | src_pitch | dst_pitch) & 15) == 0) &&
	         ((((size_t)a.w * a.h2) & 15) == 0) && ((((size_t)(a.w / 2) * (a.h2 / 2)) & 3) == 0);
	const dim3 grid((unsigned)mi::ceil_div(a.groups * a.npairs, 256), (unsigned)nframes);
	switch (a.fmt) {
		case MI_PIX_YUY2: hipLaunchKernelGGL(pixconv_kernel<MI_PIX_YUY2>, grid, dim3(256), 0, p->ctx->stream, a); break;
		case MI_PIX_UYVY: hipLaunchKernelGGL(pixconv_kernel<MI_PIX_UYVY>, grid, dim3(256), 0, p->ctx->stream, a); break;
		case MI_PIX_BGR24: hipLaunchKernelGGL(pixconv_kernel<MI_PIX_BGR24>, grid, dim3(256), 0, p->ctx->stream, a); break;
		case MI_PIX_RGB24_RAW: hipLaunchKernelGGL(pixconv_kernel<MI_PIX_RGB24_RAW>, grid, dim3(256), 0, p->ctx->stream, a); break;
		default: hipLaunchKernelGGL(pixconv_kernel<MI_PIX_BGRA32>, grid, dim3(256), 0, p->ctx->stream, a); break;
	}
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_pixconv_process_host(mi_pixconv *p, int nframes, const uint8_t *h_src, size_t src_pitch, uint8_t *h_dst,
                            size_t dst_pitch) {
	MI_CHECK_ARG(p && h_src && h_dst && nframes > 0);
	mi_ctx *c = p->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	void *din, *dout;
	int rc;
	if ((rc = c->ensure_scratch(0, src_pitch * nframes + 32, &din)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, dst_pitch * nframes + 32, &dout)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(din, h_src, src_pitch * nframes, hipMemcpyHostToDevice, c->stream));
	rc = mi_pixconv_process(p, nframes, (const uint8_t *)din, src_pitch, (uint8_t *)dout, dst_pitch);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_dst, dout, dst_pitch * nframes, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"
