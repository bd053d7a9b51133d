// scaler.hip -- batched I420 bilinear scaler (+ fused BT.601 -> RGB24) for gfx950.
//
// Replaces MSScalerDesc.context_process (include/mediastreamer2/msvideo.h:473-478)
// as used by MSSizeConv / MSPixConv (src/videofilters/sizeconv.c:133-181,
// pixconv.c:66-93 -> ms_scaler_process src/voip/msvideo.c:711-713).  The
// reference delegates to libyuv I420Scale(kFilterBilinear) (msvideo.c:548) or
// swscale (un-vendored, unpinned); the arithmetic here is the libyuv portable
// C definition in 16.16 fixed point (rows: 8-bit fraction, +128; columns:
// 16-bit fraction, +0x8000), and the colour stage uses the in-tree Q13 BT.601
// limited-range constants of src/voip/scaler_arm.S:54-63.  All integer.
//
// Mapping: one workgroup produces BAND luma row-pairs of one frame.  For each
// output luma row the two source rows are loaded with 16-byte coalesced loads,
// blended vertically into an LDS row, then every lane filters 4 output pixels
// horizontally from LDS; the chroma row of the pair is produced the same way
// from the half-resolution planes.  RGB24 leaves as 12 bytes per lane per row
// (three dword stores, 768 contiguous bytes per wave).  The 3x3 colour matrix
// is 9 integer MACs per pixel on the VALU: the kernel is a byte stream bound
// by HBM (3.11 MB in + 2.76 MB out per 1080p->720p frame), so it is not
// reshaped for MFMA.
#include "common.hpp"

namespace {

constexpr int SC_THREADS = 256;

struct PlaneMap {
	int x0, dx, y0, dy; // 16.16
};

struct ScArgs {
	const uint8_t *src;
	uint8_t *dst;
	size_t src_pitch, dst_pitch; // bytes between frames
	int sw, sh, dw, dh;          // luma sizes
	int scw, sch, dcw, dch;      // chroma sizes
	int sh2, dh2;                // heights rounded up to even (ms_yuv_buf_init)
	PlaneMap ym, cm;
	int pairs_per_block, npairs; // output luma row pairs
	int rgb;
};

__device__ __forceinline__ int clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// vertical blend of source rows (yi, yi+1) at 8-bit fraction yf into an LDS row
__device__ __forceinline__ void blend_row(uint8_t *row, const uint8_t *plane, int stride, int w, int h, int y) {
	const int max_y = (h - 1) << 16;
	if (y > max_y) y = max_y;
	if (y < 0) y = 0;
	const int yi = y >> 16, yf = (y >> 8) & 255;
	const uint8_t *r0 = plane + (size_t)yi * stride;
	const uint8_t *r1 = (yi + 1 < h) ? r0 + stride : r0;
	const int tid = threadIdx.x;
	if (((stride & 15) == 0) && ((reinterpret_cast<uintptr_t>(plane) & 15) == 0)) {
		const int nq = (w + 15) >> 4; // stride multiple of 16 => whole groups are in-row
		for (int q = tid; q < nq; q += SC_THREADS) {
			const uint4 a = *reinterpret_cast<const uint4 *>(r0 + 16 * q);
			uint4 o = a;
			if (yf != 0) {
				const uint4 b = *reinterpret_cast<const uint4 *>(r1 + 16 * q);
				const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
				unsigned ov[4];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					unsigned r = 0;
#pragma unroll
					for (int byte = 0; byte < 4; ++byte) {
						const int p0 = (av[k] >> (8 * byte)) & 255, p1 = (bv[k] >> (8 * byte)) & 255;
						r |= (unsigned)((p0 * (256 - yf) + p1 * yf + 128) >> 8) << (8 * byte);
					}
					ov[k] = r;
				}
				o = make_uint4(ov[0], ov[1], ov[2], ov[3]);
			}
			*reinterpret_cast<uint4 *>(row + 16 * q) = o;
		}
	} else {
		for (int i = tid; i < w; i += SC_THREADS) {
			const int p0 = r0[i], p1 = r1[i];
			row[i] = (uint8_t)(yf == 0 ? p0 : ((p0 * (256 - yf) + p1 * yf + 128) >> 8));
		}
	}
}

__device__ __forceinline__ int filter_col(const uint8_t *row, int sw, long long x) {
	int xi = (int)(x >> 16), f = (int)(x & 0xffff);
	if (xi < 0) {
		xi = 0;
		f = 0;
	}
	const int xn = xi + 1 < sw ? xi + 1 : sw - 1;
	const int a = row[xi], b = row[xn];
	return a + ((f * (b - a) + 0x8000) >> 16);
}

template <bool RGB>
__global__ __launch_bounds__(SC_THREADS) void scaler_kernel(ScArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lw = (a.sw + 31) & ~15, cwp = (a.scw + 31) & ~15;
	uint8_t *rowY0 = reinterpret_cast<uint8_t *>(smem);
	uint8_t *rowY1 = rowY0 + lw;
	uint8_t *rowU = rowY1 + lw;
	uint8_t *rowV = rowU + cwp;
	uint8_t *outU = rowV + cwp;          // filtered chroma row (dcw bytes)
	uint8_t *outV = outU + ((a.dcw + 15) & ~15);

	const int frame = blockIdx.y;
	const uint8_t *sY = a.src + (size_t)frame * a.src_pitch;
	const uint8_t *sU = sY + (size_t)a.sw * a.sh2;
	const uint8_t *sV = sU + (size_t)a.scw * a.sch;
	uint8_t *d = a.dst + (size_t)frame * a.dst_pitch;
	const int tid = threadIdx.x;

	const int p_begin = blockIdx.x * a.pairs_per_block;
	const int p_end = min(p_begin + a.pairs_per_block, a.npairs);
	for (int pr = p_begin; pr < p_end; ++pr) {
		const int oy0 = 2 * pr, oy1 = 2 * pr + 1;
		const bool has1 = oy1 < a.dh;
		__syncthreads(); // previous pair's LDS rows are free
		blend_row(rowY0, sY, a.sw, a.sw, a.sh, a.ym.y0 + oy0 * a.ym.dy);
		if (has1) blend_row(rowY1, sY, a.sw, a.sw, a.sh, a.ym.y0 + oy1 * a.ym.dy);
		if (pr < a.dch) {
			blend_row(rowU, sU, a.scw, a.scw, a.sch, a.cm.y0 + pr * a.cm.dy);
			blend_row(rowV, sV, a.scw, a.scw, a.sch, a.cm.y0 + pr * a.cm.dy);
		}
		__syncthreads();
		if (pr < a.dch) {
			for (int x = tid; x < a.dcw; x += SC_THREADS) {
				const long long fx = (long long)a.cm.x0 + (long long)x * a.cm.dx;
				outU[x] = (uint8_t)filter_col(rowU, a.scw, fx);
				outV[x] = (uint8_t)filter_col(rowV, a.scw, fx);
			}
		}
		if (!RGB) {
			uint8_t *dY = d, *dU = d + (size_t)a.dw * a.dh2, *dV = dU + (size_t)a.dcw * a.dch;
			for (int x = tid; x < a.dw; x += SC_THREADS) {
				const long long fx = (long long)a.ym.x0 + (long long)x * a.ym.dx;
				dY[(size_t)oy0 * a.dw + x] = (uint8_t)filter_col(rowY0, a.sw, fx);
				if (has1) dY[(size_t)oy1 * a.dw + x] = (uint8_t)filter_col(rowY1, a.sw, fx);
			}
			__syncthreads();
			if (pr < a.dch) {
				for (int x = tid; x < a.dcw; x += SC_THREADS) {
					dU[(size_t)pr * a.dcw + x] = outU[x];
					dV[(size_t)pr * a.dcw + x] = outV[x];
				}
			}
		} else {
			__syncthreads();
			// 4 pixels (12 bytes) per lane per row
			const int groups = (a.dw + 3) >> 2;
			const size_t pitch = (size_t)a.dw * 3;
			const bool fast = ((a.dw & 3) == 0) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0);
			for (int g = tid; g < groups; g += SC_THREADS) {
#pragma unroll
				for (int rsel = 0; rsel < 2; ++rsel) {
					if (rsel == 1 && !has1) break;
					const uint8_t *rowY = rsel ? rowY1 : rowY0;
					const int oy = rsel ? oy1 : oy0;
					uint8_t px[12];
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						const int x = 4 * g + k;
						const int xc = min(x, a.dw - 1);
						const long long fx = (long long)a.ym.x0 + (long long)xc * a.ym.dx;
						const int cx = min(xc >> 1, a.dcw - 1);
						const int c = filter_col(rowY, a.sw, fx) - 16;
						const int dd = (int)outU[cx] - 128, ee = (int)outV[cx] - 128;
						const int yy = 9535 * c + 4096;
						px[3 * k + 0] = (uint8_t)clamp8((yy + 13074 * ee) >> 13);
						px[3 * k + 1] = (uint8_t)clamp8((yy - 3203 * dd - 6660 * ee) >> 13);
						px[3 * k + 2] = (uint8_t)clamp8((yy + 16531 * dd) >> 13);
					}
					uint8_t *o = d + (size_t)oy * pitch + (size_t)12 * g;
					if (fast) {
						uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
						o32[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
						o32[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
						o32[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24);
					} else {
						for (int k = 0; k < 12; ++k)
							if (4 * g + k / 3 < a.dw) o[k] = px[k];
					}
				}
			}
		}
	}
}

void plane_map(int src, int dst, int *x, int *dx) { // libyuv ScaleSlope, bilinear
	*x = 0;
	*dx = 0;
	if (dst <= src) {
		*dx = (int)(((int64_t)src << 16) / dst);
		*x = (*dx >> 1) - 32768;
	} else if (src > 1 && dst > 1) {
		*dx = (int)((((int64_t)src << 16) - 0x00010001) / (dst - 1));
		*x = 0;
	}
}

} // namespace

struct mi_scaler {
	mi_ctx *ctx = nullptr;
	ScArgs a;
	size_t src_bytes = 0, dst_bytes = 0, lds = 0;
};

extern "C" {

int mi_scaler_create(mi_ctx *ctx, int sw, int sh, int dw, int dh, int dst_fmt, mi_scaler **out) {
	MI_CHECK_ARG(ctx && out && sw >= 2 && sh >= 2 && dw >= 2 && dh >= 2);
	MI_CHECK_ARG(dst_fmt == MI_PIX_I420 || dst_fmt == MI_PIX_RGB24);
	*out = nullptr;
	if (sw >= 32768 || sh >= 32768 || dw >= 32768 || dh >= 32768) {
		mi::set_error("picture dimension >= 32768 not supported");
		return MI_ENOTSUP;
	}
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_scaler *s = new mi_scaler();
	s->ctx = ctx;
	ScArgs &a = s->a;
	memset(&a, 0, sizeof(a));
	a.sw = sw, a.sh = sh, a.dw = dw, a.dh = dh;
	a.sh2 = sh + (sh & 1), a.dh2 = dh + (dh & 1); // msvideo.c:87
	a.scw = sw / 2, a.sch = a.sh2 / 2, a.dcw = dw / 2, a.dch = a.dh2 / 2;
	plane_map(sw, dw, &a.ym.x0, &a.ym.dx);
	plane_map(sh, dh, &a.ym.y0, &a.ym.dy);
	plane_map(a.scw, a.dcw, &a.cm.x0, &a.cm.dx);
	plane_map(a.sch, a.dch, &a.cm.y0, &a.cm.dy);
	a.rgb = dst_fmt == MI_PIX_RGB24;
	a.npairs = a.dh2 / 2;
	a.pairs_per_block = 4;
	s->src_bytes = (size_t)sw * a.sh2 + 2 * (size_t)a.scw * a.sch;
	s->dst_bytes = a.rgb ? (size_t)dw * dh * 3 : (size_t)dw * a.dh2 + 2 * (size_t)a.dcw * a.dch;
	const int lw = (sw + 31) & ~15, cwp = (a.scw + 31) & ~15;
	s->lds = (size_t)2 * lw + 2 * cwp + 2 * (size_t)((a.dcw + 15) & ~15);
	if (s->lds > 64 * 1024) {
		mi::set_error("source width %d too large for the scaler kernel's LDS rows", sw);
		delete s;
		return MI_ENOTSUP;
	}
	*out = s;
	return MI_OK;
}

void mi_scaler_destroy(mi_scaler *s) { delete s; }
size_t mi_scaler_src_bytes(const mi_scaler *s) { return s ? s->src_bytes : 0; }
size_t mi_scaler_dst_bytes(const mi_scaler *s) { return s ? s->dst_bytes : 0; }

int mi_scaler_process(mi_scaler *s, int nframes, const uint8_t *d_src, size_t src_pitch, uint8_t *d_dst,
                      size_t dst_pitch) {
	MI_CHECK_ARG(s && d_src && d_dst && nframes > 0);
	MI_CHECK_ARG(src_pitch >= s->src_bytes && dst_pitch >= s->dst_bytes);
	MI_CHECK_ARG(nframes <= 65535);
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	ScArgs a = s->a;
	a.src = d_src;
	a.dst = d_dst;
	a.src_pitch = src_pitch;
	a.dst_pitch = dst_pitch;
	const dim3 grid((unsigned)mi::ceil_div(a.npairs, a.pairs_per_block), (unsigned)nframes);
	if (a.rgb) hipLaunchKernelGGL(scaler_kernel<true>, grid, dim3(SC_THREADS), s->lds, s->ctx->stream, a);
	else hipLaunchKernelGGL(scaler_kernel<false>, grid, dim3(SC_THREADS), s->lds, s->ctx->stream, a);
	MI_LAUNCH_CHECK();
	return MI_OK;
}

int mi_scaler_process_host(mi_scaler *s, int nframes, const uint8_t *h_src, size_t src_pitch, uint8_t *h_dst,
                           size_t dst_pitch) {
	MI_CHECK_ARG(s && h_src && h_dst && nframes > 0);
	mi_ctx *c = s->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	void *din, *dout;
	int rc;
	// +32 bytes: the 16-byte row loads may touch the pad after the last plane row
	if ((rc = c->ensure_scratch(0, src_pitch * nframes + 32, &din)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, dst_pitch * nframes, &dout)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(din, h_src, src_pitch * nframes, hipMemcpyHostToDevice, c->stream));
	rc = mi_scaler_process(s, nframes, (const uint8_t *)din, src_pitch, (uint8_t *)dout, dst_pitch);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_dst, dout, dst_pitch * nframes, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"
