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
#include <vector>

namespace {

constexpr int SC_THREADS = 256; // generic (any-alignment) kernel

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

// one vertical-blend job: source rows (yi, yi+1) of a plane at 8-bit fraction yf -> an LDS row
struct RowJob {
	const uint8_t *r0, *r1;
	uint8_t *dst;
	int w, yf, nq; // nq = 16-byte groups
};

__device__ __forceinline__ RowJob make_job(uint8_t *dst, const uint8_t *plane, int stride, int w, int h, int y) {
	const int max_y = (h - 1) << 16;
	if (y > max_y) y = max_y;
	if (y < 0) y = 0;
	const int yi = y >> 16;
	RowJob j;
	j.r0 = plane + (size_t)yi * stride;
	j.r1 = (yi + 1 < h) ? j.r0 + stride : j.r0;
	j.dst = dst;
	j.w = w;
	j.yf = (y >> 8) & 255;
	j.nq = (w + 15) >> 4;
	return j;
}

// InterpolateRow_C of libyuv on 16 bytes: (p0*(256-yf) + p1*yf + 128) >> 8 per byte.  Two bytes per 32-bit
// operation: the even bytes of a dword sit in its two 16-bit halves (x & 0x00ff00ff), the odd ones after a
// shift; each half's result is at most 255*256+128 < 65536, so nothing carries between the halves, and the
// 24-bit multiplier (v_mad_u32_u24) takes 0x00ff00ff whole.
__device__ __forceinline__ uint4 blend16(uint4 a, uint4 b, int yf) {
	if (yf == 0) return a;
	const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
	const unsigned w1 = (unsigned)yf, w0 = 256u - w1;
	unsigned ov[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const unsigned ae = av[k] & 0x00ff00ffu, ao = (av[k] >> 8) & 0x00ff00ffu;
		const unsigned be = bv[k] & 0x00ff00ffu, bo = (bv[k] >> 8) & 0x00ff00ffu;
		const unsigned te = __umul24(be, w1) + (__umul24(ae, w0) + 0x00800080u);
		const unsigned to = __umul24(bo, w1) + (__umul24(ao, w0) + 0x00800080u);
		ov[k] = ((te >> 8) & 0x00ff00ffu) | (to & 0xff00ff00u);
	}
	return make_uint4(ov[0], ov[1], ov[2], ov[3]);
}

// All row jobs of one output row pair are flattened into one work list so that every lane has
// its (up to 2) items' loads in flight before the first blend: one memory round trip per pair.
__device__ __forceinline__ void run_jobs(const RowJob *jobs, int njobs, bool vec) {
	const int tid = threadIdx.x;
	if (vec) {
		int total = 0;
		for (int j = 0; j < njobs; ++j) total += jobs[j].nq;
		// two items per lane, both pairs of loads issued before the first blend
		uint4 a0 = make_uint4(0, 0, 0, 0), b0 = a0, a1 = a0, b1 = a0;
		uint8_t *d0 = nullptr, *d1 = nullptr;
		int yf0 = 0, yf1 = 0;
		{
			int it = tid;
			if (it < total) {
				int j = 0, q = it;
				while (q >= jobs[j].nq) {
					q -= jobs[j].nq;
					++j;
				}
				yf0 = jobs[j].yf;
				d0 = jobs[j].dst + 16 * q;
				a0 = *reinterpret_cast<const uint4 *>(jobs[j].r0 + 16 * q);
				b0 = yf0 ? *reinterpret_cast<const uint4 *>(jobs[j].r1 + 16 * q) : a0;
			}
			it += SC_THREADS;
			if (it < total) {
				int j = 0, q = it;
				while (q >= jobs[j].nq) {
					q -= jobs[j].nq;
					++j;
				}
				yf1 = jobs[j].yf;
				d1 = jobs[j].dst + 16 * q;
				a1 = *reinterpret_cast<const uint4 *>(jobs[j].r0 + 16 * q);
				b1 = yf1 ? *reinterpret_cast<const uint4 *>(jobs[j].r1 + 16 * q) : a1;
			}
		}
		if (d0) *reinterpret_cast<uint4 *>(d0) = blend16(a0, b0, yf0);
		if (d1) *reinterpret_cast<uint4 *>(d1) = blend16(a1, b1, yf1);
		// pictures wider than 2*SC_THREADS*16/… fall through to the generic loop for the remainder
		for (int it = tid + 2 * SC_THREADS; it < total; it += SC_THREADS) {
			int j = 0, q = it;
			while (q >= jobs[j].nq) {
				q -= jobs[j].nq;
				++j;
			}
			const uint4 x = *reinterpret_cast<const uint4 *>(jobs[j].r0 + 16 * q);
			const uint4 y = jobs[j].yf ? *reinterpret_cast<const uint4 *>(jobs[j].r1 + 16 * q) : x;
			*reinterpret_cast<uint4 *>(jobs[j].dst + 16 * q) = blend16(x, y, jobs[j].yf);
		}
	} else {
		for (int j = 0; j < njobs; ++j) {
			const RowJob &J = jobs[j];
			for (int i = tid; i < J.w; i += SC_THREADS) {
				const int p0 = J.r0[i], p1 = J.r1[i];
				J.dst[i] = (uint8_t)(J.yf == 0 ? p0 : ((p0 * (256 - J.yf) + p1 * J.yf + 128) >> 8));
			}
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

// copy `n` bytes LDS -> global with 16-byte stores when both sides allow it
__device__ __forceinline__ void store_row(uint8_t *dst, const uint8_t *lds, int n) {
	const int tid = threadIdx.x;
	if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (n & 15) == 0) {
		for (int q = tid; q < (n >> 4); q += SC_THREADS)
			*reinterpret_cast<uint4 *>(dst + 16 * q) = *reinterpret_cast<const uint4 *>(lds + 16 * q);
	} else {
		for (int i = tid; i < n; i += SC_THREADS) dst[i] = lds[i];
	}
}

template <bool RGB>
__global__ __launch_bounds__(SC_THREADS) void scaler_kernel(ScArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	__shared__ RowJob jobs[4];
	const int lw = (a.sw + 31) & ~15, cwp = (a.scw + 31) & ~15;
	const int ow = RGB ? ((a.dw * 3 + 15) & ~15) : ((a.dw + 15) & ~15); // staged output row
	uint8_t *rowY0 = reinterpret_cast<uint8_t *>(smem);
	uint8_t *rowY1 = rowY0 + lw;
	uint8_t *rowU = rowY1 + lw;
	uint8_t *rowV = rowU + cwp;
	uint8_t *outU = rowV + cwp; // filtered chroma rows (dcw bytes)
	uint8_t *outV = outU + ((a.dcw + 15) & ~15);
	uint8_t *out0 = outV + ((a.dcw + 15) & ~15); // staged output rows
	uint8_t *out1 = out0 + ow;

	const int frame = blockIdx.y;
	const uint8_t *sY = a.src + (size_t)frame * a.src_pitch;
	const uint8_t *sU = sY + (size_t)a.sw * a.sh2;
	const uint8_t *sV = sU + (size_t)a.scw * a.sch;
	uint8_t *d = a.dst + (size_t)frame * a.dst_pitch;
	const int tid = threadIdx.x;
	const bool vec = ((a.sw & 15) == 0) && ((a.scw & 15) == 0) && ((reinterpret_cast<uintptr_t>(sY) & 15) == 0) &&
	                 ((reinterpret_cast<uintptr_t>(sU) & 15) == 0) && ((reinterpret_cast<uintptr_t>(sV) & 15) == 0);

	const int p_begin = blockIdx.x * a.pairs_per_block;
	const int p_end = min(p_begin + a.pairs_per_block, a.npairs);
	for (int pr = p_begin; pr < p_end; ++pr) {
		const int oy0 = 2 * pr, oy1 = 2 * pr + 1;
		const bool has1 = oy1 < a.dh;
		// the (uniform) job descriptors live in LDS: indexing a per-lane array would go to scratch
		int nj = 1 + (has1 ? 1 : 0) + (pr < a.dch ? 2 : 0);
		__syncthreads(); // previous pair's LDS rows and descriptors are free
		if (tid == 0) {
			int n = 0;
			jobs[n++] = make_job(rowY0, sY, a.sw, a.sw, a.sh, a.ym.y0 + oy0 * a.ym.dy);
			if (has1) jobs[n++] = make_job(rowY1, sY, a.sw, a.sw, a.sh, a.ym.y0 + oy1 * a.ym.dy);
			if (pr < a.dch) {
				jobs[n++] = make_job(rowU, sU, a.scw, a.scw, a.sch, a.cm.y0 + pr * a.cm.dy);
				jobs[n++] = make_job(rowV, sV, a.scw, a.scw, a.sch, a.cm.y0 + pr * a.cm.dy);
			}
		}
		__syncthreads();
		run_jobs(jobs, nj, vec);
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
				out0[x] = (uint8_t)filter_col(rowY0, a.sw, fx);
				if (has1) out1[x] = (uint8_t)filter_col(rowY1, a.sw, fx);
			}
			__syncthreads();
			store_row(dY + (size_t)oy0 * a.dw, out0, a.dw);
			if (has1) store_row(dY + (size_t)oy1 * a.dw, out1, a.dw);
			if (pr < a.dch) {
				store_row(dU + (size_t)pr * a.dcw, outU, a.dcw);
				store_row(dV + (size_t)pr * a.dcw, outV, a.dcw);
			}
		} else {
			__syncthreads();
			// 4 pixels (12 bytes) per lane per row, staged so the row leaves as 16-byte stores
			const int groups = (a.dw + 3) >> 2;
			for (int g = tid; g < groups; g += SC_THREADS) {
#pragma unroll
				for (int rsel = 0; rsel < 2; ++rsel) {
					if (rsel == 1 && !has1) break;
					const uint8_t *rowY = rsel ? rowY1 : rowY0;
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
					uint32_t *o32 = reinterpret_cast<uint32_t *>((rsel ? out1 : out0) + 12 * g);
					o32[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
					o32[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
					o32[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24);
				}
			}
			__syncthreads();
			const size_t pitch = (size_t)a.dw * 3;
			store_row(d + (size_t)oy0 * pitch, out0, a.dw * 3);
			if (has1) store_row(d + (size_t)oy1 * pitch, out1, a.dw * 3);
		}
	}
}

// ------------------------------------------------------------------------------------------
// Fast path: ONE wavefront per 256-pixel-wide strip of a band of output row pairs.  No
// workgroup barriers: every wave has its own 16-byte source loads in flight (3-4 per lane per
// row pair, all issued before the first blend), blends them into its private LDS rows, then
// each lane filters and colour-converts its own 4x2 pixels (the two chroma samples it needs
// are its own), and the strip leaves as 16-byte stores.  With ~3.5 KB of LDS per wave a CU
// keeps 24-32 waves in different phases, which is what keeps HBM busy on a byte stream.
constexpr int WS_LUMA_Q = 28;   // 16-byte groups of luma a strip may need per source row (<= 448 B)
constexpr int WS_CHROMA_Q = 16; // same for a chroma row

struct WsRow { // per-wave LDS image of one blended source row segment
	int base;  // first source x held (multiple of 16)
};

template <bool RGB>
__global__ __launch_bounds__(64, 7) void scaler_wave_kernel(ScArgs a, int strips, int band_pairs, int bands, int total_items,
                                                             int xcd_chunk) {
	__shared__ __attribute__((aligned(16))) uint8_t rowY[2][WS_LUMA_Q * 16];
	__shared__ __attribute__((aligned(16))) uint8_t rowC[2][WS_CHROMA_Q * 16];
	__shared__ __attribute__((aligned(16))) uint8_t stage[2][256 * 3];
	const int lane = threadIdx.x;
	// Work item w = (frame, band, strip), strips fastest.  Workgroups are dealt round-robin to the 8 XCDs, each with
	// its own L2: item = (id % 8) * chunk + id / 8 gives every XCD one contiguous run of items, so the bands above
	// and below a band (which share a blended source row) and the neighbouring strips go through the same L2.
	int w = blockIdx.x;
	if (xcd_chunk > 0) w = (w & 7) * xcd_chunk + (w >> 3);
	if (w >= total_items) return;
	const int per_frame = strips * bands;
	const int frame = w / per_frame;
	const int rem = w - frame * per_frame;
	const int band = rem / strips, strip = rem - band * strips;
	const uint8_t *sY = a.src + (size_t)frame * a.src_pitch;
	const uint8_t *sU = sY + (size_t)a.sw * a.sh2;
	const uint8_t *sV = sU + (size_t)a.scw * a.sch;
	uint8_t *d = a.dst + (size_t)frame * a.dst_pitch;

	const int ox0 = strip * 256;                        // first output luma x of the strip
	const int npx = min(256, a.dw - ox0);               // luma pixels in this strip
	const int ocx0 = ox0 >> 1, ncx = (npx + 1) >> 1;    // chroma
	// source segments (16-byte aligned) the strip needs
	const long long fy0 = (long long)a.ym.x0 + (long long)ox0 * a.ym.dx;
	const long long fy1 = (long long)a.ym.x0 + (long long)(ox0 + npx - 1) * a.ym.dx;
	const int ybase = max(0, (int)(fy0 >> 16)) & ~15;
	const int yq = min((min(a.sw, (int)(fy1 >> 16) + 2) - ybase + 15) >> 4, WS_LUMA_Q);
	const long long fc0 = (long long)a.cm.x0 + (long long)ocx0 * a.cm.dx;
	const long long fc1 = (long long)a.cm.x0 + (long long)(ocx0 + ncx - 1) * a.cm.dx;
	const int cbase = max(0, (int)(fc0 >> 16)) & ~15;
	const int cq = min((min(a.scw, (int)(fc1 >> 16) + 2) - cbase + 15) >> 4, WS_CHROMA_Q);

	const int p_begin = band * band_pairs, p_end = min(p_begin + band_pairs, a.npairs);
	// vertical positions (uniform)
	auto vpos = [](int y, int h, int &yi, int &yf) {
		const int max_y = (h - 1) << 16;
		if (y > max_y) y = max_y;
		if (y < 0) y = 0;
		yi = y >> 16;
		yf = (y >> 8) & 255;
	};
	// ---- this lane's load items, fixed for the band: work list [Y0: yq][Y1: yq][U: cq][V: cq], two per lane
	const int total = 2 * yq + 2 * cq;
	int it_sel[2], it_col[2];
	uint8_t *it_dst[2];
#pragma unroll
	for (int n = 0; n < 2; ++n) {
		const int it = lane + 64 * n;
		it_sel[n] = -1, it_col[n] = 0, it_dst[n] = rowY[0];
		if (it < total) {
			if (it < 2 * yq) {
				const int sel = it >= yq, q = it - sel * yq;
				it_sel[n] = sel, it_col[n] = ybase + 16 * q, it_dst[n] = rowY[sel] + 16 * q;
			} else {
				const int t = it - 2 * yq, sel = t >= cq, q = t - sel * cq;
				it_sel[n] = 2 + sel, it_col[n] = cbase + 16 * q, it_dst[n] = rowC[sel] + 16 * q;
			}
		}
	}
	uint4 va[2], vb[2];
	int yfs[2];
	// all of a row pair's source loads (both rows of the vertical blend) go out together
	auto issue = [&](int pr) {
		const int oy0 = 2 * pr, oy1 = 2 * pr + 1;
		int yi0, yf0, yi1, yf1, ci, cf;
		vpos(a.ym.y0 + oy0 * a.ym.dy, a.sh, yi0, yf0);
		vpos(a.ym.y0 + (oy1 < a.dh ? oy1 : oy0) * a.ym.dy, a.sh, yi1, yf1);
		vpos(a.cm.y0 + (pr < a.dch ? pr : 0) * a.cm.dy, a.sch, ci, cf);
#pragma unroll
		for (int n = 0; n < 2; ++n) {
			va[n] = vb[n] = make_uint4(0, 0, 0, 0);
			yfs[n] = 0;
			const int sel = it_sel[n];
			if (sel >= 0) {
				const uint8_t *r0, *r1;
				if (sel < 2) {
					const int yi = sel ? yi1 : yi0;
					yfs[n] = sel ? yf1 : yf0;
					r0 = sY + (size_t)yi * a.sw + it_col[n];
					r1 = (yi + 1 < a.sh) ? r0 + a.sw : r0;
				} else {
					yfs[n] = cf;
					r0 = (sel == 3 ? sV : sU) + (size_t)ci * a.scw + it_col[n];
					r1 = (ci + 1 < a.sch) ? r0 + a.scw : r0;
				}
				va[n] = *reinterpret_cast<const uint4 *>(r0);
				vb[n] = yfs[n] ? *reinterpret_cast<const uint4 *>(r1) : va[n];
			}
		}
	};
	for (int pr = p_begin; pr < p_end; ++pr) {
		const int oy0 = 2 * pr, oy1 = 2 * pr + 1;
		const bool has1 = oy1 < a.dh;
		const bool hasc = pr < a.dch;
		// (keeping the next pair's loads in flight during the arithmetic was measured slower twice: the 16 extra
		// live VGPRs cost a wave per SIMD, and resident waves are what keeps HBM busy here)
		issue(pr);
		__syncthreads(); // the previous pair's rows are no longer read
#pragma unroll
		for (int n = 0; n < 2; ++n)
			if (it_sel[n] >= 0) *reinterpret_cast<uint4 *>(it_dst[n]) = blend16(va[n], vb[n], yfs[n]);
		__syncthreads();
		// ---- this lane's pixels: x = ox0 + 4*lane .. +3, chroma cx = ocx0 + 2*lane, +1.  Source positions in
		// 32-bit fixed point RELATIVE to the strip's LDS segment: (x - ox0) * dx < 2^25 on this path (launch check).
		const int x0l = 4 * lane;
		int cu[2] = {128, 128}, cv[2] = {128, 128};
		if (hasc) {
			const int cfx0 = (int)(fc0 - ((long long)cbase << 16));
			const int clast = a.dcw - 1 - ocx0, cmaxl = a.scw - 1 - cbase;
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const int fx = cfx0 + min(2 * lane + k, clast) * a.cm.dx;
				const int xi = fx >> 16, f = fx & 0xffff;
				const int xn = min(xi + 1, cmaxl);
				const int ua = rowC[0][xi], ub = rowC[0][xn];
				const int wa = rowC[1][xi], wb = rowC[1][xn];
				cu[k] = ua + ((f * (ub - ua) + 0x8000) >> 16);
				cv[k] = wa + ((f * (wb - wa) + 0x8000) >> 16);
			}
		}
		int yv[2][4];
		{
			const int yfx0 = (int)(fy0 - ((long long)ybase << 16));
			const int ylast = a.dw - 1 - ox0, ymaxl = a.sw - 1 - ybase;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int fx = yfx0 + min(x0l + k, ylast) * a.ym.dx;
				const int xi = fx >> 16, f = fx & 0xffff;
				const int xn = min(xi + 1, ymaxl);
#pragma unroll
				for (int r = 0; r < 2; ++r) {
					const int pa = rowY[r][xi], pb = rowY[r][xn];
					yv[r][k] = pa + ((f * (pb - pa) + 0x8000) >> 16);
				}
			}
		}
		if (RGB) {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				uint8_t px[12];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int c = yv[r][k] - 16, dd = cu[k >> 1] - 128, ee = cv[k >> 1] - 128;
					const int yy = 9535 * c + 4096;
					px[3 * k + 0] = (uint8_t)clamp8((yy + 13074 * ee) >> 13);
					px[3 * k + 1] = (uint8_t)clamp8((yy - 3203 * dd - 6660 * ee) >> 13);
					px[3 * k + 2] = (uint8_t)clamp8((yy + 16531 * dd) >> 13);
				}
				uint32_t *o32 = reinterpret_cast<uint32_t *>(stage[r] + 12 * lane);
				o32[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((uint32_t)px[3] << 24);
				o32[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((uint32_t)px[7] << 24);
				o32[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((uint32_t)px[11] << 24);
			}
			__syncthreads();
			const size_t pitch = (size_t)a.dw * 3;
			const int nbytes = npx * 3;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				if (r == 1 && !has1) break;
				uint8_t *o = d + (size_t)(r ? oy1 : oy0) * pitch + (size_t)ox0 * 3;
				if ((nbytes & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
					if (lane < (nbytes >> 4)) *reinterpret_cast<uint4 *>(o + 16 * lane) = *reinterpret_cast<const uint4 *>(stage[r] + 16 * lane);
				} else {
#pragma unroll 1
					for (int i = lane; i < nbytes; i += 64) o[i] = stage[r][i];
				}
			}
		} else {
			uint8_t *dY = d, *dU = d + (size_t)a.dw * a.dh2, *dV = dU + (size_t)a.dcw * a.dch;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				if (r == 1 && !has1) break;
				uint8_t *o = dY + (size_t)(r ? oy1 : oy0) * a.dw + ox0 + x0l;
				if (x0l + 4 <= npx && (reinterpret_cast<uintptr_t>(o) & 3) == 0) {
					*reinterpret_cast<uint32_t *>(o) = (uint32_t)yv[r][0] | (yv[r][1] << 8) | (yv[r][2] << 16) | ((uint32_t)yv[r][3] << 24);
				} else {
#pragma unroll 1
					for (int k = 0; k < 4; ++k)
						if (x0l + k < npx) o[k] = (uint8_t)yv[r][k];
				}
			}
			if (hasc) {
				for (int k = 0; k < 2; ++k)
					if (2 * lane + k < ncx && ocx0 + 2 * lane + k < a.dcw) {
						dU[(size_t)pr * a.dcw + ocx0 + 2 * lane + k] = (uint8_t)cu[k];
						dV[(size_t)pr * a.dcw + ocx0 + 2 * lane + k] = (uint8_t)cv[k];
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
	const int ow = a.rgb ? ((dw * 3 + 15) & ~15) + 16 : ((dw + 15) & ~15) + 16;
	s->lds = (size_t)2 * lw + 2 * cwp + 2 * (size_t)((a.dcw + 15) & ~15) + 2 * (size_t)ow;
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
	// fast wave-per-strip path: 16-byte aligned planes/rows and strips that fit the per-wave LDS rows
	const bool aligned = ((a.sw & 15) == 0) && ((a.scw & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_src) & 15) == 0) &&
	                     ((src_pitch & 15) == 0) && ((((size_t)a.sw * a.sh2) & 15) == 0) &&
	                     ((((size_t)a.scw * a.sch) & 15) == 0);
	const long long need_y = ((long long)255 * a.ym.dx >> 16) + 2 + 15 + 16;
	const long long need_c = ((long long)127 * a.cm.dx >> 16) + 2 + 15 + 16;
	if (aligned && a.ym.dx > 0 && a.cm.dx > 0 && need_y <= WS_LUMA_Q * 16 && need_c <= WS_CHROMA_Q * 16) {
		const char *bp = getenv("MSMI355X_SCALER_BAND");
		const int strips = mi::ceil_div(a.dw, 256), band_pairs = bp ? atoi(bp) : 4;
		const int bands = mi::ceil_div(a.npairs, band_pairs);
		const long long total_ll = (long long)strips * bands * nframes;
		if (total_ll > 0x7fffffffLL - 8) {
			mi::set_error("scaler batch too large for one launch");
			return MI_ENOTSUP;
		}
		const int total = (int)total_ll;
		static const bool xcd_map = getenv("MSMI355X_SCALER_NO_XCD_MAP") == nullptr;
		const int chunk = xcd_map ? mi::ceil_div(total, 8) : 0;
		const dim3 grid((unsigned)(xcd_map ? chunk * 8 : total));
		if (a.rgb) hipLaunchKernelGGL(scaler_wave_kernel<true>, grid, dim3(64), 0, s->ctx->stream, a, strips, band_pairs, bands, total, chunk);
		else hipLaunchKernelGGL(scaler_wave_kernel<false>, grid, dim3(64), 0, s->ctx->stream, a, strips, band_pairs, bands, total, chunk);
		MI_LAUNCH_CHECK();
		return MI_OK;
	}
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

int mi_scaler_process_planes_host(mi_scaler *s, const uint8_t *const src[3], const int src_strides[3],
                                  uint8_t *const dst[3], const int dst_strides[3]) {
	MI_CHECK_ARG(s && src && src_strides && dst && dst_strides && src[0] && src[1] && src[2] && dst[0]);
	const ScArgs &a = s->a;
	MI_CHECK_ARG(src_strides[0] >= a.sw && src_strides[1] >= a.scw && src_strides[2] >= a.scw);
	MI_CHECK_ARG(a.rgb ? dst_strides[0] >= 3 * a.dw : (dst[1] && dst[2] && dst_strides[0] >= a.dw && dst_strides[1] >= a.dcw && dst_strides[2] >= a.dcw));
	mi_ctx *c = s->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	void *din, *dout;
	int rc;
	if ((rc = c->ensure_scratch(0, s->src_bytes + 32, &din)) != MI_OK) return rc;
	if ((rc = c->ensure_scratch(1, s->dst_bytes + 32, &dout)) != MI_OK) return rc;
	uint8_t *d = (uint8_t *)din, *o = (uint8_t *)dout;
	// gather the three planes into the packed device layout (ms_yuv_buf_init order); rows beyond an odd
	// height are the pad row of that layout and are never sampled by the kernels
	const size_t ysz = (size_t)a.sw * a.sh2, csz = (size_t)a.scw * a.sch;
	const int ch_rows = (a.sh + 1) / 2;
	MI_HIP(hipMemcpy2DAsync(d, (size_t)a.sw, src[0], (size_t)src_strides[0], (size_t)a.sw, (size_t)a.sh, hipMemcpyHostToDevice, c->stream));
	MI_HIP(hipMemcpy2DAsync(d + ysz, (size_t)a.scw, src[1], (size_t)src_strides[1], (size_t)a.scw, (size_t)ch_rows, hipMemcpyHostToDevice, c->stream));
	MI_HIP(hipMemcpy2DAsync(d + ysz + csz, (size_t)a.scw, src[2], (size_t)src_strides[2], (size_t)a.scw, (size_t)ch_rows, hipMemcpyHostToDevice, c->stream));
	rc = mi_scaler_process(s, 1, d, s->src_bytes, o, s->dst_bytes);
	if (rc != MI_OK) return rc;
	if (a.rgb) {
		MI_HIP(hipMemcpy2DAsync(dst[0], (size_t)dst_strides[0], o, (size_t)a.dw * 3, (size_t)a.dw * 3, (size_t)a.dh, hipMemcpyDeviceToHost, c->stream));
	} else {
		const size_t dysz = (size_t)a.dw * a.dh2, dcsz = (size_t)a.dcw * a.dch;
		const int dch_rows = (a.dh + 1) / 2;
		MI_HIP(hipMemcpy2DAsync(dst[0], (size_t)dst_strides[0], o, (size_t)a.dw, (size_t)a.dw, (size_t)a.dh, hipMemcpyDeviceToHost, c->stream));
		MI_HIP(hipMemcpy2DAsync(dst[1], (size_t)dst_strides[1], o + dysz, (size_t)a.dcw, (size_t)a.dcw, (size_t)dch_rows, hipMemcpyDeviceToHost, c->stream));
		MI_HIP(hipMemcpy2DAsync(dst[2], (size_t)dst_strides[2], o + dysz + dcsz, (size_t)a.dcw, (size_t)a.dcw, (size_t)dch_rows, hipMemcpyDeviceToHost, c->stream));
	}
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"

// ---- the scaler fed from host buffers with the copies overlapped (BASELINE config 5: the frames cross PCIe both ways and
// the kernel is ~100x faster than either copy, so the copies ARE the path): batches of frames in a ring of pinned buffers,
// upload | kernel | download on three HIP streams ordered by events, up to `depth` batches in flight -- what session.hip
// does for the audio chain.
struct mi_scaler_pipe {
	mi_scaler *sc = nullptr;
	int batch = 0, depth = 0;
	size_t src_pitch = 0, dst_pitch = 0;
	hipStream_t s_up = nullptr, s_down = nullptr;
	struct Slot {
		uint8_t *h_src = nullptr, *h_dst = nullptr, *d_src = nullptr, *d_dst = nullptr;
		hipEvent_t ev_up = nullptr, ev_k = nullptr, ev_down = nullptr;
		int nframes = 0;
		bool used = false;
	};
	std::vector<Slot> slots;
	uint64_t submitted = 0, collected = 0;
	bool acquired = false;
};

extern "C" {

void mi_scaler_pipe_destroy(mi_scaler_pipe *p) {
	if (!p) return;
	(void)hipSetDevice(p->sc->ctx->device);
	if (p->s_up) (void)hipStreamSynchronize(p->s_up);
	(void)hipStreamSynchronize(p->sc->ctx->stream);
	if (p->s_down) (void)hipStreamSynchronize(p->s_down);
	for (auto &sl : p->slots) {
		if (sl.h_src) (void)hipHostFree(sl.h_src);
		if (sl.h_dst) (void)hipHostFree(sl.h_dst);
		if (sl.d_src) (void)hipFree(sl.d_src);
		if (sl.d_dst) (void)hipFree(sl.d_dst);
		for (hipEvent_t e : {sl.ev_up, sl.ev_k, sl.ev_down})
			if (e) (void)hipEventDestroy(e);
	}
	if (p->s_up) (void)hipStreamDestroy(p->s_up);
	if (p->s_down) (void)hipStreamDestroy(p->s_down);
	delete p;
}

int mi_scaler_pipe_create(mi_scaler *s, int batch_frames, int depth, mi_scaler_pipe **out) {
	MI_CHECK_ARG(s && out && batch_frames > 0 && batch_frames <= 65535 && depth >= 1 && depth <= 8);
	*out = nullptr;
	if (s->ctx->activate() != MI_OK) return MI_ENODEV;
	mi_scaler_pipe *p = new mi_scaler_pipe();
	p->sc = s, p->batch = batch_frames, p->depth = depth;
	p->src_pitch = (s->src_bytes + 31) & ~(size_t)15; // slack for the kernels' 16-byte row loads
	p->dst_pitch = (s->dst_bytes + 15) & ~(size_t)15;
	p->slots.resize((size_t)depth);
	bool ok = hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking) == hipSuccess;
	for (auto &sl : p->slots) {
		const size_t sb = (size_t)batch_frames * p->src_pitch, db = (size_t)batch_frames * p->dst_pitch;
		ok = ok && hipHostMalloc((void **)&sl.h_src, sb, hipHostMallocDefault) == hipSuccess && hipHostMalloc((void **)&sl.h_dst, db, hipHostMallocDefault) == hipSuccess &&
		     hipMalloc((void **)&sl.d_src, sb + 32) == hipSuccess && hipMalloc((void **)&sl.d_dst, db + 32) == hipSuccess &&
		     hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&sl.ev_k, hipEventDisableTiming) == hipSuccess &&
		     hipEventCreateWithFlags(&sl.ev_down, hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) {
		mi::set_error("mi_scaler_pipe_create: streams / events / %d x %d frames of pinned and device memory could not be had", depth, batch_frames);
		mi_scaler_pipe_destroy(p);
		return MI_ENOMEM;
	}
	*out = p;
	return MI_OK;
}

int mi_scaler_pipe_in_flight(const mi_scaler_pipe *p) { return p ? (int)(p->submitted - p->collected) : MI_EINVAL; }

int mi_scaler_pipe_acquire(mi_scaler_pipe *p, uint8_t **h_src, size_t *src_pitch) {
	MI_CHECK_ARG(p && h_src);
	if ((int)(p->submitted - p->collected) >= p->depth) {
		mi::set_error("mi_scaler_pipe_acquire: %d batches in flight; collect the oldest first", p->depth);
		return MI_EINVAL;
	}
	mi_scaler_pipe::Slot &sl = p->slots[(size_t)(p->submitted % (uint64_t)p->depth)];
	if (sl.used) { // the upload that last read this staging buffer is long done (its batch was collected); be sure all the same
		if (p->sc->ctx->activate() != MI_OK) return MI_ENODEV;
		MI_HIP(hipEventSynchronize(sl.ev_up));
	}
	*h_src = sl.h_src;
	if (src_pitch) *src_pitch = p->src_pitch;
	p->acquired = true;
	return MI_OK;
}

int mi_scaler_pipe_submit(mi_scaler_pipe *p, int nframes) {
	MI_CHECK_ARG(p && nframes > 0 && nframes <= p->batch);
	if (!p->acquired) {
		mi::set_error("mi_scaler_pipe_submit without mi_scaler_pipe_acquire");
		return MI_EINVAL;
	}
	mi_ctx *c = p->sc->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	mi_scaler_pipe::Slot &sl = p->slots[(size_t)(p->submitted % (uint64_t)p->depth)];
	MI_HIP(hipMemcpyAsync(sl.d_src, sl.h_src, (size_t)nframes * p->src_pitch, hipMemcpyHostToDevice, p->s_up));
	MI_HIP(hipEventRecord(sl.ev_up, p->s_up));
	MI_HIP(hipStreamWaitEvent(c->stream, sl.ev_up, 0));
	if (sl.used) MI_HIP(hipStreamWaitEvent(c->stream, sl.ev_down, 0)); // the download that last read this slot's output
	const int rc = mi_scaler_process(p->sc, nframes, sl.d_src, p->src_pitch, sl.d_dst, p->dst_pitch);
	if (rc != MI_OK) return rc;
	MI_HIP(hipEventRecord(sl.ev_k, c->stream));
	MI_HIP(hipStreamWaitEvent(p->s_down, sl.ev_k, 0));
	MI_HIP(hipMemcpyAsync(sl.h_dst, sl.d_dst, (size_t)nframes * p->dst_pitch, hipMemcpyDeviceToHost, p->s_down));
	MI_HIP(hipEventRecord(sl.ev_down, p->s_down));
	sl.nframes = nframes;
	sl.used = true;
	p->acquired = false;
	p->submitted++;
	return MI_OK;
}

int mi_scaler_pipe_collect(mi_scaler_pipe *p, const uint8_t **h_dst, size_t *dst_pitch, int *nframes) {
	MI_CHECK_ARG(p && h_dst);
	if (p->submitted == p->collected) {
		mi::set_error("mi_scaler_pipe_collect: nothing in flight");
		return MI_EINVAL;
	}
	if (p->sc->ctx->activate() != MI_OK) return MI_ENODEV;
	mi_scaler_pipe::Slot &sl = p->slots[(size_t)(p->collected % (uint64_t)p->depth)];
	MI_HIP(hipEventSynchronize(sl.ev_down));
	*h_dst = sl.h_dst;
	if (dst_pitch) *dst_pitch = p->dst_pitch;
	if (nframes) *nframes = sl.nframes;
	p->collected++;
	return MI_OK;
}

} // extern "C"
