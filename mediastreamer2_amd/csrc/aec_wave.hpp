// aec_wave.hpp -- the AEC kernels in their one-wavefront-per-stream form (included by aec.hip,
// inside its anonymous namespace, after the shared structs/helpers).
//
// Mapping: ONE 64-lane wavefront owns one stream; lane l holds K = F/64 consecutive bins (and the
// K consecutive time samples l*K..l*K+K-1) in registers.  Consequences on gfx950:
//   * no workgroup barriers between waves: the only cross-lane exchange is the FFT, through LDS,
//     inside one wave; every radix-4 stage keeps all 64 lanes busy (F=256: 64 butterflies/stage);
//   * the time-domain vectors (mic, notch output, responses, errors) never touch LDS: serial
//     recurrences and the library-ordered sums read them with v_readlane straight from VGPRs;
//   * the streaming pass moves 32 contiguous bytes per lane per block (two dwordx4) with the next
//     block's loads in flight;
//   * ~11 KB of LDS per stream -> 13 streams resident per CU.
// Arithmetic (operation order, twiddles, accumulation order over blocks) is unchanged from the
// workgroup-per-stream version, so results are bit-identical to it.

template <int F>
struct alignas(16) WLds {
	float2 zbuf[F];      // complex FFT work
	float tbuf[2 * F];   // time-domain exchange / inverse-transform staging
	float spec[2 * F];   // bin-interleaved spectrum exchange, filterbank products
	float2 tw[F], super[F];
	uint16_t perm[F];
	float prop[64];
	float band[4 * NB_BANDS + 8];
	float vec[F]; // per-bin exchange (neighbour access in the post-filter)
};

#define WSYNC() __syncthreads() /* one wave per workgroup: an LDS fence, no cross-wave wait */

// where w_rfft_inverse leaves its 2F time samples (re, im interleaved = consecutive samples): no copy into L.tbuf
template <typename LT>
__device__ __forceinline__ float *w_time(LT &L) { return reinterpret_cast<float *>(L.zbuf); }

template <int F, typename LT>
__device__ void w_cfft(LT &L, const AecTables &T, const float2 *src, bool inverse) {
	constexpr int K = F / 64;
	const int lane = threadIdx.x;
	float2 val[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		val[k] = src[L.perm[lane * K + k]];
	}
	// The deepest stage (m = 1) of lane i works on elements i p .. i p + p - 1: with K = p those are the K values the lane has
	// just gathered, so the stage runs in registers (same expressions, twiddle tw[0]) and one LDS round trip is gone.
	constexpr bool kRegStage0 = (K == plan_p(F, 0));
	if constexpr (kRegStage0) {
		float2 w0 = L.tw[0];
		if (inverse) w0.y = -w0.y;
		if constexpr (K == 2) {
			const float2 t = cmulf(val[1], w0);
			const float2 a = val[0];
			val[1] = make_float2(a.x - t.x, a.y - t.y);
			val[0] = make_float2(a.x + t.x, a.y + t.y);
		} else {
			const float2 s0 = cmulf(val[1], w0);
			const float2 s1 = cmulf(val[2], w0);
			const float2 s2 = cmulf(val[3], w0);
			float2 f0 = val[0];
			const float2 s5 = make_float2(f0.x - s1.x, f0.y - s1.y);
			f0.x += s1.x;
			f0.y += s1.y;
			const float2 s3 = make_float2(s0.x + s2.x, s0.y + s2.y);
			const float2 s4 = make_float2(s0.x - s2.x, s0.y - s2.y);
			val[2] = make_float2(f0.x - s3.x, f0.y - s3.y);
			f0.x += s3.x;
			f0.y += s3.y;
			val[0] = f0;
			if (inverse) {
				val[1] = make_float2(s5.x - s4.y, s5.y + s4.x);
				val[3] = make_float2(s5.x + s4.y, s5.y - s4.x);
			} else {
				val[1] = make_float2(s5.x + s4.y, s5.y - s4.x);
				val[3] = make_float2(s5.x - s4.y, s5.y + s4.x);
			}
		}
	}
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) L.zbuf[lane * K + k] = val[k];
	WSYNC();
#pragma unroll
	for (int s = kRegStage0 ? 1 : 0; s < plan_n(F); ++s) {
		constexpr int FF = F;
		const int p = plan_p(FF, s), m = plan_m(FF, s), fs = plan_fs(FF, s);
		if (lane < F / p) {
			const int i = lane / m, j = lane - i * m;
			float2 *Fo = L.zbuf + i * (p * m) + j;
			if (p == 2) {
				float2 w = L.tw[j * fs];
				if (inverse) w.y = -w.y;
				const float2 t = cmulf(Fo[m], w);
				const float2 a = Fo[0];
				Fo[m] = make_float2(a.x - t.x, a.y - t.y);
				Fo[0] = make_float2(a.x + t.x, a.y + t.y);
			} else {
				float2 w1 = L.tw[j * fs], w2 = L.tw[j * fs * 2], w3 = L.tw[j * fs * 3];
				if (inverse) {
					w1.y = -w1.y;
					w2.y = -w2.y;
					w3.y = -w3.y;
				}
				const float2 s0 = cmulf(Fo[m], w1);
				const float2 s1 = cmulf(Fo[2 * m], w2);
				const float2 s2 = cmulf(Fo[3 * m], w3);
				float2 f0 = Fo[0];
				const float2 s5 = make_float2(f0.x - s1.x, f0.y - s1.y);
				f0.x += s1.x;
				f0.y += s1.y;
				const float2 s3 = make_float2(s0.x + s2.x, s0.y + s2.y);
				const float2 s4 = make_float2(s0.x - s2.x, s0.y - s2.y);
				Fo[2 * m] = make_float2(f0.x - s3.x, f0.y - s3.y);
				f0.x += s3.x;
				f0.y += s3.y;
				Fo[0] = f0;
				if (inverse) {
					Fo[m] = make_float2(s5.x - s4.y, s5.y + s4.x);
					Fo[3 * m] = make_float2(s5.x + s4.y, s5.y - s4.x);
				} else {
					Fo[m] = make_float2(s5.x + s4.y, s5.y - s4.x);
					Fo[3 * m] = make_float2(s5.x - s4.y, s5.y + s4.x);
				}
			}
		}
		WSYNC();
	}
}

// L.tbuf (2F time samples) -> this lane's K bins, scaled 1/N.  Bin 0 = (DC, Nyquist).
// src: the 2F time samples -- L.tbuf (the default) or L.zbuf, where the inverse transform leaves its result
template <int F, typename LT>
__device__ void w_rfft_forward(LT &L, const AecTables &T, float2 (&out)[F / 64], const float *src = nullptr) {
	if (src == nullptr) src = L.tbuf;
	constexpr int K = F / 64;
	const int lane = threadIdx.x;
	WSYNC();
	w_cfft<F>(L, T, reinterpret_cast<const float2 *>(src), false);
	const float scale = 1.f / (2 * F);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int b = lane * K + k;
		if (b == 0) {
			const float2 t0 = L.zbuf[0];
			out[k] = make_float2((t0.x + t0.y) * scale, (t0.x - t0.y) * scale);
		} else {
			// the library's loop visits kk = 1..F/2 and writes bins kk and F-kk from the same intermediates
			const bool upper = b >= F - b; // bins >= F/2 take the "F-kk" formulas (F/2 itself: the later write)
			const int kk = upper ? F - b : b;
			const float2 a = L.zbuf[kk], c = L.zbuf[F - kk];
			const float2 sw = L.super[kk];
			// f1 = (a.x + c.x, a.y - c.y), f2 = (a.x - c.x, a.y + c.y), tw = f2 sw; lower bins (f1 + tw) / 2, upper bins
			// (f1.x - tw.x, tw.y - f1.y) / 2: the upper form is the lower one with the signs of f1.y and tw.x flipped (exact)
			const v2f av = {a.x, a.y}, cv = {c.x, c.y};
			v2f f1, f2;
			asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(f1) : "v"(av), "v"(cv));
			asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(f2) : "v"(av), "v"(cv));
			v2f tw = pk_cmul(f2, (v2f){sw.x, sw.y}); // (f2r sw.x - f2i sw.y, f2i sw.x + f2r sw.y)
			const unsigned flip = upper ? 0x80000000u : 0u;
			f1.y = __uint_as_float(__float_as_uint(f1.y) ^ flip);
			tw.x = __uint_as_float(__float_as_uint(tw.x) ^ flip);
			const v2f r = (((v2f){.5f, .5f}) * (f1 + tw)) * (v2f){scale, scale};
			out[k] = make_float2(r.x, r.y);
		}
	}
	WSYNC();
}

// this lane's K bins -> 2F time samples in w_time(L), unscaled
template <int F, typename LT>
__device__ void w_rfft_inverse(LT &L, const AecTables &T, const float2 (&in)[F / 64]) {
	constexpr int K = F / 64;
	const int lane = threadIdx.x;
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.spec[2 * (lane * K + k)] = in[k].x;
		L.spec[2 * (lane * K + k) + 1] = in[k].y;
	}
	WSYNC();
	float2 *tmp = reinterpret_cast<float2 *>(L.tbuf);
	float2 t[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int i = lane * K + k;
		if (i == 0) {
			t[k] = make_float2(L.spec[0] + L.spec[1], L.spec[0] - L.spec[1]);
		} else {
			const bool upper = i >= F - i;
			const int kk = upper ? F - i : i;
			// fnkc = conj(spec[F - kk]); fek = fk + fnkc; d = fk - fnkc; fok = d conj(sw); lower bins fek + fok, upper bins
			// conj(fek - fok) -- the conjugations are operand signs, the upper form is the lower one with fok negated and the
			// imaginary part of the result negated (sign flips: exact)
			const v2f fk = {L.spec[2 * kk], L.spec[2 * kk + 1]};
			const v2f s2 = {L.spec[2 * (F - kk)], L.spec[2 * (F - kk) + 1]};
			const float2 sw = L.super[kk];
			const v2f sv = {sw.x, sw.y};
			v2f fek, d, p, q, fok;
			asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(fek) : "v"(fk), "v"(s2)); // (fk.x + s2.x, fk.y - s2.y)
			asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(fk), "v"(s2));   // (fk.x - s2.x, fk.y + s2.y)
			asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(d), "v"(sv));                                            // (d.x sw.x, d.y sw.x)
			asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q) : "v"(d), "v"(sv)); // (d.y (-sw.y), d.x (-sw.y))
			asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(fok) : "v"(p), "v"(q));
			const unsigned flip = upper ? 0x80000000u : 0u;
			fok.x = __uint_as_float(__float_as_uint(fok.x) ^ flip);
			fok.y = __uint_as_float(__float_as_uint(fok.y) ^ flip);
			v2f r = fek + fok;
			r.y = __uint_as_float(__float_as_uint(r.y) ^ flip);
			t[k] = make_float2(r.x, r.y);
		}
	}
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) tmp[lane * K + k] = t[k];
	WSYNC();
	w_cfft<F>(L, T, tmp, true); // ends with an LDS fence: the 2F time samples are in L.zbuf (see w_time)
}


// ---- the library's SERIAL loops over a frame (sums in C loop order, the DC notch and de-emphasis recurrences), element
// index = lane * K + k.  Each is a chain of dependent float operations across the 64 lanes.  It runs SYSTOLICALLY: in every
// step each lane takes the running value of the lane before it (v_*_dpp wave_shr:1, wave_shl:1 for a descending loop) and
// adds its own K elements in order.  Lane l's input is final after step l-1, so lane l is final after step l and later
// steps only recompute the same value: 64 steps leave the result in lane 63 (lane 0 descending) -- bit for bit the
// sequential loop, at one VALU instruction per operation instead of v_readlane + operation (2.5x faster measured,
// scripts/micro/dpp_chain.hip).  Independent chains are stepped in one loop so that each fills the other's DPP latency.
__device__ __forceinline__ float dpp_shr1_zero(float v) { // lane l <- lane l-1, lane 0 <- 0.0f
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shr1(float first, float v) { // lane l <- lane l-1, lane 0 <- first
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float last, float v) { // lane l <- lane l+1, lane 63 <- last
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(last), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// Sum / maximum over the 64 lanes in a fixed butterfly order, in registers: four DPP exchanges inside the rows of 16
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then row_bcast:15 and row_bcast:31 -- no LDS, no index arithmetic.
// For values whose summation order the library does not fix (a tree either way).  The result is in every lane (readlane 63).
template <typename Op>
__device__ __forceinline__ float wave_tree(float v, Op op) {
	auto dpp = [](float x, auto ctrl) {
		return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
	};
	v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));  // quad_perm [1,0,3,2]
	v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));  // quad_perm [2,3,0,1]
	v = op(v, dpp(v, std::integral_constant<int, 0x141>{})); // row_half_mirror
	v = op(v, dpp(v, std::integral_constant<int, 0x140>{})); // row_mirror: every lane of a row holds the row's result
	const float r0 = rdlane(v, 0), r1 = rdlane(v, 16), r2 = rdlane(v, 32), r3 = rdlane(v, 48);
	return op(op(r0, r1), op(r2, r3));
}

template <int K>
struct WSeq {
	static constexpr int KP = K >= 2 ? K / 2 : 1; // the library adds the products in pairs: part = (0 + x0 y0) + x1 y1; sum += part
	__device__ static void pairs(const float (&x)[K], const float (&y)[K], float (&part)[KP]) {
		if constexpr (K == 1) {
			// pair (2i, 2i+1) = neighbouring lanes: the even lane carries the pair's sum, the odd lane hands the running sum
			// on unchanged (s + -0.0f == s for every s)
			const float pr = x[0] * y[0];
			const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pr), 0xB1, 0xf, 0xf, true)); // quad_perm [1,0,3,2]
			float p = 0;
			p = p + pr;
			p = p + nb;
			part[0] = (threadIdx.x & 1) ? -0.0f : p;
		} else {
#pragma unroll
			for (int k = 0; k < K; k += 2) {
				float p = 0;
				p = p + x[k] * y[k];
				p = p + x[k + 1] * y[k + 1];
				part[k / 2] = p;
			}
		}
	}
	// NC sums at once: sum = 0; for l in 0..63: for k: sum += part[k] of lane l
	template <int NC>
	__device__ static void chain_up(const float (&part)[NC][KP], float (&sum)[NC]) {
		float s[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) s[c] = 0;
#pragma unroll 4
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				s[c] = dpp_shr1_zero(s[c]) + part[c][0];
#pragma unroll
				for (int k = 1; k < KP; ++k) s[c] = s[c] + part[c][k];
			}
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) sum[c] = rdlane(s[c], 63);
	}
	__device__ static float inner_prod(const float (&x)[K], const float (&y)[K]) {
		float part[1][KP], sum[1];
		pairs(x, y, part[0]);
		chain_up<1>(part, sum);
		return sum[0];
	}
	__device__ static void inner_prod3(const float (&x0)[K], const float (&y0)[K], const float (&x1)[K], const float (&y1)[K],
	                                   const float (&x2)[K], const float (&y2)[K], float &r0, float &r1, float &r2) {
		float part[3][KP], sum[3];
		pairs(x0, y0, part[0]);
		pairs(x1, y1, part[1]);
		pairs(x2, y2, part[2]);
		chain_up<3>(part, sum);
		r0 = sum[0], r1 = sum[1], r2 = sum[2];
	}
	// two descending dot products at once: acc = init; for l = 63..0: for k = K-1..0: acc += a[k] b[k] of lane l
	__device__ static void dot_desc2(float init0, const float (&a0)[K], const float (&b0)[K], float init1, const float (&a1)[K],
	                                 const float (&b1)[K], float &r0, float &r1) {
		float p0[K], p1[K];
#pragma unroll
		for (int k = 0; k < K; ++k) p0[k] = a0[k] * b0[k], p1[k] = a1[k] * b1[k];
		float s0 = init0, s1 = init1;
#pragma unroll 4
		for (int l = 0; l < 64; ++l) { // the two chains ride in one register pair: K packed additions per step
			v2f t = {dpp_shl1(init0, s0), dpp_shl1(init1, s1)};
#pragma unroll
			for (int k = K - 1; k >= 0; --k) t = t + (v2f){p0[k], p1[k]};
			s0 = t.x, s1 = t.y;
		}
		r0 = rdlane(s0, 0), r1 = rdlane(s1, 0);
	}
};

// filter_dc_notch16 of the library: vout = m0 + vin; m0 = m1 + 2 (-vin + radius vout); m1 = vin - den2 vout; out = radius vout,
// sample after sample.  (m0, m1) come in as the state before the frame and go out as the state after it (all lanes alike).
// One sample of the notch in four instructions: add, packed multiply, packed add with source modifiers, fma.
//   tu = (radius vout, den2 vout);  am = (tu.x + (-vin), (-tu.y) + vin) = (-vin + radius vout, vin - den2 vout) bit for bit;
//   m0' = fma(2, am.x, m1) = m1 + 2 am.x bit for bit (2 x is exact).  HI selects the half of `pair` that holds vin.
template <bool HI>
__device__ __forceinline__ void notch_sample(v2f rc, v2f pair, float vin, float &a0, float &a1, float &out) {
	v2f vo, tu, am;
	vo.x = a0 + vin;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(tu) : "v"(rc), "v"(vo));
	if constexpr (HI) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(am) : "v"(tu), "v"(pair));
	else asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(am) : "v"(tu), "v"(pair));
	a0 = __builtin_fmaf(2.f, am.x, a1);
	a1 = am.y;
	out = tu.x;
}
template <int K>
__device__ __forceinline__ void w_dc_notch(const float (&in)[K], float radius, float den2, float &m0io, float &m1io, float (&out)[K]) {
	const float i0 = m0io, i1 = m1io;
	float m0 = i0, m1 = i1;
	const v2f rc = {radius, den2};
	constexpr int NP = (K + 1) / 2;
	v2f pr[NP];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (k & 1) pr[k / 2].y = in[k];
		else pr[k / 2].x = in[k];
	}
#pragma unroll 2
	for (int l = 0; l < 64; ++l) {
		float a0 = dpp_shr1(i0, m0), a1 = dpp_shr1(i1, m1);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (k & 1) notch_sample<true>(rc, pr[k / 2], in[k], a0, a1, out[k]);
			else notch_sample<false>(rc, pr[k / 2], in[k], a0, a1, out[k]);
		}
		m0 = a0, m1 = a1;
	}
	m0io = rdlane(m0, 63);
	m1io = rdlane(m1, 63);
}

// de-emphasis of the output: t = d + 0.9 mem; mem = t, sample after sample
template <int K>
__device__ __forceinline__ void w_deemphasis(const float (&d)[K], float &memio, float (&out)[K]) {
	const float init = memio;
	float m = init;
#pragma unroll 2
	for (int l = 0; l < 64; ++l) {
		float a = dpp_shr1(init, m);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const float t = d[k] + .9f * a;
			a = t;
			out[k] = t;
		}
		m = a;
	}
	memio = rdlane(m, 63);
}

template <int K>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[K]) {
	if constexpr (K == 4) {
		const float4 t = *reinterpret_cast<const float4 *>(p);
		v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
	} else if constexpr (K == 2) {
		const float2 t = *reinterpret_cast<const float2 *>(p);
		v[0] = t.x, v[1] = t.y;
	} else {
		v[0] = *p;
	}
}
template <int K>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[K]) {
	if constexpr (K == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
	else if constexpr (K == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
	else *p = v[0];
}
template <int K>
__device__ __forceinline__ void load_bins(const float2 *p, float2 (&v)[K]) {
	if constexpr (K == 1) {
		v[0] = *p;
	} else {
#pragma unroll
		for (int k = 0; k < K; k += 2) {
			const float4 t = *reinterpret_cast<const float4 *>(p + k);
			v[k] = make_float2(t.x, t.y);
			v[k + 1] = make_float2(t.z, t.w);
		}
	}
}
template <int K>
__device__ __forceinline__ void store_bins(float2 *p, const float2 (&v)[K]) {
	if constexpr (K == 1) {
		*p = v[0];
	} else {
#pragma unroll
		for (int k = 0; k < K; k += 2) *reinterpret_cast<float4 *>(p + k) = make_float4(v[k].x, v[k].y, v[k + 1].x, v[k + 1].y);
	}
}

