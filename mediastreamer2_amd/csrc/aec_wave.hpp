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

template <int F>
__device__ void w_cfft(WLds<F> &L, const float2 *src, bool inverse) {
	constexpr int K = F / 64;
	const int lane = threadIdx.x;
	float2 val[K];
#pragma unroll
	for (int k = 0; k < K; ++k) val[k] = src[L.perm[lane * K + k]];
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) L.zbuf[lane * K + k] = val[k];
	WSYNC();
#pragma unroll
	for (int s = 0; s < plan_n(F); ++s) {
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
template <int F>
__device__ void w_rfft_forward(WLds<F> &L, float2 (&out)[F / 64]) {
	constexpr int K = F / 64;
	const int lane = threadIdx.x;
	WSYNC();
	w_cfft<F>(L, reinterpret_cast<const float2 *>(L.tbuf), false);
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
			const float f2r = a.x - c.x, f2i = a.y + c.y;
			const float f1r = a.x + c.x, f1i = a.y - c.y;
			const float twr = f2r * sw.x - f2i * sw.y;
			const float twi = f2i * sw.x + f2r * sw.y;
			if (!upper) out[k] = make_float2((.5f * (f1r + twr)) * scale, (.5f * (f1i + twi)) * scale);
			else out[k] = make_float2((.5f * (f1r - twr)) * scale, (.5f * (twi - f1i)) * scale);
		}
	}
	WSYNC();
}

// this lane's K bins -> L.tbuf (2F time samples), unscaled
template <int F>
__device__ void w_rfft_inverse(WLds<F> &L, const float2 (&in)[F / 64]) {
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
			const float2 fk = make_float2(L.spec[2 * kk], L.spec[2 * kk + 1]);
			const float2 fnkc = make_float2(L.spec[2 * (F - kk)], -L.spec[2 * (F - kk) + 1]);
			float2 sw = L.super[kk];
			sw.y = -sw.y;
			const float2 fek = make_float2(fk.x + fnkc.x, fk.y + fnkc.y);
			const float2 d = make_float2(fk.x - fnkc.x, fk.y - fnkc.y);
			const float2 fok = cmulf(d, sw);
			if (!upper) {
				t[k] = make_float2(fek.x + fok.x, fek.y + fok.y);
			} else {
				float2 c = make_float2(fek.x - fok.x, fek.y - fok.y);
				c.y *= -1;
				t[k] = c;
			}
		}
	}
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) tmp[lane * K + k] = t[k];
	WSYNC();
	w_cfft<F>(L, tmp, true);
	float2 r[K];
#pragma unroll
	for (int k = 0; k < K; ++k) r[k] = L.zbuf[lane * K + k];
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.tbuf[2 * (lane * K + k)] = r[k].x;
		L.tbuf[2 * (lane * K + k) + 1] = r[k].y;
	}
	WSYNC();
}

// library-ordered sums over register-resident vectors (element index = lane*K + k)
template <int K>
struct WSeq {
	__device__ static float inner_prod(const float (&x)[K], const float (&y)[K]) {
		if constexpr (K == 1) { // the library's pairs (2i, 2i+1) are neighbouring lanes
			const float pr = x[0] * y[0];
			float sum = 0;
#pragma unroll 2
			for (int l = 0; l < 64; l += 2) {
				float p = 0;
				p = p + rdlane(pr, l);
				p = p + rdlane(pr, l + 1);
				sum = sum + p;
			}
			return sum;
		} else {
			float part[K / 2];
#pragma unroll
			for (int k = 0; k < K; k += 2) {
				float p = 0;
				p = p + x[k] * y[k];
				p = p + x[k + 1] * y[k + 1];
				part[k / 2] = p;
			}
			float sum = 0;
#pragma unroll 2
			for (int l = 0; l < 64; ++l) {
#pragma unroll
				for (int k = 0; k < K / 2; ++k) sum = sum + rdlane(part[k], l);
			}
			return sum;
		}
	}
	__device__ static float dot_desc(float init, const float (&a)[K], const float (&b)[K]) {
		float p[K];
#pragma unroll
		for (int k = 0; k < K; ++k) p[k] = a[k] * b[k];
		float acc = init;
#pragma unroll 2
		for (int l = 63; l >= 0; --l) {
#pragma unroll
			for (int k = K - 1; k >= 0; --k) acc = acc + rdlane(p[k], l);
		}
		return acc;
	}
};

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

// ===================================================================== MDF canceller, one frame
template <int F>
__global__ __launch_bounds__(64, (F == 256 ? 3 : (F == 128 ? 5 : 6))) void aec_mdf_wave_kernel(AecArgs a) {
	__shared__ WLds<F> L;
	using SL = SmallLayout<F>;
	constexpr int N = 2 * F, K = F / 64;
	const int s = a.first + blockIdx.x;
	if (a.run && !a.run[s]) return;
	const int lane = threadIdx.x;
	const int e0 = lane * K; // first element (sample / bin) this lane owns
	const int M = a.M;
	float *sm = a.small + (size_t)s * a.small_stride;
	float2 *Xs = reinterpret_cast<float2 *>(a.X + (size_t)s * (M + 1) * N);
	float2 *Ws = reinterpret_cast<float2 *>(a.W + (size_t)s * M * N);
	float2 *FGs = reinterpret_cast<float2 *>(a.FG + (size_t)s * M * N);
	AecScalars sc = a.scal[s];

	// ---- tables + inputs
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.tw[e0 + k] = a.t.tw[e0 + k];
		L.super[e0 + k] = a.t.super[e0 + k];
		L.perm[e0 + k] = a.t.perm[e0 + k];
	}
	if (lane < M) L.prop[lane] = sm[SL::PROP + lane];
	int mic_i[K];
	float fin[K], xnew[K], xprev[K];
	{
		const int16_t *mp = a.mic + (size_t)s * a.stride + e0, *rp = a.ref + (size_t)s * a.stride + e0;
		float far[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			mic_i[k] = mp[k];
			fin[k] = (float)mic_i[k];
			far[k] = (float)rp[k];
		}
		float prev = __shfl_up(far[K - 1], 1);
		if (lane == 0) prev = sc.memX;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			xnew[k] = far[k] - .9f * prev;
			prev = far[k];
		}
		sc.memX = rdlane(far[K - 1], 63);
	}
	load_vec<K>(sm + SL::XPREV + e0, xprev);
	store_vec<K>(sm + SL::XPREV + e0, xnew);
	store_vec<K>(L.tbuf + e0, xprev);
	store_vec<K>(L.tbuf + F + e0, xnew);
	bool satl = false;
#pragma unroll
	for (int k = 0; k < K; ++k) satl |= (mic_i[k] <= -32000 || mic_i[k] >= 32000);
	const int any_sat = __any(satl);
	sc.cancel_count++;

	// ---- DC notch (serial IIR) + pre-emphasis, Sxx
	float input[K];
	{
		const float radius = a.notch_radius;
		const float den2 = (float)(radius * radius + .7 * (1 - radius) * (1 - radius));
		float m0 = sc.notch0, m1 = sc.notch1;
		float v[K];
#pragma unroll 2
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const float vin = rdlane(fin[k], l);
				const float vout = m0 + vin;
				m0 = m1 + 2 * (-vin + radius * vout);
				m1 = vin - den2 * vout;
				const float y = radius * vout;
				if (lane == l) v[k] = y;
			}
		}
		sc.notch0 = m0;
		sc.notch1 = m1;
		float vprev = __shfl_up(v[K - 1], 1);
		if (lane == 0) vprev = sc.memD;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			input[k] = v[k] - .9f * vprev;
			vprev = v[k];
		}
		sc.memD = rdlane(v[K - 1], 63);
	}
	float Sxx = WSeq<K>::inner_prod(xnew, xnew);

	// ---- X0 = FFT(x) into the ring
	float2 X0[K];
	w_rfft_forward<F>(L, X0);
	const int head = (sc.xhead + M) % (M + 1);
	sc.xhead = head;
	store_bins<K>(Xs + (size_t)head * F + e0, X0);
	auto xslot = [&](int j) { return (size_t)((head + j) % (M + 1)) * F + e0; };

	float2 Eprev[K];
	float p1[K];
	load_bins<K>(reinterpret_cast<const float2 *>(sm + SL::E) + e0, Eprev);
	load_vec<K>(sm + SL::POWER1 + e0, p1);
	const float p1_F = sm[SL::POWER1 + F];

	// ---- proportional step
	if (sc.adapted) {
		if (lane == 0) {
			float max_sum = 1, prop_sum = 1;
			for (int i = 0; i < M; ++i) {
				const float p = sqrt_via_double(1.0f + sm[SL::WNORM + i]);
				L.prop[i] = p;
				if (p > max_sum) max_sum = p;
			}
			for (int i = 0; i < M; ++i) {
				L.prop[i] += .1f * max_sum;
				prop_sum += L.prop[i];
			}
			for (int i = 0; i < M; ++i) L.prop[i] = (.99f * L.prop[i]) / prop_sum;
		}
		WSYNC();
		if (lane < M) sm[SL::PROP + lane] = L.prop[lane];
	}
	WSYNC();
	const bool do_grad = (sc.saturated == 0);
	if (!do_grad) sc.saturated--;

	auto grad = [&](float2 (&w)[K], const float2 (&x)[K], float prop) {
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (e0 + k == 0) {
				const float W0 = prop * p1[k], WN = prop * p1_F;
				w[k].x += W0 * (x[k].x * Eprev[k].x);
				w[k].y += WN * (x[k].y * Eprev[k].y);
			} else {
				const float Wt = prop * p1[k];
				w[k].x += Wt * ((x[k].x * Eprev[k].x) + x[k].y * Eprev[k].y);
				w[k].y += Wt * (((-x[k].y) * Eprev[k].x) + x[k].x * Eprev[k].y);
			}
		}
	};

	// ---- AUMDF blocks first
	const int jc = (M > 1) ? (sc.cancel_count % (M - 1)) + 1 : -1;
	float2 wsp0[K], wspc[K];
#pragma unroll
	for (int k = 0; k < K; ++k) wsp0[k] = wspc[k] = make_float2(0, 0);
	for (int pass = 0; pass < 2; ++pass) {
		const int jb = pass == 0 ? 0 : jc;
		if (jb < 0) break;
		float2 w[K], x1[K];
		load_bins<K>(Ws + (size_t)jb * F + e0, w);
		if (do_grad) {
			load_bins<K>(Xs + xslot(jb + 1), x1);
			grad(w, x1, L.prop[jb]);
		}
		w_rfft_inverse<F>(L, w);
		{
			float z[K];
#pragma unroll
			for (int k = 0; k < K; ++k) z[k] = 0.f;
			store_vec<K>(L.tbuf + F + e0, z);
		}
		w_rfft_forward<F>(L, w);
		store_bins<K>(Ws + (size_t)jb * F + e0, w);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (pass == 0) wsp0[k] = w[k];
			else wspc[k] = w[k];
		}
	}

	// ---- one streaming pass over X, FG, W (next block's loads in flight)
	float2 yfg[K], ybgs[K];
#pragma unroll
	for (int k = 0; k < K; ++k) yfg[k] = ybgs[k] = make_float2(0, 0);
	{
		float2 xj[K], xn[K], fg[K], wl[K];
#pragma unroll
		for (int k = 0; k < K; ++k) xj[k] = X0[k];
		load_bins<K>(Xs + xslot(1), xn);
		load_bins<K>(FGs + e0, fg);
		load_bins<K>(Ws + e0, wl);
		for (int j = 0; j < M; ++j) {
			float2 xn2[K], fg2[K], wl2[K];
			if (j + 1 < M) {
				load_bins<K>(Xs + xslot(j + 2), xn2);
				load_bins<K>(FGs + (size_t)(j + 1) * F + e0, fg2);
				load_bins<K>(Ws + (size_t)(j + 1) * F + e0, wl2);
			} else {
#pragma unroll
				for (int k = 0; k < K; ++k) xn2[k] = xn[k], fg2[k] = fg[k], wl2[k] = wl[k];
			}
			float2 w[K];
			if (j == 0) {
#pragma unroll
				for (int k = 0; k < K; ++k) w[k] = wsp0[k];
			} else if (j == jc) {
#pragma unroll
				for (int k = 0; k < K; ++k) w[k] = wspc[k];
			} else {
#pragma unroll
				for (int k = 0; k < K; ++k) w[k] = wl[k];
				if (do_grad) {
					grad(w, xn, L.prop[j]);
					store_bins<K>(Ws + (size_t)j * F + e0, w);
				}
			}
			float nn = 0;
#pragma unroll
			for (int k = 0; k < K; ++k) {
				if (e0 + k == 0) {
					yfg[k].x += xj[k].x * fg[k].x;
					yfg[k].y += xj[k].y * fg[k].y;
					ybgs[k].x += xj[k].x * w[k].x;
					ybgs[k].y += xj[k].y * w[k].y;
				} else {
					yfg[k].x += (xj[k].x * fg[k].x - xj[k].y * fg[k].y);
					yfg[k].y += (xj[k].y * fg[k].x + xj[k].x * fg[k].y);
					ybgs[k].x += (xj[k].x * w[k].x - xj[k].y * w[k].y);
					ybgs[k].y += (xj[k].y * w[k].x + xj[k].x * w[k].y);
				}
				nn += w[k].x * w[k].x + w[k].y * w[k].y;
			}
			for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o);
			if (lane == 0) sm[SL::WNORM + j] = nn; // feeds the NEXT frame's proportional step
#pragma unroll
			for (int k = 0; k < K; ++k) xj[k] = xn[k], xn[k] = xn2[k], fg[k] = fg2[k], wl[k] = wl2[k];
		}
	}

	// ---- time-domain responses
	float efg[K], ybg[K], e1[K], e2[K], dresp[K];
	w_rfft_inverse<F>(L, yfg);
	load_vec<K>(L.tbuf + F + e0, efg);
#pragma unroll
	for (int k = 0; k < K; ++k) e1[k] = input[k] - efg[k];
	w_rfft_inverse<F>(L, ybgs);
	load_vec<K>(L.tbuf + F + e0, ybg);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		e2[k] = input[k] - ybg[k];
		dresp[k] = efg[k] - ybg[k];
	}
	const float Sff = WSeq<K>::inner_prod(e1, e1);
	const float Dbf = 10 + WSeq<K>::inner_prod(dresp, dresp);
	float See = WSeq<K>::inner_prod(e2, e2);

	// ---- two-path control
	sc.Davg1 = .6f * sc.Davg1 + .4f * (Sff - See);
	sc.Davg2 = .85f * sc.Davg2 + .15f * (Sff - See);
	sc.Dvar1 = .36f * sc.Dvar1 + (.4f * Sff) * (.4f * Dbf);
	sc.Dvar2 = .7225f * sc.Dvar2 + (.15f * Sff) * (.15f * Dbf);
	bool update_foreground = false;
	if ((Sff - See) * fabsf(Sff - See) > Sff * Dbf) update_foreground = true;
	else if (sc.Davg1 * fabsf(sc.Davg1) > .5f * sc.Dvar1) update_foreground = true;
	else if (sc.Davg2 * fabsf(sc.Davg2) > .25f * sc.Dvar2) update_foreground = true;
	if (update_foreground) {
		sc.Davg1 = sc.Davg2 = 0;
		sc.Dvar1 = sc.Dvar2 = 0;
		for (int j = 0; j < M; ++j) {
			float2 w[K];
			load_bins<K>(Ws + (size_t)j * F + e0, w);
			store_bins<K>(FGs + (size_t)j * F + e0, w);
		}
		float h0[K], h1[K];
		load_vec<K>(a.t.hann + e0, h0);
		load_vec<K>(a.t.hann + F + e0, h1);
#pragma unroll
		for (int k = 0; k < K; ++k) efg[k] = h1[k] * efg[k] + h0[k] * ybg[k];
	} else {
		bool reset_background = false;
		if ((-(Sff - See)) * fabsf(Sff - See) > 4.f * (Sff * Dbf)) reset_background = true;
		if ((-sc.Davg1) * fabsf(sc.Davg1) > 4.f * sc.Dvar1) reset_background = true;
		if ((-sc.Davg2) * fabsf(sc.Davg2) > 4.f * sc.Dvar2) reset_background = true;
		if (reset_background) {
			for (int j = 0; j < M; ++j) {
				float2 w[K];
				load_bins<K>(FGs + (size_t)j * F + e0, w);
				store_bins<K>(Ws + (size_t)j * F + e0, w);
				float nn = 0;
#pragma unroll
				for (int k = 0; k < K; ++k) nn += w[k].x * w[k].x + w[k].y * w[k].y;
				for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o);
				if (lane == 0) sm[SL::WNORM + j] = nn;
			}
#pragma unroll
			for (int k = 0; k < K; ++k) {
				ybg[k] = efg[k];
				e2[k] = input[k] - efg[k];
			}
			See = Sff;
			sc.Davg1 = sc.Davg2 = 0;
			sc.Dvar1 = sc.Dvar2 = 0;
		}
	}

	// ---- output (serial de-emphasis) and correlations
	int out_i[K];
	{
		float d[K], tout[K];
#pragma unroll
		for (int k = 0; k < K; ++k) d[k] = input[k] - efg[k];
		float memE = sc.memE;
#pragma unroll 2
		for (int l = 0; l < 64; ++l) {
#pragma unroll
			for (int k = 0; k < K; ++k) {
				float t = rdlane(d[k], l);
				t = t + .9f * memE;
				memE = t;
				if (lane == l) tout[k] = t;
			}
		}
		sc.memE = memE;
#pragma unroll
		for (int k = 0; k < K; ++k) out_i[k] = word2int(tout[k]);
	}
	const float Sey = WSeq<K>::inner_prod(e2, ybg);
	const float Syy = WSeq<K>::inner_prod(ybg, ybg);
	const float Sdd = WSeq<K>::inner_prod(input, input);
	if (any_sat && sc.saturated == 0) sc.saturated = 1;

	// ---- error / response spectra
	float2 Ecur[K], Ycur[K];
	{
		float z[K];
#pragma unroll
		for (int k = 0; k < K; ++k) z[k] = 0.f;
		WSYNC();
		store_vec<K>(L.tbuf + e0, z);
		store_vec<K>(L.tbuf + F + e0, e2);
		w_rfft_forward<F>(L, Ecur);
		store_vec<K>(L.tbuf + e0, z);
		store_vec<K>(L.tbuf + F + e0, ybg);
		w_rfft_forward<F>(L, Ycur);
	}
	store_bins<K>(reinterpret_cast<float2 *>(sm + SL::E) + e0, Ecur);
	float Rf[K], Yf[K], Xf[K], Rf_F = 0, Yf_F = 0, Xf_F = 0;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (e0 + k == 0) {
			Rf[k] = Ecur[k].x * Ecur[k].x;
			Rf_F = Ecur[k].y * Ecur[k].y;
			Yf[k] = Ycur[k].x * Ycur[k].x;
			Yf_F = Ycur[k].y * Ycur[k].y;
			Xf[k] = X0[k].x * X0[k].x;
			Xf_F = X0[k].y * X0[k].y;
		} else {
			Rf[k] = Ecur[k].x * Ecur[k].x + Ecur[k].y * Ecur[k].y;
			Yf[k] = Ycur[k].x * Ycur[k].x + Ycur[k].y * Ycur[k].y;
			Xf[k] = X0[k].x * X0[k].x + X0[k].y * X0[k].y;
		}
	}
	// the Nyquist powers live in lane 0; everyone needs them for the ordered sums below
	Rf_F = rdlane(Rf_F, 0);
	Yf_F = rdlane(Yf_F, 0);
	Xf_F = rdlane(Xf_F, 0);

	// ---- sanity checks
	bool zero_out = false;
	if (!(Syy >= 0 && Sxx >= 0 && See >= 0) || !(Sff < N * 1e9 && Syy < N * 1e9 && Sxx < N * 1e9)) {
		sc.screwed_up += 50;
		zero_out = true;
	} else if (Sff > Sdd + (float)(N * 10000)) {
		sc.screwed_up++;
	} else {
		sc.screwed_up = 0;
	}
	if (zero_out) {
#pragma unroll
		for (int k = 0; k < K; ++k) out_i[k] = 0;
	}
	int16_t *op = a.out + (size_t)s * a.stride + e0;
	if (sc.screwed_up >= 50) { // speex_echo_state_reset
		float z[K];
		float2 z2[K];
		float one[K];
#pragma unroll
		for (int k = 0; k < K; ++k) z[k] = 0.f, z2[k] = make_float2(0, 0), one[k] = 1.0f;
		for (int j = 0; j < M; ++j) {
			store_bins<K>(Ws + (size_t)j * F + e0, z2);
			store_bins<K>(FGs + (size_t)j * F + e0, z2);
		}
		for (int j = 0; j <= M; ++j) store_bins<K>(Xs + (size_t)j * F + e0, z2);
		store_vec<K>(sm + SL::POWER + e0, z);
		store_vec<K>(sm + SL::POWER1 + e0, one);
		store_vec<K>(sm + SL::EH + e0, z);
		store_vec<K>(sm + SL::YH + e0, z);
		if (lane == 0) {
			sm[SL::POWER + F] = 0;
			sm[SL::POWER1 + F] = 1.0f;
			sm[SL::EH + F] = 0;
			sm[SL::YH + F] = 0;
		}
		store_vec<K>(sm + SL::LASTY + e0, z);
		store_vec<K>(sm + SL::LASTY + F + e0, z);
		store_bins<K>(reinterpret_cast<float2 *>(sm + SL::E) + e0, z2);
		store_vec<K>(sm + SL::XPREV + e0, z);
		if (lane < M) sm[SL::WNORM + lane] = 0;
		AecScalars zc = sc;
		zc.cancel_count = 0;
		zc.screwed_up = 0;
		zc.notch0 = zc.notch1 = 0;
		zc.memD = zc.memE = zc.memX = 0;
		zc.saturated = 0;
		zc.adapted = 0;
		zc.sum_adapt = 0;
		zc.Pey = zc.Pyy = 1.0f;
		zc.Davg1 = zc.Davg2 = zc.Dvar1 = zc.Dvar2 = 0;
		if (lane == 0) a.scal[s] = zc;
#pragma unroll
		for (int k = 0; k < K; ++k) op[k] = (int16_t)out_i[k];
		return;
	}
	if (See < (float)(N * 100)) See = (float)(N * 100);
	Sxx += Sxx; // sic: the library accumulates the far-end energy a second time here

	// ---- far-end power, leak estimate
	float pw[K], pw_F;
	load_vec<K>(sm + SL::POWER + e0, pw);
#pragma unroll
	for (int k = 0; k < K; ++k) pw[k] = a.ss_1 * pw[k] + 1 + a.ss * Xf[k];
	store_vec<K>(sm + SL::POWER + e0, pw);
	pw_F = sm[SL::POWER + F];
	pw_F = a.ss_1 * pw_F + 1 + a.ss * Xf_F;
	float Ehd[K], Yhd[K], Ehd_F, Yhd_F;
	{
		float eh[K], yh[K];
		load_vec<K>(sm + SL::EH + e0, eh);
		load_vec<K>(sm + SL::YH + e0, yh);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			Ehd[k] = Rf[k] - eh[k];
			Yhd[k] = Yf[k] - yh[k];
			eh[k] = (1 - a.spec_average) * eh[k] + a.spec_average * Rf[k];
			yh[k] = (1 - a.spec_average) * yh[k] + a.spec_average * Yf[k];
		}
		store_vec<K>(sm + SL::EH + e0, eh);
		store_vec<K>(sm + SL::YH + e0, yh);
		const float ehF = sm[SL::EH + F], yhF = sm[SL::YH + F];
		Ehd_F = Rf_F - ehF;
		Yhd_F = Yf_F - yhF;
		if (lane == 0) {
			sm[SL::POWER + F] = pw_F;
			sm[SL::EH + F] = (1 - a.spec_average) * ehF + a.spec_average * Rf_F;
			sm[SL::YH + F] = (1 - a.spec_average) * yhF + a.spec_average * Yf_F;
		}
	}
	float Pey = 1.0f, Pyy = 1.0f;
	Pey = Pey + Ehd_F * Yhd_F;
	Pyy = Pyy + Yhd_F * Yhd_F;
	Pey = WSeq<K>::dot_desc(Pey, Ehd, Yhd);
	Pyy = WSeq<K>::dot_desc(Pyy, Yhd, Yhd);
	Pyy = sqrt_via_double(Pyy);
	Pey = Pey / Pyy;
	float tmp32 = a.beta0 * Syy;
	if (tmp32 > a.beta_max * See) tmp32 = a.beta_max * See;
	const float alpha = tmp32 / See;
	const float alpha_1 = 1.0f - alpha;
	sc.Pey = alpha_1 * sc.Pey + alpha * Pey;
	sc.Pyy = alpha_1 * sc.Pyy + alpha * Pyy;
	if (sc.Pyy < 1.0f) sc.Pyy = 1.0f;
	if (sc.Pey < .005f * sc.Pyy) sc.Pey = .005f * sc.Pyy;
	if (sc.Pey > sc.Pyy) sc.Pey = sc.Pyy;
	sc.leak_estimate = sc.Pey / sc.Pyy;
	float RER = (float)((.0001 * Sxx + 3. * (sc.leak_estimate * Syy)) / See);
	if (RER < Sey * Sey / (1 + See * Syy)) RER = Sey * Sey / (1 + See * Syy);
	if (RER > .5) RER = .5;
	if (!sc.adapted && sc.sum_adapt > (float)M && sc.leak_estimate * Syy > .03f * Syy) sc.adapted = 1;

	auto step = [&](float Yfv, float Rfv, float pwv) -> float {
		float r = sc.leak_estimate * Yfv;
		const float e = Rfv + 1;
		if (r > .5 * e) r = (float)(.5 * e);
		r = .7f * r + .3f * (float)(RER * e);
		return r / (e * (pwv + 10));
	};
	float np1[K];
	if (sc.adapted) {
#pragma unroll
		for (int k = 0; k < K; ++k) np1[k] = step(Yf[k], Rf[k], pw[k]);
		if (lane == 0) sm[SL::POWER1 + F] = step(Yf_F, Rf_F, pw_F);
	} else {
		float adapt_rate = 0;
		if (Sxx > (float)(N * 1000)) {
			tmp32 = .25f * Sxx;
			if (tmp32 > .25 * See) tmp32 = (float)(.25 * See);
			adapt_rate = tmp32 / See;
		}
#pragma unroll
		for (int k = 0; k < K; ++k) np1[k] = adapt_rate / (pw[k] + 10);
		if (lane == 0) sm[SL::POWER1 + F] = adapt_rate / (pw_F + 10);
		sc.sum_adapt = sc.sum_adapt + adapt_rate;
	}
	store_vec<K>(sm + SL::POWER1 + e0, np1);

	// ---- last_y for the residual-echo estimate
	{
		float lo[K], ln[K];
		load_vec<K>(sm + SL::LASTY + F + e0, lo);
#pragma unroll
		for (int k = 0; k < K; ++k) ln[k] = sc.adapted ? (float)(mic_i[k] - out_i[k]) : lo[k];
		store_vec<K>(sm + SL::LASTY + e0, lo);
		store_vec<K>(sm + SL::LASTY + F + e0, ln);
	}
#pragma unroll
	for (int k = 0; k < K; ++k) op[k] = (int16_t)out_i[k];
	if (lane == 0) a.scal[s] = sc;
}

// ===================================================================== post-filter, one frame
template <int F>
__global__ __launch_bounds__(64) void aec_post_wave_kernel(AecArgs a) {
	__shared__ WLds<F> L;
	using SL = SmallLayout<F>;
	constexpr int K = F / 64;
	const int s = a.first + blockIdx.x;
	if (a.run && !a.run[s]) return;
	const int lane = threadIdx.x;
	const int e0 = lane * K;
	float *sm = a.small + (size_t)s * a.small_stride;
	AecScalars sc = a.scal[s];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.tw[e0 + k] = a.t.tw[e0 + k];
		L.super[e0 + k] = a.t.super[e0 + k];
		L.perm[e0 + k] = a.t.perm[e0 + k];
	}
	sc.nb_adapt++;
	if (sc.nb_adapt > 20000) sc.nb_adapt = 20000;
	sc.min_count++;
	float beta = 1.0f / sc.nb_adapt;
	if (beta < .03f) beta = .03f;
	const float beta_1 = 1.0f - beta;

	// residual echo spectrum (speex_echo_get_residual)
	{
		float lo[K], ln[K], h0[K], h1[K];
		load_vec<K>(sm + SL::LASTY + e0, lo);
		load_vec<K>(sm + SL::LASTY + F + e0, ln);
		load_vec<K>(a.t.hann + e0, h0);
		load_vec<K>(a.t.hann + F + e0, h1);
#pragma unroll
		for (int k = 0; k < K; ++k) lo[k] = h0[k] * lo[k], ln[k] = h1[k] * ln[k];
		store_vec<K>(L.tbuf + e0, lo);
		store_vec<K>(L.tbuf + F + e0, ln);
	}
	float2 Yr[K];
	w_rfft_forward<F>(L, Yr);
	const float leak2 = (sc.leak_estimate > .5) ? 1.f : 2 * sc.leak_estimate;
	float res[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const float r = (e0 + k == 0) ? Yr[k].x * Yr[k].x : Yr[k].x * Yr[k].x + Yr[k].y * Yr[k].y;
		res[k] = (float)(int32_t)(leak2 * r);
	}
	const float res0 = rdlane(res[0], 0);
	const bool bad = !(res0 >= 0 && res0 < F * 1e9f);
	float en[K], wl[K], wr[K];
	load_vec<K>(sm + SL::ECHON + e0, en);
	load_vec<K>(a.t.bfl + e0, wl);
	load_vec<K>(a.t.bfr + e0, wr);
	float *pl = L.spec, *pr = L.spec + F;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const float rr = bad ? 0.f : res[k];
		const float c = .6f * en[k];
		en[k] = c > rr ? c : rr;
		pl[e0 + k] = wl[k] * en[k];
		pr[e0 + k] = wr[k] * en[k];
	}
	store_vec<K>(sm + SL::ECHON + e0, en);
	// analysis frame [inbuf, x] * window
	int16_t *op = a.out + (size_t)s * a.stride + e0;
	float xcur[K];
	{
		float inb[K], w0[K], w1[K];
		load_vec<K>(sm + SL::INBUF + e0, inb);
		load_vec<K>(a.t.pwin + e0, w0);
		load_vec<K>(a.t.pwin + F + e0, w1);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			xcur[k] = (float)op[k];
			inb[k] = inb[k] * w0[k];
			w1[k] = xcur[k] * w1[k];
		}
		store_vec<K>(sm + SL::INBUF + e0, xcur);
		store_vec<K>(L.tbuf + e0, inb);
		store_vec<K>(L.tbuf + F + e0, w1);
	}
	WSYNC();
	float *bandv = L.band;
	if (lane < NB_BANDS) bandv[lane] = band_sum<F>(a.t, lane, pl, pr);
	float2 ft[K];
	w_rfft_forward<F>(L, ft);
	float ps[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		ps[k] = (e0 + k == 0) ? ft[k].x * ft[k].x : ft[k].x * ft[k].x + ft[k].y * ft[k].y;
		L.vec[e0 + k] = ps[k];
		pl[e0 + k] = wl[k] * ps[k];
		pr[e0 + k] = wr[k] * ps[k];
	}
	WSYNC();
	if (lane < NB_BANDS) bandv[NB_BANDS + lane] = band_sum<F>(a.t, lane, pl, pr);
	// update_noise_prob
	float S[K], Smin[K], Stmp[K], noise[K];
	load_vec<K>(sm + SL::S_ + e0, S);
	load_vec<K>(sm + SL::SMIN + e0, Smin);
	load_vec<K>(sm + SL::STMP + e0, Stmp);
	load_vec<K>(sm + SL::NOISE + e0, noise);
	int min_range;
	if (sc.nb_adapt < 100) min_range = 15;
	else if (sc.nb_adapt < 1000) min_range = 50;
	else if (sc.nb_adapt < 10000) min_range = 150;
	else min_range = 300;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int b = e0 + k;
		if (b == 0 || b == F - 1) S[k] = .8f * S[k] + .2f * ps[k];
		else S[k] = .8f * S[k] + .05f * L.vec[b - 1] + .1f * ps[k] + .05f * L.vec[b + 1];
		if (sc.nb_adapt == 1) Smin[k] = Stmp[k] = 0;
		if (sc.min_count > min_range) {
			Smin[k] = Stmp[k] < S[k] ? Stmp[k] : S[k];
			Stmp[k] = S[k];
		} else {
			Smin[k] = Smin[k] < S[k] ? Smin[k] : S[k];
			Stmp[k] = Stmp[k] < S[k] ? Stmp[k] : S[k];
		}
		const int update_prob = (.4f * S[k] > Smin[k]) ? 1 : 0;
		if (!update_prob || ps[k] < noise[k]) {
			const float v = beta_1 * noise[k] + beta * ps[k];
			noise[k] = v > 0 ? v : 0;
		}
	}
	if (sc.min_count > min_range) sc.min_count = 0;
	store_vec<K>(sm + SL::S_ + e0, S);
	store_vec<K>(sm + SL::SMIN + e0, Smin);
	store_vec<K>(sm + SL::STMP + e0, Stmp);
	store_vec<K>(sm + SL::NOISE + e0, noise);
	WSYNC();
#pragma unroll
	for (int k = 0; k < K; ++k) {
		pl[e0 + k] = wl[k] * noise[k];
		pr[e0 + k] = wr[k] * noise[k];
	}
	WSYNC();
	if (lane < NB_BANDS) bandv[2 * NB_BANDS + lane] = band_sum<F>(a.t, lane, pl, pr);
	WSYNC();

	auto snr = [&](float psv, float noisev, float echov, float oldps, float &post, float &prior) {
		const float tot_noise = 1.f + noisev + echov + 0.f;
		post = psv / tot_noise - 1.f;
		if (post > 100.f) post = 100.f;
		const float t = oldps / (oldps + tot_noise);
		const float gamma = .1f + .89f * (t * t);
		prior = gamma * (post > 0 ? post : 0) + (1.0f - gamma) * (oldps / tot_noise);
		if (prior > 100.f) prior = 100.f;
	};
	float old_ps[K], post[K], prior[K];
	load_vec<K>(sm + SL::OLDPS + e0, old_ps);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (sc.nb_adapt == 1) old_ps[k] = ps[k];
		snr(ps[k], noise[k], en[k], old_ps[k], post[k], prior[k]);
		L.vec[e0 + k] = prior[k];
	}
	float old_ps_b = 0, post_b = 0, prior_b = 0, ps_b = 0;
	if (lane < NB_BANDS) {
		ps_b = bandv[NB_BANDS + lane];
		old_ps_b = sm[SL::OLDPS + F + lane];
		if (sc.nb_adapt == 1) old_ps_b = ps_b;
		snr(ps_b, bandv[2 * NB_BANDS + lane], bandv[lane], old_ps_b, post_b, prior_b);
	}
	WSYNC();
	float zeta[K];
	load_vec<K>(sm + SL::ZETA + e0, zeta);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int b = e0 + k;
		if (b == 0 || b >= F - 1) zeta[k] = .7f * zeta[k] + .3f * prior[k];
		else zeta[k] = .7f * zeta[k] + .15f * prior[k] + .075f * L.vec[b - 1] + .075f * L.vec[b + 1];
	}
	store_vec<K>(sm + SL::ZETA + e0, zeta);
	float zeta_b = 0;
	if (lane < NB_BANDS) {
		zeta_b = .7f * sm[SL::ZETA + F + lane] + .3f * prior_b;
		sm[SL::ZETA + F + lane] = zeta_b;
	}
	float Zframe = 0;
#pragma unroll
	for (int i = 0; i < NB_BANDS; ++i) Zframe = Zframe + rdlane(zeta_b, i);
	const float Pframe = .1f + .899f * qcurve(Zframe / NB_BANDS);
	const int eff_echo = (int)((1.0f - Pframe) * -40 + Pframe * -15);
	if (lane < NB_BANDS) {
		const float noise_floor = (float)exp((double)(.2302585f * -15));
		const float echo_floor = (float)exp((double)(.2302585f * eff_echo));
		const float nb = bandv[2 * NB_BANDS + lane], eb = bandv[lane];
		const float gfloor = (float)(sqrt((double)(noise_floor * nb + echo_floor * eb)) / sqrt((double)(1 + nb + eb)));
		const float prior_ratio = prior_b / (prior_b + 1.f);
		const float theta = prior_ratio * (1.f + post_b);
		const float MM = hypergeom_gain(theta);
		float g = prior_ratio * MM;
		if (g > 1.f) g = 1.f;
		old_ps_b = .2f * old_ps_b + (.8f * (g * g)) * ps_b;
		sm[SL::OLDPS + F + lane] = old_ps_b;
		const float P1 = .199f + .8f * qcurve(zeta_b);
		const float q = 1.0f - Pframe * P1;
		const float g2 = (float)(1 / (1.f + (q / (1.f - q)) * (1 + prior_b) * exp((double)(-theta))));
		bandv[lane] = g2;
		bandv[NB_BANDS + lane] = g;
		bandv[2 * NB_BANDS + lane] = gfloor;
	}
	WSYNC();
	float gain2[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int bl = a.t.bleft[e0 + k], br = bl + 1;
		auto psd = [&](const float *mel) -> float {
			float t = mel[bl] * wl[k];
			t += mel[br] * wr[k];
			return t;
		};
		const float p = psd(bandv);
		const float gain_bark = psd(bandv + NB_BANDS);
		const float gfl = psd(bandv + 2 * NB_BANDS);
		const float prior_ratio = prior[k] / (prior[k] + 1.f);
		const float theta = prior_ratio * (1.f + post[k]);
		const float MM = hypergeom_gain(theta);
		float g = prior_ratio * MM;
		if (g > 1.f) g = 1.f;
		if (.333f * g > gain_bark) g = 3 * gain_bark;
		float gain = g;
		old_ps[k] = .2f * old_ps[k] + (.8f * (gain * gain)) * ps[k];
		if (gain < gfl) gain = gfl;
		const float tmp = p * sqrt_via_double(gain) + (1.0f - p) * sqrt_via_double(gfl);
		gain2[k] = tmp * tmp;
	}
	store_vec<K>(sm + SL::OLDPS + e0, old_ps);
	const float g_last = rdlane(gain2[K - 1], 63); // gain2[F-1] scales the Nyquist term
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (e0 + k == 0) {
			ft[k].x = gain2[k] * ft[k].x;
			ft[k].y = g_last * ft[k].y;
		} else {
			ft[k].x = gain2[k] * ft[k].x;
			ft[k].y = gain2[k] * ft[k].y;
		}
	}
	w_rfft_inverse<F>(L, ft);
	{
		float lo[K], hi[K], w0[K], w1[K], ob[K];
		load_vec<K>(L.tbuf + e0, lo);
		load_vec<K>(L.tbuf + F + e0, hi);
		load_vec<K>(a.t.pwin + e0, w0);
		load_vec<K>(a.t.pwin + F + e0, w1);
		load_vec<K>(sm + SL::OUTBUF + e0, ob);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			op[k] = word2int(ob[k] + lo[k] * w0[k]);
			hi[k] = hi[k] * w1[k];
		}
		store_vec<K>(sm + SL::OUTBUF + e0, hi);
	}
	if (lane == 0) {
		a.scal[s].nb_adapt = sc.nb_adapt;
		a.scal[s].min_count = sc.min_count;
	}
}
