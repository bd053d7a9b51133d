// aec_group.hpp -- the canceller at the SMALL frame sizes (8 kHz: F = 64, 16 kHz: F = 128; speexec.c:171-180), several legs
// per wavefront (included by aec.hip after aec_tick.hpp, whose addressing helpers and layout it shares).
//
// One wavefront per leg (aec_tick.hpp) is built for F = 256: four bins per lane.  At F = 64 a lane holds ONE bin, three
// quarters of the lanes idle through every radix-4 stage of the 13 transforms of a frame, and every one of the library's
// serial loops -- ~30 chains of 64 dependent steps -- advances one element per step: the kernel is instruction-bound at
// 2.6 TB/s (F = 64) / 3.1 TB/s (F = 128) where the memory system gives 5.  Here a leg owns G = F / 4 lanes -- FOUR bins per
// lane, as at F = 256 -- and a wavefront serves 64 / G legs: four at 8 kHz, two at 16 kHz.
//   * the transforms: F / 4 butterflies per radix-4 stage = G lanes, every lane busy; the deepest stage in registers;
//   * the serial chains: G steps of four elements, stepped for all the wave's legs at once (row_shr inside a row of 16 lanes
//     at G = 16; wave_shr with the chain re-started at lane 32 at G = 32);
//   * what was a scalar per leg (the two-path control, the adaptation flags, ring heads) is a register per lane, equal in a
//     leg's lanes; what was a uniform branch is a predicate -- except the AUMDF constraint, whose transforms must run with
//     all lanes: it runs for the whole wave whenever ANY of its legs asks (block 0 for all, the round-robin block per leg);
//   * every per-leg array is addressed through ONE buffer descriptor per array for the wave's legs + a per-lane offset; a
//     leg that has no frame in this launch (gated, or beyond the batch) gets an offset beyond the descriptor: its loads
//     read zeros, its stores are dropped, it computes on zeros and nobody sees it.
// ONE frame per launch (mi_aec_process; mi_aec_process_frames runs the tick's frames as one launch each): the two-frame
// machinery of the tick form (speculated foreground response, unwritten W1) needs the registers that the per-leg scalars
// now take.  Arithmetic, operation order and the state in HBM are those of aec_tick_kernel<F>: the same sums in the
// library's order, the same trees (a balanced tree over the bins in natural order is the same tree whichever lanes hold
// them), so a leg may change between the two kernels from one launch to the next -- the FIFO entries stay on the tick form
// -- and tests/test_gpu_aec.py holds both to the oracle.

template <int F>
struct GroupShape {
	static constexpr int K = 4, G = F / K, LPW = 64 / G; // bins per lane, lanes per leg, legs per wavefront
};

template <int F>
struct alignas(16) GLdsLeg {  // a leg's share of the wave's LDS
	float2 zbuf[F];           // complex FFT work
	float tbuf[2 * F];        // time-domain exchange / inverse-transform staging
	float spec[2 * F];        // bin-interleaved spectrum exchange, filterbank products
	float prop[64], wnorm[64], fgnorm[64];
	float vec[F];             // per-bin exchange (neighbour access in the post-filter)
	float band[4 * NB_BANDS + 8];
};
template <int F>
struct alignas(16) GLds {
	float2 tw[F], super[F];   // the tables, once per wave
	uint16_t perm[F];
	GLdsLeg<F> leg[GroupShape<F>::LPW];
};

template <int F>
struct Grp { // the leg's lanes inside the wave
	static constexpr int K = GroupShape<F>::K, G = GroupShape<F>::G, LPW = GroupShape<F>::LPW;
	__device__ static __forceinline__ int lane() { return threadIdx.x & (G - 1); }
	__device__ static __forceinline__ int index() { return threadIdx.x / G; }
	// the value lane `from` of the leg holds, in all of the leg's lanes
	__device__ static __forceinline__ float from_lane(float v, int from) {
		return __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x & ~(G - 1)) + from) << 2, __float_as_int(v)));
	}
	// ... of its first / last lane: one DPP move inside a row of 16 (row_newbcast), two v_readlane and a select for a leg of two rows
	template <int WHICH>
	__device__ static __forceinline__ float edge(float v) {
		if constexpr (G == 16) {
			return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), WHICH ? 0x15F : 0x150, 0xf, 0xf, false));
		} else {
			const float a = rdlane(v, WHICH ? G - 1 : 0), b = rdlane(v, G + (WHICH ? G - 1 : 0));
			return threadIdx.x < G ? a : b;
		}
	}
	__device__ static __forceinline__ float first(float v) { return edge<0>(v); }
	__device__ static __forceinline__ float last(float v) { return edge<1>(v); }
	// lane l <- lane l - 1 of the leg, its first lane <- `head`
	__device__ static __forceinline__ float shr1(float head, float v) {
		if constexpr (G == 16) {
			return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(head), __float_as_int(v), 0x111, 0xf, 0xf, false)); // row_shr:1
		} else {
			const float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(head), __float_as_int(v), 0x138, 0xf, 0xf, false)); // wave_shr:1
			return (threadIdx.x & (G - 1)) == 0 ? head : t; // (lane 32 took lane 31's: the second leg's chain starts here)
		}
	}
	// lane l <- lane l + 1 of the leg, its last lane <- `tail`
	__device__ static __forceinline__ float shl1(float tail, float v) {
		if constexpr (G == 16) {
			return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(tail), __float_as_int(v), 0x101, 0xf, 0xf, false)); // row_shl:1
		} else {
			const float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(tail), __float_as_int(v), 0x130, 0xf, 0xf, false)); // wave_shl:1
			return (threadIdx.x & (G - 1)) == G - 1 ? tail : t;
		}
	}
	// sum / maximum over the leg's lanes in the butterfly order of wave_tree: inside the rows of 16, then across the leg's rows
	template <typename Op>
	__device__ static __forceinline__ float tree(float v, Op op) {
		auto dpp = [](float x, auto ctrl) {
			return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
		};
		v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));  // quad_perm [1,0,3,2]
		v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));  // quad_perm [2,3,0,1]
		v = op(v, dpp(v, std::integral_constant<int, 0x141>{})); // row_half_mirror
		v = op(v, dpp(v, std::integral_constant<int, 0x140>{})); // row_mirror: every lane of a row holds the row's result
		if constexpr (G == 32) {
			const float o = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x ^ 16) << 2, __float_as_int(v)));
			v = (threadIdx.x & 16) ? op(o, v) : op(v, o); // (row 0 of the leg is the left operand in both rows: the same bits)
		}
		return v;
	}
	__device__ static __forceinline__ bool any(bool p) {
		const unsigned long long b = __ballot(p);
		const unsigned long long m = (G == 32 ? 0xffffffffull : 0xffffull) << (threadIdx.x & ~(G - 1));
		return (b & m) != 0;
	}
};

// ---- the transforms, a leg in G lanes (same butterflies, twiddles and operation order as w_cfft / w_rfft_*)
template <int F, typename LL, typename LT>
__device__ void g_cfft(LL &L, const LT &T, const float2 *src, bool inverse) {
	using GP = Grp<F>;
	constexpr int K = GP::K, G = GP::G;
	const int lane = GP::lane();
	float2 val[K];
#pragma unroll
	for (int k = 0; k < K; ++k) val[k] = src[T.perm[lane * K + k]];
	// the deepest stage (m = 1) works on p consecutive gathered elements: the lane's own K = 4 (one radix-4 butterfly at
	// F = 64, two radix-2 ones at F = 128) -- in registers, twiddle tw[0]
	{
		float2 w0 = T.tw[0];
		if (inverse) w0.y = -w0.y;
		if constexpr (plan_p(F, 0) == 2) {
#pragma unroll
			for (int h = 0; h < K; h += 2) {
				const float2 t = cmulf(val[h + 1], w0);
				const float2 a = val[h];
				val[h + 1] = make_float2(a.x - t.x, a.y - t.y);
				val[h] = make_float2(a.x + t.x, a.y + t.y);
			}
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
	for (int s = 1; s < plan_n(F); ++s) {
		constexpr int FF = F;
		const int p = plan_p(FF, s), m = plan_m(FF, s), fs = plan_fs(FF, s); // (p == 4 from here on: F / 4 = G butterflies, one per lane)
		static_assert(plan_p(F, 1) == 4 && F / 4 == G, "one radix-4 butterfly per lane and stage");
		const int i = lane / m, j = lane - i * m;
		float2 *Fo = L.zbuf + i * (p * m) + j;
		float2 w1 = T.tw[j * fs], w2 = T.tw[j * fs * 2], w3 = T.tw[j * fs * 3];
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
		WSYNC();
	}
}

// L.tbuf (2F time samples; or `src`) -> this lane's K bins, scaled 1/N.  Bin 0 = (DC, Nyquist).
template <int F, typename LL, typename LT>
__device__ void g_rfft_forward(LL &L, const LT &T, float2 (&out)[4], const float *src = nullptr) {
	using GP = Grp<F>;
	constexpr int K = GP::K;
	if (src == nullptr) src = L.tbuf;
	const int lane = GP::lane();
	WSYNC();
	g_cfft<F>(L, T, reinterpret_cast<const float2 *>(src), false);
	const float scale = 1.f / (2 * F);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int b = lane * K + k;
		if (b == 0) {
			const float2 t0 = L.zbuf[0];
			out[k] = make_float2((t0.x + t0.y) * scale, (t0.x - t0.y) * scale);
		} else {
			const bool upper = b >= F - b;
			const int kk = upper ? F - b : b;
			const float2 a = L.zbuf[kk], c = L.zbuf[F - kk];
			const float2 sw = T.super[kk];
			const v2f av = {a.x, a.y}, cv = {c.x, c.y};
			v2f f1, f2;
			asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(f1) : "v"(av), "v"(cv));
			asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(f2) : "v"(av), "v"(cv));
			v2f tw = pk_cmul(f2, (v2f){sw.x, sw.y});
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
template <int F, typename LL, typename LT>
__device__ void g_rfft_inverse(LL &L, const LT &T, const float2 (&in)[4]) {
	using GP = Grp<F>;
	constexpr int K = GP::K;
	const int lane = GP::lane();
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
			const v2f fk = {L.spec[2 * kk], L.spec[2 * kk + 1]};
			const v2f s2 = {L.spec[2 * (F - kk)], L.spec[2 * (F - kk) + 1]};
			const float2 sw = T.super[kk];
			const v2f sv = {sw.x, sw.y};
			v2f fek, d, p, q, fok;
			asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(fek) : "v"(fk), "v"(s2));
			asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(fk), "v"(s2));
			asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(d), "v"(sv));
			asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(q) : "v"(d), "v"(sv));
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
	g_cfft<F>(L, T, tmp, true);
}

// ---- the library's serial loops over a frame, element index = lane * 4 + k inside the leg, as systolic chains of G steps
// (WSeq<4> of aec_wave.hpp with the leg's shifts)
template <int F>
struct GSeq {
	using GP = Grp<F>;
	static constexpr int K = 4, KP = 2, G = GP::G;
	__device__ static void pairs(const float (&x)[K], const float (&y)[K], float (&part)[KP]) {
#pragma unroll
		for (int k = 0; k < K; k += 2) {
			float p = 0;
			p = p + x[k] * y[k];
			p = p + x[k + 1] * y[k + 1];
			part[k / 2] = p;
		}
	}
	template <int NC>
	__device__ static void chain_up(const float (&part)[NC][KP], float (&sum)[NC]) {
		float s[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) s[c] = 0;
#pragma unroll 4
		for (int l = 0; l < G; ++l) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				s[c] = GP::shr1(0.f, s[c]) + part[c][0];
#pragma unroll
				for (int k = 1; k < KP; ++k) s[c] = s[c] + part[c][k];
			}
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) sum[c] = GP::last(s[c]);
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
	// two descending dot products at once: acc = init; for l = G-1..0: for k = K-1..0: acc += a[k] b[k] of lane l
	__device__ static void dot_desc2(float init0, const float (&a0)[K], const float (&b0)[K], float init1, const float (&a1)[K],
	                                 const float (&b1)[K], float &r0, float &r1) {
		float p0[K], p1[K];
#pragma unroll
		for (int k = 0; k < K; ++k) p0[k] = a0[k] * b0[k], p1[k] = a1[k] * b1[k];
		float s0 = init0, s1 = init1;
#pragma unroll 4
		for (int l = 0; l < G; ++l) {
			v2f t = {GP::shl1(init0, s0), GP::shl1(init1, s1)};
#pragma unroll
			for (int k = K - 1; k >= 0; --k) t = t + (v2f){p0[k], p1[k]};
			s0 = t.x, s1 = t.y;
		}
		r0 = GP::first(s0), r1 = GP::first(s1);
	}
	// filter_dc_notch16 (w_dc_notch): (m0, m1) in and out per leg
	__device__ static void dc_notch(const float (&in)[K], float radius, float den2, float &m0io, float &m1io, float (&out)[K]) {
		const float i0 = m0io, i1 = m1io;
		float m0 = i0, m1 = i1;
		const v2f rc = {radius, den2};
		v2f pr[2] = {{in[0], in[1]}, {in[2], in[3]}};
#pragma unroll 2
		for (int l = 0; l < G; ++l) {
			float a0 = GP::shr1(i0, m0), a1 = GP::shr1(i1, m1);
			notch_sample<false>(rc, pr[0], in[0], a0, a1, out[0]);
			notch_sample<true>(rc, pr[0], in[1], a0, a1, out[1]);
			notch_sample<false>(rc, pr[1], in[2], a0, a1, out[2]);
			notch_sample<true>(rc, pr[1], in[3], a0, a1, out[3]);
			m0 = a0, m1 = a1;
		}
		m0io = GP::last(m0);
		m1io = GP::last(m1);
	}
	// de-emphasis of the output: t = d + 0.9 mem; mem = t
	__device__ static void deemphasis(const float (&d)[K], float &memio, float (&out)[K]) {
		const float init = memio;
		float m = init;
#pragma unroll 2
		for (int l = 0; l < G; ++l) {
			float a = GP::shr1(init, m);
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const float t = d[k] + .9f * a;
				a = t;
				out[k] = t;
			}
			m = a;
		}
		memio = GP::last(m);
	}
};

// the per-leg scalar record through the wave's descriptor: 28 dwords, the same in all of a leg's lanes
__device__ __forceinline__ AecScalars g_load_scalars(rsrc_t r, unsigned off) {
	AecScalars sc;
	unsigned *w = reinterpret_cast<unsigned *>(&sc);
#pragma unroll
	for (int i = 0; i < 7; ++i) {
		const u4v t = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u * i, 0, 0);
		w[4 * i] = t.x, w[4 * i + 1] = t.y, w[4 * i + 2] = t.z, w[4 * i + 3] = t.w;
	}
	return sc;
}
__device__ __forceinline__ void g_store_scalars(rsrc_t r, unsigned off, const AecScalars &sc) {
	const unsigned *w = reinterpret_cast<const unsigned *>(&sc);
#pragma unroll
	for (int i = 0; i < 7; ++i) {
		u4v t = {w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
		__builtin_amdgcn_raw_buffer_store_b128(t, r, off + 16u * i, 0, 0);
	}
}
__device__ __forceinline__ float g_load_f(rsrc_t r, unsigned off) { return u2f(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void g_store_f(rsrc_t r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(f2u(v), r, off, 0, 0); }

// One frame of every leg that has one (rows: mi_aec_process / one round of mi_aec_process_frames).  a.first: first leg of
// the launch; frame: which of the leg's rows' frames (a.count: the leg runs if it has more than `frame`).
template <int F>
__global__ __launch_bounds__(64, 2) void aec_group_kernel(AecArgs a, int frame) {
	using GP = Grp<F>;
	using SL = TickLayout<F>;
	constexpr int N = 2 * F, K = GP::K, G = GP::G, LPW = GP::LPW;
	__shared__ GLds<F> LW;
	const int lane = GP::lane(), grp = GP::index();
	auto &L = LW.leg[grp];
	const int e0 = lane * K;
	const int s0 = a.first + (int)blockIdx.x * LPW, s = s0 + grp;
	const int M = a.M;
	// ---- does this leg have a frame?  (a leg without one is parked beyond every descriptor: see the head of the file)
	bool live = s < a.nstreams;
	if (live) {
		if (a.count) live = (int)a.count[s] > frame;
		else if (a.run) live = a.run[s] != 0;
	}
	if (!__any(live)) return;
	const int nleg = min(LPW, a.nstreams - s0); // legs of the batch this wave covers
	const unsigned DEAD = 0x40000000u;
	const unsigned row_b = (unsigned)a.stride * 2u, small_b = (unsigned)a.small_stride * 4u;
	const unsigned x_b = (unsigned)((M + 1) * N) * 4u, wf_b = (unsigned)(2 * M * N) * 4u, HALF = (unsigned)(M * N) * 4u;
	const rsrc_t rMic = mk_rsrc(a.mic + (size_t)s0 * a.stride, (unsigned)nleg * row_b);
	const rsrc_t rRef = mk_rsrc(a.ref + (size_t)s0 * a.stride, (unsigned)nleg * row_b);
	const rsrc_t rOut = mk_rsrc(a.out + (size_t)s0 * a.stride, (unsigned)nleg * row_b);
	const rsrc_t rS = mk_rsrc(a.small + (size_t)s0 * a.small_stride, (unsigned)nleg * small_b);
	const rsrc_t rX = mk_rsrc(a.X + (size_t)s0 * (M + 1) * N, (unsigned)nleg * x_b);
	const rsrc_t rWF = mk_rsrc(a.WF + (size_t)s0 * 2 * M * N, (unsigned)nleg * wf_b);
	const rsrc_t rSc = mk_rsrc(a.scal + s0, (unsigned)nleg * (unsigned)sizeof(AecScalars));
	const unsigned g = (unsigned)grp;
	const unsigned oRow = live ? g * row_b + (unsigned)(frame * F + e0) * 2u : DEAD; // this lane's K samples of the frame
	const unsigned oS = live ? g * small_b : DEAD;                                    // + float index * 4
	const unsigned vb4 = oS + (unsigned)e0 * 4u;                                      // this lane's K floats of a per-bin array
	const unsigned vb8 = oS + (unsigned)e0 * 8u;                                      // ... of a bin-interleaved spectrum
	const unsigned oX = live ? g * x_b + (unsigned)e0 * 8u : DEAD;
	const unsigned oW = live ? g * wf_b + (unsigned)e0 * 8u : DEAD;
	const unsigned oSc = live ? g * (unsigned)sizeof(AecScalars) : DEAD;
	AecScalars sc = g_load_scalars(rSc, oSc);
	if (!live) sc.xhead = 0, sc.cancel_count = 0; // (zeros anyway; the indices derived from them stay inside their rings)
	unsigned wo = sc.wsel ? HALF : 0u, fo = HALF - wo; // byte offset of the background / foreground half

	// ---- tables (once per wave), per-block step / norm
	for (int i = threadIdx.x; i < F; i += 64) {
		LW.tw[i] = a.t.tw[i];
		LW.super[i] = a.t.super[i];
		LW.perm[i] = a.t.perm[i];
	}
	for (int j = lane; j < M; j += G) {
		L.prop[j] = g_load_f(rS, oS + (unsigned)(SL::PROP + j) * 4u);
		L.wnorm[j] = g_load_f(rS, oS + (unsigned)(SL::WNORM + j) * 4u);
		L.fgnorm[j] = g_load_f(rS, oS + (unsigned)(SL::FGNORM + j) * 4u);
	}
	float2 Eprev[K];
	float p1[K];
	bload_bins<K>(rS, vb8, SL::E * 4, Eprev);
	bload_vec<K>(rS, vb4, SL::POWER1 * 4, p1);
	float pw_F = g_load_f(rS, oS + (SL::TAIL + 0) * 4u), p1_F = g_load_f(rS, oS + (SL::TAIL + 1) * 4u);
	float eh_F = g_load_f(rS, oS + (SL::TAIL + 2) * 4u), yh_F = g_load_f(rS, oS + (SL::TAIL + 3) * 4u);
	bool prop_dirty = false;
	WSYNC();

	auto load_row = [&](rsrc_t r, float (&v)[K]) { // K int16 samples of the frame
		const u2v t = __builtin_amdgcn_raw_buffer_load_b64(r, oRow, 0, 0);
		v[0] = (float)(int)(short)(t.x & 0xffffu), v[1] = (float)(int)(short)(t.x >> 16);
		v[2] = (float)(int)(short)(t.y & 0xffffu), v[3] = (float)(int)(short)(t.y >> 16);
	};

	// ---- far end: pre-emphasis, energy, spectrum of [previous frame | this frame]
	float2 X0[K];
	float Sxx;
	{
		float xp[K], far[K], xn[K];
		bload_vec<K>(rS, vb4, SL::XPREV * 4, xp);
		load_row(rRef, far);
		float prev = GP::shr1(sc.memX, far[K - 1]);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			xn[k] = far[k] - .9f * prev;
			prev = far[k];
		}
		sc.memX = GP::last(far[K - 1]);
		Sxx = GSeq<F>::inner_prod(xn, xn);
		WSYNC();
		store_vec<K>(L.tbuf + e0, xp);
		store_vec<K>(L.tbuf + F + e0, xn);
		g_rfft_forward<F>(L, LW, X0);
		bstore_vec<K>(rS, vb4, SL::XPREV * 4, xn);
	}
	bool pendingFG = sc.fg_pending != 0, pendingBG = sc.bg_pending != 0;
	const bool postfilter = (a.flags & 1) != 0; // MI_AEC_POSTFILTER

	// ---- near end: saturation flag, DC notch (serial IIR), pre-emphasis
	float input[K], micf[K];
	bool any_sat;
	{
		bool satl = false;
		load_row(rMic, micf);
#pragma unroll
		for (int k = 0; k < K; ++k) satl |= (micf[k] <= -32000.f || micf[k] >= 32000.f);
		any_sat = GP::any(satl);
		const float radius = a.notch_radius;
		const float den2 = (float)(radius * radius + .7 * (1 - radius) * (1 - radius));
		float v[K];
		GSeq<F>::dc_notch(micf, radius, den2, sc.notch0, sc.notch1, v);
		float vprev = GP::shr1(sc.memD, v[K - 1]);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			input[k] = v[k] - .9f * vprev;
			vprev = v[k];
		}
		sc.memD = GP::last(v[K - 1]);
	}
	sc.cancel_count++;
	sc.frames++;

	// ---- newest far-end spectrum into the ring; its power spectrum is all the rest of the frame needs of it
	const int head = (sc.xhead + M) % (M + 1);
	sc.xhead = head;
	bstore_bins<K>(rX, oX + (unsigned)head * (F * 8), 0, X0);
	float Xf[K], Xf_F = 0;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (e0 + k == 0) {
			Xf[k] = X0[k].x * X0[k].x;
			Xf_F = X0[k].y * X0[k].y;
		} else {
			Xf[k] = X0[k].x * X0[k].x + X0[k].y * X0[k].y;
		}
	}
	Xf_F = GP::first(Xf_F);

	// ---- proportional step (mdf_adjust_prop): prop_j = sqrt(1 + |W_j|^2) + .1 max, normalised by the library's serial sum
	{
		WSYNC();
		float mx = 1.f;
		for (int j = lane; j < M; j += G) {
			const float p = sqrt_via_double(1.0f + L.wnorm[j]);
			L.vec[j] = p; // (per-bin exchange array: free here, F >= 64 >= M)
			mx = p > mx ? p : mx;
		}
		const float max_sum = GP::tree(mx, [](float x, float y) { return y > x ? y : x; });
		WSYNC();
		float prop_sum = 1.f;
		for (int j = 0; j < M; ++j) prop_sum = prop_sum + (L.vec[j] + .1f * max_sum);
		WSYNC();
		if (sc.adapted) {
			for (int j = lane; j < M; j += G) L.prop[j] = (.99f * (L.vec[j] + .1f * max_sum)) / prop_sum;
			prop_dirty = true;
		}
	}
	WSYNC();
	const bool do_grad = (sc.saturated == 0);
	if (!do_grad) sc.saturated--;

	// W += prop p1 conj(X) E, bin by bin (weighted_spectral_mul_conj); bin 0 = (DC, Nyquist): real products with their own steps
	auto grad = [&](v2f (&w)[K], const v2f (&x)[K], float prop) {
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const v2f xv = x[k], ev = {Eprev[k].x, Eprev[k].y};
			const float Wt = prop * p1[k];
			v2f st = pk_cmul_conj(xv, ev) * (v2f){Wt, Wt};
			if (k == 0) {
				const v2f dc = (xv * ev) * (v2f){Wt, prop * p1_F};
				if (e0 == 0) st = dc;
			}
			w[k] = w[k] + st;
		}
	};

	// ---- one streaming pass over the blocks (next block's loads in flight); AUMDF constraint where the pass meets block
	// 0 and the leg's round-robin block -- for the whole wave (the transforms need every lane), kept by the legs that ask
	const int jc = (M > 1) ? (sc.cancel_count % (M - 1)) + 1 : -1;
	float2 yfg[K], ybgs[K];
#pragma unroll
	for (int k = 0; k < K; ++k) yfg[k] = ybgs[k] = make_float2(0, 0);
	{
		v2f xj[K], xn[K], fg[K], wl[K];
#pragma unroll
		for (int k = 0; k < K; ++k) xj[k] = (v2f){X0[k].x, X0[k].y};
		// a filter copy the previous frame asked for (aec_tick.hpp: pendingFG / pendingBG): which half is read as what, and
		// where the updated background goes
		const bool carryFG = pendingFG, carryBG = pendingBG;
		pendingFG = pendingBG = false;
		const unsigned fsrc = oW + (carryFG ? wo : fo), wsrc = oW + (carryBG ? fo : wo), wdst = oW + (carryFG ? fo : wo);
		int xi = head + 1; // ring position of X(1)
		if (xi > M) xi = 0;
		bload_bins<K>(rX, oX + (unsigned)xi * (F * 8), 0, xn);
		bload_bins<K>(rWF, fsrc, 0, fg);
		bload_bins<K>(rWF, wsrc, 0, wl);
		for (int j = 0; j < M; ++j) {
			v2f xn2[K], fg2[K], wl2[K];
			{ // (behind the last block: its own again, dropped)
				const int jn = j + 1 < M ? j + 1 : j;
				int xi2 = xi + (j + 1 < M ? 1 : 0);
				if (xi2 > M) xi2 = 0;
				bload_bins<K>(rX, oX + (unsigned)xi2 * (F * 8), 0, xn2);
				bload_bins<K>(rWF, fsrc + (unsigned)jn * (F * 8), 0, fg2);
				bload_bins<K>(rWF, wsrc + (unsigned)jn * (F * 8), 0, wl2);
				xi = xi2;
			}
			const bool aumdf = (j == 0 || j == jc);
			if (do_grad) grad(wl, xn, L.prop[j]);
			if (__any(aumdf)) {
				float2 w[K];
#pragma unroll
				for (int k = 0; k < K; ++k) w[k] = make_float2(wl[k].x, wl[k].y);
				g_rfft_inverse<F>(L, LW, w);
				float z[K];
#pragma unroll
				for (int k = 0; k < K; ++k) z[k] = 0.f;
				store_vec<K>(w_time(L) + F + e0, z);
				g_rfft_forward<F>(L, LW, w, w_time(L));
				if (aumdf) {
#pragma unroll
					for (int k = 0; k < K; ++k) wl[k] = (v2f){w[k].x, w[k].y};
				}
			}
			if (do_grad || aumdf || carryBG || carryFG) bstore_bins<K>(rWF, wdst + (unsigned)j * (F * 8), 0, wl);
			cmac_bins<K>(yfg, xj, fg, e0);
			cmac_bins<K>(ybgs, xj, wl, e0);
			{
				float t[K];
#pragma unroll
				for (int k = 0; k < K; ++k) t[k] = wl[k].x * wl[k].x + wl[k].y * wl[k].y;
				const float nn = GP::tree((t[0] + t[1]) + (t[2] + t[3]), [](float x, float y) { return x + y; });
				if (lane == 0) L.wnorm[j] = nn; // feeds the NEXT frame's proportional step
			}
#pragma unroll
			for (int k = 0; k < K; ++k) xj[k] = xn[k], xn[k] = xn2[k], fg[k] = fg2[k], wl[k] = wl2[k];
		}
		if (carryFG) { // the halves have swapped roles
			const unsigned t = wo;
			wo = fo, fo = t;
			sc.wsel ^= 1;
		}
	}

	// ---- time-domain responses
	float efg[K], ybg[K], e1[K], e2[K], dresp[K];
	g_rfft_inverse<F>(L, LW, yfg);
	load_vec<K>(w_time(L) + F + e0, efg);
#pragma unroll
	for (int k = 0; k < K; ++k) e1[k] = input[k] - efg[k];
	g_rfft_inverse<F>(L, LW, ybgs);
	load_vec<K>(w_time(L) + F + e0, ybg);
#pragma unroll
	for (int k = 0; k < K; ++k) {
		e2[k] = input[k] - ybg[k];
		dresp[k] = efg[k] - ybg[k];
	}
	float Sff, Dbf, See;
	GSeq<F>::inner_prod3(e1, e1, dresp, dresp, e2, e2, Sff, Dbf, See);
	Dbf = 10 + Dbf;

	// ---- two-path control
	sc.Davg1 = .6f * sc.Davg1 + .4f * (Sff - See);
	sc.Davg2 = .85f * sc.Davg2 + .15f * (Sff - See);
	sc.Dvar1 = .36f * sc.Dvar1 + (.4f * Sff) * (.4f * Dbf);
	sc.Dvar2 = .7225f * sc.Dvar2 + (.15f * Sff) * (.15f * Dbf);
	bool update_foreground = false;
	if ((Sff - See) * fabsf(Sff - See) > Sff * Dbf) update_foreground = true;
	else if (sc.Davg1 * fabsf(sc.Davg1) > .5f * sc.Dvar1) update_foreground = true;
	else if (sc.Davg2 * fabsf(sc.Davg2) > .25f * sc.Dvar2) update_foreground = true;
	WSYNC();
	if (update_foreground) {
		sc.fg_updates++;
		sc.Davg1 = sc.Davg2 = 0;
		sc.Dvar1 = sc.Dvar2 = 0;
		pendingFG = true; // the next pass over the filter carries the copy out
		for (int j = lane; j < M; j += G) L.fgnorm[j] = L.wnorm[j];
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
			sc.bg_resets++;
			pendingBG = true; // the next pass reads the foreground's blocks as the background's
			for (int j = lane; j < M; j += G) L.wnorm[j] = L.fgnorm[j];
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
		GSeq<F>::deemphasis(d, sc.memE, tout);
#pragma unroll
		for (int k = 0; k < K; ++k) out_i[k] = word2int(tout[k]);
	}
	float Sey, Syy, Sdd;
	GSeq<F>::inner_prod3(e2, ybg, ybg, ybg, input, input, Sey, Syy, Sdd);
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
		g_rfft_forward<F>(L, LW, Ecur);
		store_vec<K>(L.tbuf + e0, z);
		store_vec<K>(L.tbuf + F + e0, ybg);
		g_rfft_forward<F>(L, LW, Ycur);
	}
	float Rf[K], Yf[K], Rf_F = 0, Yf_F = 0;
#pragma unroll
	for (int k = 0; k < K; ++k) {
		if (e0 + k == 0) {
			Rf[k] = Ecur[k].x * Ecur[k].x;
			Rf_F = Ecur[k].y * Ecur[k].y;
			Yf[k] = Ycur[k].x * Ycur[k].x;
			Yf_F = Ycur[k].y * Ycur[k].y;
		} else {
			Rf[k] = Ecur[k].x * Ecur[k].x + Ecur[k].y * Ecur[k].y;
			Yf[k] = Ycur[k].x * Ycur[k].x + Ycur[k].y * Ycur[k].y;
		}
	}
	Rf_F = GP::first(Rf_F);
	Yf_F = GP::first(Yf_F);
#pragma unroll
	for (int k = 0; k < K; ++k) Eprev[k] = Ecur[k];

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
	// the echo estimate pair the residual-echo stage works on: [the last frame's | this frame's]
	float ly_old[K], ly_new[K];
	bload_vec<K>(rS, vb4, (SL::LASTY + F) * 4, ly_old);
	float leak_pf; // leak estimate the post-filter sees
	const bool was_reset = sc.screwed_up >= 50;
	float pw[K];
	bload_vec<K>(rS, vb4, SL::POWER * 4, pw);
	float eh[K], yh[K];
	bload_vec<K>(rS, vb4, SL::EH * 4, eh);
	bload_vec<K>(rS, vb4, SL::YH * 4, yh);
	if (was_reset) { // speex_echo_state_reset
		float2 z2[K];
#pragma unroll
		for (int k = 0; k < K; ++k) z2[k] = make_float2(0, 0);
		for (int j = 0; j < M; ++j) {
			bstore_bins<K>(rWF, oW + (unsigned)j * (F * 8), 0, z2);
			bstore_bins<K>(rWF, oW + HALF + (unsigned)j * (F * 8), 0, z2);
		}
		for (int j = 0; j <= M; ++j) bstore_bins<K>(rX, oX + (unsigned)j * (F * 8), 0, z2);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			p1[k] = 1.0f;
			Eprev[k] = make_float2(0, 0);
			pw[k] = eh[k] = yh[k] = 0.f;
			ly_old[k] = ly_new[k] = 0.f;
		}
		pw_F = eh_F = yh_F = 0.f;
		p1_F = 1.0f;
		for (int j = lane; j < M; j += G) L.wnorm[j] = 0, L.fgnorm[j] = 0;
		{
			float z[K];
#pragma unroll
			for (int k = 0; k < K; ++k) z[k] = 0.f;
			bstore_vec<K>(rS, vb4, SL::XPREV * 4, z);
		}
		sc.state_resets++;
		sc.cancel_count = 0;
		sc.screwed_up = 0;
		sc.notch0 = sc.notch1 = 0;
		sc.memD = sc.memE = sc.memX = 0;
		sc.saturated = 0;
		sc.adapted = 0;
		sc.sum_adapt = 0;
		sc.Pey = sc.Pyy = 1.0f;
		sc.Davg1 = sc.Davg2 = sc.Dvar1 = sc.Dvar2 = 0;
		pendingFG = pendingBG = false;
		leak_pf = sc.leak_estimate;
	} else {
		if (See < (float)(N * 100)) See = (float)(N * 100);
		float Sxx2 = Sxx + Sxx; // sic: the library accumulates the far-end energy a second time here

		// ---- far-end power, leak estimate
#pragma unroll
		for (int k = 0; k < K; ++k) pw[k] = a.ss_1 * pw[k] + 1 + a.ss * Xf[k];
		pw_F = a.ss_1 * pw_F + 1 + a.ss * Xf_F;
		float Ehd[K], Yhd[K], Ehd_F, Yhd_F;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			Ehd[k] = Rf[k] - eh[k];
			Yhd[k] = Yf[k] - yh[k];
			eh[k] = (1 - a.spec_average) * eh[k] + a.spec_average * Rf[k];
			yh[k] = (1 - a.spec_average) * yh[k] + a.spec_average * Yf[k];
		}
		Ehd_F = Rf_F - eh_F;
		Yhd_F = Yf_F - yh_F;
		eh_F = (1 - a.spec_average) * eh_F + a.spec_average * Rf_F;
		yh_F = (1 - a.spec_average) * yh_F + a.spec_average * Yf_F;
		float Pey = 1.0f, Pyy = 1.0f;
		Pey = Pey + Ehd_F * Yhd_F;
		Pyy = Pyy + Yhd_F * Yhd_F;
		GSeq<F>::dot_desc2(Pey, Ehd, Yhd, Pyy, Yhd, Yhd, Pey, Pyy);
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
		float RER = (float)((.0001 * Sxx2 + 3. * (sc.leak_estimate * Syy)) / See);
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
		if (sc.adapted) {
#pragma unroll
			for (int k = 0; k < K; ++k) p1[k] = step(Yf[k], Rf[k], pw[k]);
			p1_F = step(Yf_F, Rf_F, pw_F);
		} else {
			float adapt_rate = 0;
			if (Sxx2 > (float)(N * 1000)) {
				tmp32 = .25f * Sxx2;
				if (tmp32 > .25 * See) tmp32 = (float)(.25 * See);
				adapt_rate = tmp32 / See;
			}
#pragma unroll
			for (int k = 0; k < K; ++k) p1[k] = adapt_rate / (pw[k] + 10);
			p1_F = adapt_rate / (pw_F + 10);
			sc.sum_adapt = sc.sum_adapt + adapt_rate;
		}
		// ---- echo estimate of this frame for the residual-echo stage
#pragma unroll
		for (int k = 0; k < K; ++k) ly_new[k] = sc.adapted ? (float)((int)micf[k] - out_i[k]) : ly_old[k];
		leak_pf = sc.leak_estimate;
	}

	// ---- the frame's state back to HBM
	bstore_bins<K>(rS, vb8, SL::E * 4, Eprev);
	bstore_vec<K>(rS, vb4, SL::POWER1 * 4, p1);
	bstore_vec<K>(rS, vb4, SL::POWER * 4, pw);
	bstore_vec<K>(rS, vb4, SL::EH * 4, eh);
	bstore_vec<K>(rS, vb4, SL::YH * 4, yh);
	WSYNC();
	for (int j = lane; j < M; j += G) {
		g_store_f(rS, oS + (unsigned)(SL::WNORM + j) * 4u, L.wnorm[j]);
		g_store_f(rS, oS + (unsigned)(SL::FGNORM + j) * 4u, L.fgnorm[j]);
		if (prop_dirty) g_store_f(rS, oS + (unsigned)(SL::PROP + j) * 4u, L.prop[j]);
	}
	bstore_vec<K>(rS, vb4, SL::LASTY * 4, ly_old);
	bstore_vec<K>(rS, vb4, (SL::LASTY + F) * 4, ly_new);
	if (lane == 0) {
		g_store_f(rS, oS + (SL::TAIL + 0) * 4u, pw_F);
		g_store_f(rS, oS + (SL::TAIL + 1) * 4u, p1_F);
		g_store_f(rS, oS + (SL::TAIL + 2) * 4u, eh_F);
		g_store_f(rS, oS + (SL::TAIL + 3) * 4u, yh_F);
	}
	auto store_out = [&](const int (&o)[K]) {
		u2v t = {(unsigned)(o[0] & 0xffff) | ((unsigned)o[1] << 16), (unsigned)(o[2] & 0xffff) | ((unsigned)o[3] << 16)};
		__builtin_amdgcn_raw_buffer_store_b64(t, rOut, oRow, 0, 0);
	};
	if (!postfilter) {
		sc.fg_pending = pendingFG ? 1 : 0, sc.bg_pending = pendingBG ? 1 : 0;
		store_out(out_i);
		if (lane == 0) g_store_scalars(rSc, oSc, sc);
		return;
	}

	// ---- residual-echo / noise post-filter (speex_preprocess_run) of the same frame: per-bin state in registers, one or
	// two Bark bands per lane (24 bands over the leg's G lanes)
	constexpr int NBL = (NB_BANDS + G - 1) / G;
	float en[K], inb[K], S[K], Smin[K], Stmp[K], noise[K], old_ps[K], zeta[K], ob[K], wl[K], wr[K], h0[K], h1[K], w0[K], w1[K];
	bload_vec<K>(rS, vb4, SL::ECHON * 4, en);
	bload_vec<K>(rS, vb4, SL::INBUF * 4, inb);
	bload_vec<K>(rS, vb4, SL::S_ * 4, S);
	bload_vec<K>(rS, vb4, SL::SMIN * 4, Smin);
	bload_vec<K>(rS, vb4, SL::STMP * 4, Stmp);
	bload_vec<K>(rS, vb4, SL::NOISE * 4, noise);
	bload_vec<K>(rS, vb4, SL::OLDPS * 4, old_ps);
	bload_vec<K>(rS, vb4, SL::ZETA * 4, zeta);
	bload_vec<K>(rS, vb4, SL::OUTBUF * 4, ob);
	load_vec<K>(a.t.bfl + e0, wl);
	load_vec<K>(a.t.bfr + e0, wr);
	load_vec<K>(a.t.hann + e0, h0);
	load_vec<K>(a.t.hann + F + e0, h1);
	load_vec<K>(a.t.pwin + e0, w0);
	load_vec<K>(a.t.pwin + F + e0, w1);
	float old_ps_b[NBL], zeta_b[NBL];
#pragma unroll
	for (int bi = 0; bi < NBL; ++bi) {
		const int b = lane + bi * G;
		old_ps_b[bi] = zeta_b[bi] = 0;
		if (b < NB_BANDS) {
			old_ps_b[bi] = g_load_f(rS, oS + (unsigned)(SL::OLDPS_B + b) * 4u);
			zeta_b[bi] = g_load_f(rS, oS + (unsigned)(SL::ZETA_B + b) * 4u);
		}
	}
	float *pl = L.spec, *pr = L.spec + F;
	float *bandv = L.band;
	float *lvec = L.vec;
	int out_f[K];
	{
		sc.nb_adapt++;
		if (sc.nb_adapt > 20000) sc.nb_adapt = 20000;
		sc.min_count++;
		float beta = 1.0f / sc.nb_adapt;
		if (beta < .03f) beta = .03f;
		const float beta_1 = 1.0f - beta;
		const float leak = leak_pf;

		// residual echo spectrum (speex_echo_get_residual)
		{
			float lo[K], ln[K];
#pragma unroll
			for (int k = 0; k < K; ++k) lo[k] = h0[k] * (was_reset ? 0.f : ly_old[k]), ln[k] = h1[k] * ly_new[k];
			WSYNC();
			store_vec<K>(L.tbuf + e0, lo);
			store_vec<K>(L.tbuf + F + e0, ln);
		}
		float2 Yr[K];
		g_rfft_forward<F>(L, LW, Yr);
		const float leak2 = (leak > .5) ? 1.f : 2 * leak;
		float res[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const float r = (e0 + k == 0) ? Yr[k].x * Yr[k].x : Yr[k].x * Yr[k].x + Yr[k].y * Yr[k].y;
			res[k] = (float)(int32_t)(leak2 * r);
		}
		const float res0 = GP::first(res[0]);
		const bool bad = !(res0 >= 0 && res0 < F * 1e9f);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const float rr = bad ? 0.f : res[k];
			const float c = .6f * en[k];
			en[k] = c > rr ? c : rr;
			pl[e0 + k] = wl[k] * en[k];
			pr[e0 + k] = wr[k] * en[k];
		}
		// analysis frame [inbuf, x] * window
		{
			float xcur[K], a0[K], a1[K];
#pragma unroll
			for (int k = 0; k < K; ++k) {
				xcur[k] = (float)(int16_t)out_i[k];
				a0[k] = inb[k] * w0[k];
				a1[k] = xcur[k] * w1[k];
				inb[k] = xcur[k];
			}
			store_vec<K>(L.tbuf + e0, a0);
			store_vec<K>(L.tbuf + F + e0, a1);
		}
		float2 ft[K];
		g_rfft_forward<F>(L, LW, ft);
		float ps[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			ps[k] = (e0 + k == 0) ? ft[k].x * ft[k].x : ft[k].x * ft[k].x + ft[k].y * ft[k].y;
			lvec[e0 + k] = ps[k];
			L.tbuf[e0 + k] = wl[k] * ps[k];     // the analysis transform is done with its input: the frame's filterbank
			L.tbuf[F + e0 + k] = wr[k] * ps[k]; // products wait there (left halves, right halves) for the fused band sums
		}
		WSYNC();
		// update_noise_prob
		int min_range;
		if (sc.nb_adapt < 100) min_range = 15;
		else if (sc.nb_adapt < 1000) min_range = 50;
		else if (sc.nb_adapt < 10000) min_range = 150;
		else min_range = 300;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int b = e0 + k;
			if (b == 0 || b == F - 1) S[k] = .8f * S[k] + .2f * ps[k];
			else S[k] = .8f * S[k] + .05f * lvec[b - 1] + .1f * ps[k] + .05f * lvec[b + 1];
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
		{
			float *np = w_time(L); // free between the analysis and the synthesis transform
#pragma unroll
			for (int k = 0; k < K; ++k) {
				np[e0 + k] = wl[k] * noise[k];
				np[F + e0 + k] = wr[k] * noise[k];
			}
			WSYNC();
			// echo estimate (products in L.spec since the start of the frame), frame, noise: three band sums in one loop
#pragma unroll
			for (int bi = 0; bi < NBL; ++bi) {
				const int b = lane + bi * G;
				if (b < NB_BANDS) band_sum3<F>(a.t, b, L.spec, L.tbuf, np, bandv[b], bandv[NB_BANDS + b], bandv[2 * NB_BANDS + b]);
			}
		}
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
		float post[K], prior[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (sc.nb_adapt == 1) old_ps[k] = ps[k];
			snr(ps[k], noise[k], en[k], old_ps[k], post[k], prior[k]);
			lvec[e0 + k] = prior[k];
		}
		float post_b[NBL], prior_b[NBL], ps_b[NBL];
#pragma unroll
		for (int bi = 0; bi < NBL; ++bi) {
			const int b = lane + bi * G;
			post_b[bi] = prior_b[bi] = ps_b[bi] = 0;
			if (b < NB_BANDS) {
				ps_b[bi] = bandv[NB_BANDS + b];
				if (sc.nb_adapt == 1) old_ps_b[bi] = ps_b[bi];
				snr(ps_b[bi], bandv[2 * NB_BANDS + b], bandv[b], old_ps_b[bi], post_b[bi], prior_b[bi]);
			}
		}
		WSYNC();
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int b = e0 + k;
			if (b == 0 || b >= F - 1) zeta[k] = .7f * zeta[k] + .3f * prior[k];
			else zeta[k] = .7f * zeta[k] + .15f * prior[k] + .075f * lvec[b - 1] + .075f * lvec[b + 1];
		}
#pragma unroll
		for (int bi = 0; bi < NBL; ++bi) {
			const int b = lane + bi * G;
			if (b < NB_BANDS) {
				zeta_b[bi] = .7f * zeta_b[bi] + .3f * prior_b[bi];
				bandv[3 * NB_BANDS + b] = zeta_b[bi];
			}
		}
		WSYNC();
		float Zframe = 0;
		for (int i = 0; i < NB_BANDS; ++i) Zframe = Zframe + bandv[3 * NB_BANDS + i];
		const float Pframe = .1f + .899f * qcurve(Zframe / NB_BANDS);
		const int eff_echo = (int)((1.0f - Pframe) * -40 + Pframe * -15);
		float g2v[NBL], gv[NBL], gfv[NBL];
#pragma unroll
		for (int bi = 0; bi < NBL; ++bi) {
			const int b = lane + bi * G;
			g2v[bi] = gv[bi] = gfv[bi] = 0;
			if (b < NB_BANDS) {
				const float noise_floor = (float)exp((double)(.2302585f * -15));
				const float echo_floor = (float)exp((double)(.2302585f * eff_echo));
				const float nb = bandv[2 * NB_BANDS + b], eb = bandv[b];
				const float gfloor = (float)(sqrt((double)(noise_floor * nb + echo_floor * eb)) / sqrt((double)(1 + nb + eb)));
				const float prior_ratio = prior_b[bi] / (prior_b[bi] + 1.f);
				const float theta = prior_ratio * (1.f + post_b[bi]);
				const float MM = hypergeom_gain(theta);
				float gg = prior_ratio * MM;
				if (gg > 1.f) gg = 1.f;
				old_ps_b[bi] = .2f * old_ps_b[bi] + (.8f * (gg * gg)) * ps_b[bi];
				const float P1 = .199f + .8f * qcurve(zeta_b[bi]);
				const float q = 1.0f - Pframe * P1;
				g2v[bi] = (float)(1 / (1.f + (q / (1.f - q)) * (1 + prior_b[bi]) * exp((double)(-theta))));
				gv[bi] = gg;
				gfv[bi] = gfloor;
			}
		}
		WSYNC(); // (every lane has read the band energies: the gains take their place)
#pragma unroll
		for (int bi = 0; bi < NBL; ++bi) {
			const int b = lane + bi * G;
			if (b < NB_BANDS) {
				bandv[b] = g2v[bi];
				bandv[NB_BANDS + b] = gv[bi];
				bandv[2 * NB_BANDS + b] = gfv[bi];
			}
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
			float gg = prior_ratio * MM;
			if (gg > 1.f) gg = 1.f;
			if (.333f * gg > gain_bark) gg = 3 * gain_bark;
			float gain = gg;
			old_ps[k] = .2f * old_ps[k] + (.8f * (gain * gain)) * ps[k];
			if (gain < gfl) gain = gfl;
			const float tmp = p * sqrt_via_double(gain) + (1.0f - p) * sqrt_via_double(gfl);
			gain2[k] = tmp * tmp;
		}
		const float g_last = GP::last(gain2[K - 1]); // gain2[F-1] scales the Nyquist term
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
		g_rfft_inverse<F>(L, LW, ft);
		{
			float lo[K], hi[K];
			load_vec<K>(w_time(L) + e0, lo);
			load_vec<K>(w_time(L) + F + e0, hi);
#pragma unroll
			for (int k = 0; k < K; ++k) {
				out_f[k] = word2int(ob[k] + lo[k] * w0[k]);
				ob[k] = hi[k] * w1[k];
			}
		}
	}
	store_out(out_f);
	bstore_vec<K>(rS, vb4, SL::ECHON * 4, en);
	bstore_vec<K>(rS, vb4, SL::INBUF * 4, inb);
	bstore_vec<K>(rS, vb4, SL::S_ * 4, S);
	bstore_vec<K>(rS, vb4, SL::SMIN * 4, Smin);
	bstore_vec<K>(rS, vb4, SL::STMP * 4, Stmp);
	bstore_vec<K>(rS, vb4, SL::NOISE * 4, noise);
	bstore_vec<K>(rS, vb4, SL::OLDPS * 4, old_ps);
	bstore_vec<K>(rS, vb4, SL::ZETA * 4, zeta);
	bstore_vec<K>(rS, vb4, SL::OUTBUF * 4, ob);
#pragma unroll
	for (int bi = 0; bi < NBL; ++bi) {
		const int b = lane + bi * G;
		if (b < NB_BANDS) {
			g_store_f(rS, oS + (unsigned)(SL::OLDPS_B + b) * 4u, old_ps_b[bi]);
			g_store_f(rS, oS + (unsigned)(SL::ZETA_B + b) * 4u, zeta_b[bi]);
		}
	}
	sc.fg_pending = pendingFG ? 1 : 0, sc.bg_pending = pendingBG ? 1 : 0;
	if (lane == 0) g_store_scalars(rSc, oSc, sc);
}
