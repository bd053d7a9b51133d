// aec_tick.hpp -- the AEC kernels in their per-TICK form (included by aec.hip after aec_wave.hpp, whose FFT / ordered-sum
// helpers they share).  One launch serves every frame a stream has ready in this 10 ms tick: at 48 kHz the filter's
// 256-sample frames (speexec.c:171-180) make that one or two frames (15 per 8 ticks).
//
// Why per tick: the canceller is HBM-bound, and with both frames of a tick inside one wavefront
//   * the per-stream "small" state (error spectrum, step sizes, power / Eh / Yh, post-filter estimates: ~40 KB moved per
//     frame by the per-frame kernels) is read once and written once per TICK -- it lives in registers in between;
//   * the foreground filter is streamed ONCE for both frames: frame 2's foreground response sum_j X'(j) FG(j) uses the
//     same FG blocks as frame 1's (X'(j) = X(j-1)), so it is accumulated in frame 1's pass -- "speculatively": if frame
//     1 decides to copy the background into the foreground (update_foreground), frame 2's response is instead
//     sum_j X'(j) W1(j), which frame 2's own pass over W yields for free (W1(j) is what it loads before updating it),
//     and the copy itself degenerates into stores of the blocks that pass already holds (no second read of W);
//   * nothing else changes: every sum keeps the library's operand order, so results are bit-identical to the per-frame
//     kernels (tests/test_gpu_aec.py compares both with the oracle).
//   * the residual-echo / noise post-filter (speex_preprocess_run) runs as the TAIL PHASE of the same wavefront: the
//     canceller is HBM-bound and the post-filter VALU-bound, so waves in their post-filter phase leave the memory system
//     to the waves that are streaming -- the overlap two launches on two HIP streams only approximated (a separate
//     post-filter launch kept HBM idle for 17 % of the tick); the canceller's output frames and echo estimates reach the
//     post-filter through LDS instead of HBM.
// Cost: the second frame's far-end spectrum is needed early, and ~50 more live registers: 2 waves per SIMD at F = 256
// instead of 3 (measured on the per-frame kernel: 1.5 % slower per launch at that occupancy; the bytes saved are ~20 %).

template <int F>
struct TickLayout { // offsets in floats inside the per-stream small-state block
	static constexpr int XPREV = 0;          // F   pre-emphasised far end of the last frame (first half of the next FFT input)
	static constexpr int E = F;              // 2F  error spectrum of the last frame, bin-interleaved
	static constexpr int POWER = 3 * F;      // F   (+1 in TAIL)
	static constexpr int POWER1 = 4 * F;
	static constexpr int EH = 5 * F;
	static constexpr int YH = 6 * F;
	static constexpr int LASTY = 7 * F;      // 2F  echo estimate of the last two frames [older | newest] (3F reserved)
	static constexpr int ECHON = 10 * F;     // post-filter
	static constexpr int INBUF = 11 * F;
	static constexpr int OUTBUF = 12 * F;
	static constexpr int NOISE = 13 * F;
	static constexpr int OLDPS = 14 * F;
	static constexpr int ZETA = 15 * F;
	static constexpr int S_ = 16 * F;
	static constexpr int SMIN = 17 * F;
	static constexpr int STMP = 18 * F;
	static constexpr int MISC = 19 * F;
	static constexpr int PROP = MISC;        // M <= 64
	static constexpr int WNORM = MISC + 64;  // M
	static constexpr int TAIL = MISC + 128;  // POWER[F], POWER1[F], EH[F], YH[F]
	static constexpr int OLDPS_B = MISC + 136; // 24 Bark bands
	static constexpr int ZETA_B = MISC + 160;
	static constexpr int FGNORM = MISC + 192; // M: |foreground block|^2, what WNORM becomes when the background is reset to it
	static constexpr int TOTAL = 19 * F + 256;
};

// LDS of the canceller kernel: FFT work space + tables and the per-bin state parked while the blocks stream (none of the
// post-filter's arrays): 16.3 KB at F = 256.  (Tables read from global memory instead cost 24 % of the kernel at 2 waves
// per SIMD: 6.35 against 5.14 ms per tick of 65 536 legs.)
template <int F>
struct alignas(16) TLds {
	float2 zbuf[F];      // complex FFT work
	float tbuf[2 * F];   // time-domain exchange / inverse-transform staging
	float spec[2 * F];   // bin-interleaved spectrum exchange
	float2 tw[F], super[F];
	uint16_t perm[F];
	float prop[64], wnorm[64];
	// per-bin state that is not needed while the blocks stream: parked here instead of in registers
	float pw[F], eh[F], yh[F], input[F];
	// canceller -> post-filter hand-over: echo estimates [before the tick | after frame 1 | after frame 2], output frames
	float ly[3][F];
	int16_t outf[2][F];
	// the post-filter phase reuses the parked arrays (their contents are in HBM by then)
	__device__ float *vec() { return pw; }        // F
	__device__ float *band() { return eh; }       // 4 * NB_BANDS + 8
};


// ---- buffer addressing: one 128-bit descriptor (SGPRs) per per-stream array + ONE per-lane byte offset + a scalar
// offset, instead of a 64-bit per-lane address (two VGPRs) per array and block: the canceller touches ~25 arrays
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const void *p, unsigned bytes) {
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float u2f(unsigned v) { return __uint_as_float(v); }
__device__ __forceinline__ unsigned f2u(float v) { return __float_as_uint(v); }
template <typename T>
__device__ __forceinline__ T mk2(float x, float y) {
	T t;
	t.x = x, t.y = y;
	return t;
}
// T = float2, or v2f: a 64-bit register pair per bin, so that whole bins move with one v_mov_b64 / feed v_pk_* directly
template <int K, typename T>
__device__ __forceinline__ void bload_bins(rsrc_t r, unsigned voff, unsigned soff, T (&v)[K]) {
	if constexpr (K == 1) {
		const u2v t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
		v[0] = mk2<T>(u2f(t.x), u2f(t.y));
	} else {
#pragma unroll
		for (int k = 0; k < K; k += 2) {
			const u4v t = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 8 * k, soff, 0);
			v[k] = mk2<T>(u2f(t.x), u2f(t.y));
			v[k + 1] = mk2<T>(u2f(t.z), u2f(t.w));
		}
	}
}
template <int K, typename T>
__device__ __forceinline__ void bstore_bins(rsrc_t r, unsigned voff, unsigned soff, const T (&v)[K]) {
	if constexpr (K == 1) {
		u2v t = {f2u(v[0].x), f2u(v[0].y)};
		__builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 0);
	} else {
#pragma unroll
		for (int k = 0; k < K; k += 2) {
			u4v t = {f2u(v[k].x), f2u(v[k].y), f2u(v[k + 1].x), f2u(v[k + 1].y)};
			__builtin_amdgcn_raw_buffer_store_b128(t, r, voff + 8 * k, soff, 0);
		}
	}
}
template <int K>
__device__ __forceinline__ void bload_vec(rsrc_t r, unsigned voff, unsigned soff, float (&v)[K]) {
	if constexpr (K == 4) {
		const u4v t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
		v[0] = u2f(t.x), v[1] = u2f(t.y), v[2] = u2f(t.z), v[3] = u2f(t.w);
	} else if constexpr (K == 2) {
		const u2v t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
		v[0] = u2f(t.x), v[1] = u2f(t.y);
	} else {
		v[0] = u2f(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
	}
}
template <int K>
__device__ __forceinline__ void bstore_vec(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[K]) {
	if constexpr (K == 4) {
		u4v t = {f2u(v[0]), f2u(v[1]), f2u(v[2]), f2u(v[3])};
		__builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 0);
	} else if constexpr (K == 2) {
		u2v t = {f2u(v[0]), f2u(v[1])};
		__builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 0);
	} else {
		__builtin_amdgcn_raw_buffer_store_b32(f2u(v[0]), r, voff, soff, 0);
	}
}

// acc += x w, bin by bin; bin 0 (lane 0's first) holds (DC, Nyquist): two real products there
template <int K, typename TA, typename TX, typename TW>
__device__ __forceinline__ void cmac_bins(TA (&acc)[K], const TX (&x)[K], const TW (&w)[K], int e0) {
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const v2f xv = {x[k].x, x[k].y}, wv = {w[k].x, w[k].y};
		v2f pr = pk_cmul(xv, wv); // (x.x w.x - x.y w.y, x.y w.x + x.x w.y)
		if (k == 0) {
			const v2f dc = xv * wv; // (x.x w.x, x.y w.y)
			if (e0 == 0) pr = dc;
		}
		const v2f a = (v2f){acc[k].x, acc[k].y} + pr;
		acc[k] = mk2<TA>(a.x, a.y);
	}
}

// words of AecArgs::ctl (the list hand-over between consecutive launches of the FIFO entry)
// Eight lists (classes), one per XCD (workgroup b runs on XCD b % 8), in two sets: the set of parity p is served by this
// launch, the set of parity p ^ 1 is filled by it for the next one.  A list has room for cap8 entries: legs that will run
// two frames are entered from the front, the others from the back, the slots in between stay empty (their workgroups leave
// at once).  Every class has its own 128-byte line of control words (the atomics of different classes do not queue up behind
// each other) and a ninth line is shared:
//   PLACED  [parity] 64 bits: entries of this class's list of that parity, front | back << 32 -- counted up by the atomics of
//           the launch that fills the list, read as the list's lengths by the launch that serves it
//   global line: PARITY (which set of lists the next launch serves)
// Nothing inside a launch depends on another workgroup's progress: a launch only READS the parity and its own set's lengths
// and only ADDS to the other set's.  The turn-over -- the served set's counts back to zero, the parity flipped -- is a
// launch of its own right behind (aec_tick_advance_kernel: one wavefront, stream order does the rest).  (Rounds 2-3 had the
// last wave of the tick kernel do it, found through per-class completion counts: a class whose list was empty never
// completed -- batches of fewer than 8 legs froze on their first lists, and the counts grew past the lists -- and workgroups
// on empty slots read words the turn-over might already have rewritten.)
struct TickOrder {
	static constexpr int STRIDE = 32, PLACED = 0, GLOBAL = 8 * STRIDE, PARITY = 0, WORDS = 9 * STRIDE;
	// a class's share of the legs wanders by fewer than 16 around nstreams / 8 (the mixing rule below deals every list out
	// over all eight: sixteen roundings of an eighth of a front or back run), whatever the history
	static constexpr int SLACK = 32;
};

__global__ __launch_bounds__(64) void aec_tick_advance_kernel(int *ctl) {
	const int par = ctl[TickOrder::GLOBAL + TickOrder::PARITY];
	if (threadIdx.x < 8) *reinterpret_cast<unsigned long long *>(ctl + threadIdx.x * TickOrder::STRIDE + TickOrder::PLACED + 2 * par) = 0ull;
	if (threadIdx.x == 0) ctl[TickOrder::GLOBAL + TickOrder::PARITY] = par ^ 1;
}

// MODE: where the frames come from and go to -- decided at compile time, so that each form carries only its own addressing,
// scalars and code (the three forms in one kernel cost the headline's form ~1.5 KB of instruction cache and a dozen scalar
// registers it never used):
//   TICK_ROWS     rows of frames handed in by the caller (mi_aec_process / mi_aec_process_frames);
//   TICK_FIFO     the three device FIFOs, the tick's microphone block from the caller's row (mi_aec_process_fifos);
//   TICK_FIFO_RS  the same with the leg's MSResample folded in: the block is up-sampled by this wave (.._resampled).
enum { TICK_ROWS = 0, TICK_FIFO = 1, TICK_FIFO_RS = 2 };

// (Round 5 built a form that keeps frame 1's updated background W1 -- 48 KB at M = 24 -- in LDS instead of redoing it from W0 in frame
// 2's pass: bit-equal and 2.9x slower, two wavefronts per CU where this form runs eight.  profiles/r05_w_in_lds.txt holds the
// measurement; the form is no longer in the source.)
template <int F, int MODE>
__global__ __launch_bounds__(64, F == 256 ? 2 : (F == 128 ? 3 : 4)) void aec_tick_kernel(AecArgs a) {
	__shared__ TLds<F> L;
	using SL = TickLayout<F>;
	constexpr int N = 2 * F, K = F / 64;
	// ---- which leg this wavefront serves.  Rows / per-frame entries: leg = block.  FIFO entry: the leg comes out of a list
	// the PREVIOUS tick's launch sorted (TickOrder above) -- legs that will run two frames first, the short ones last, one
	// list per class b % 8: every XCD gets the same mix whatever pattern the legs' phases follow, and the long legs are
	// started first.
	constexpr bool sched = MODE != TICK_ROWS; // (the leg lists belong to the FIFO entries: aec_launch sets them for those)
	int s = a.first + blockIdx.x, par = 0;
	if (sched) {
		const unsigned c = blockIdx.x & 7u, i = blockIdx.x >> 3;
		par = a.ctl[TickOrder::GLOBAL + TickOrder::PARITY];
		const int front = a.ctl[c * TickOrder::STRIDE + TickOrder::PLACED + 2 * par], back = a.ctl[c * TickOrder::STRIDE + TickOrder::PLACED + 2 * par + 1];
		if (i >= (unsigned)front && i < (unsigned)(a.cap8 - back)) return; // an empty slot between the two ends of the list
		s = a.order[(size_t)par * 8 * a.cap8 + c * a.cap8 + i];
	}
	const int lane = threadIdx.x;
	const int e0 = lane * K; // first element (sample / bin) this lane owns
	// ---- where the frames come from and go to: rows (per-frame / per-tick entries) or the three FIFOs
	constexpr bool fifo = MODE != TICK_ROWS, folded = MODE == TICK_FIFO_RS;
	int nf, ref_frames = 0;
	int2 qm = make_int2(0, 0), qr = make_int2(0, 0), qo = make_int2(0, 0); // (head, level) before this tick
	bool mic_new = false, ref_new = false;                                 // the tick's blocks were taken
	if (fifo) {
		qm = a.fmic.pos[s], qr = a.fref.pos[s], qo = a.fout.pos[s];
		// ms_bufferizer_put of the tick's blocks (a block that does not fit is refused and counted, like mi_fifo_push)
		auto append = [&](const FifoView &q, int2 p, const int16_t *row, int len) -> bool {
			if (len <= 0) return false;
			if (p.y + len > q.cap) {
				if (lane == 0) atomicAdd(q.overflow, 1);
				return false;
			}
			int16_t *ring = q.ring + (size_t)s * q.cap;
			unsigned tail = (unsigned)p.x + (unsigned)p.y;
			if (tail >= (unsigned)q.cap) tail -= (unsigned)q.cap;
			if (((tail | (unsigned)len) & 7u) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
				// whole 16-byte groups (capacities are multiples of the frame size, so of 8): one per lane, 60 lanes at 480 samples
				for (int i = lane; i < (len >> 3); i += 64) {
					unsigned w = tail + 8u * (unsigned)i;
					if (w >= (unsigned)q.cap) w -= (unsigned)q.cap;
					*reinterpret_cast<uint4 *>(ring + w) = *reinterpret_cast<const uint4 *>(row + 8 * i);
				}
			} else {
				for (int i = lane; i < len; i += 64) {
					unsigned w = tail + (unsigned)i;
					if (w >= (unsigned)q.cap) w -= (unsigned)q.cap;
					ring[w] = row[i];
				}
			}
			return true;
		};
		// ---- MSResample folded in (msresample.c:122-179 for this leg's block): out[m den + p] = sum_j table[p][j] x[m + j] over
		// history ++ input, lane (tile, phase) = 8 consecutive input positions of one polyphase row (resample_tile.hpp: the
		// resampler kernel's own tile FIR, so the samples are bit for bit what mi_resampler_process delivers).  Window, table
		// and output staging live in the transform work space, which nothing uses yet.
		const int16_t *mic_row = folded ? nullptr : a.mic_tick + (size_t)s * a.mic_tick_stride;
		// a.run (the *_masked forms): 0 = this leg has no microphone block in this launch (nothing is queued, the resampler's
		// state stays; whatever frames its queue still holds run all the same)
		const int mlen = (a.run && !a.run[s]) ? 0 : a.tick_len;
		if (folded && mlen) {
			constexpr int RS_FILT = 48, RS_R = 8, RS_HIST = RS_FILT - 1;
			const int den = a.rs_den, in_len = a.rs_in_len;
			float *x = reinterpret_cast<float *>(L.zbuf);                    // [xn] history ++ input ++ zero slack
			const int xn = ((RS_HIST + in_len + RS_R + 1) + 3) & ~3;
			int16_t *obuf = reinterpret_cast<int16_t *>(x + xn);             // [in_len * den]
			float4 *tab4 = reinterpret_cast<float4 *>(obuf + ((in_len * den + 7) & ~7));
			const int hq = a.rs_hist_stride >> 2, nq = hq + (in_len >> 2);
			int16_t *hist = a.rs_hist + (size_t)s * a.rs_hist_stride;
			short4 v0 = make_short4(0, 0, 0, 0);
			if (lane < nq)
				v0 = *reinterpret_cast<const short4 *>(lane < hq ? hist + 4 * lane : a.rs_in + (size_t)s * a.rs_in_stride + 4 * (lane - hq));
			for (int i = lane; i < den * RS_FILT / 4; i += 64) tab4[i] = reinterpret_cast<const float4 *>(a.rs_table)[i];
			for (int i = RS_HIST + in_len + lane; i < xn; i += 64) x[i] = 0.f;
			if (lane < nq) {
				const int b = 4 * lane - (lane < hq ? 0 : 4 * hq - RS_HIST);
				if (lane < hq) {
					if (b + 0 < RS_HIST) x[b + 0] = (float)v0.x;
					if (b + 1 < RS_HIST) x[b + 1] = (float)v0.y;
					if (b + 2 < RS_HIST) x[b + 2] = (float)v0.z;
					if (b + 3 < RS_HIST) x[b + 3] = (float)v0.w;
				} else {
					x[b + 0] = (float)v0.x, x[b + 1] = (float)v0.y, x[b + 2] = (float)v0.z, x[b + 3] = (float)v0.w;
				}
			}
			wave_sync();
			const int nlanes = den * ((in_len + RS_R - 1) / RS_R);
			const bool on = lane < nlanes;
			const int tile = on ? lane / den : 0, ph = on ? lane - tile * den : 0, m0 = tile * RS_R;
			f2 acc2[RS_R / 2];
#pragma unroll
			for (int q = 0; q < RS_R / 2; ++q) acc2[q] = (f2){0.f, 0.f};
			fir_tile_rolled<RS_FILT, RS_R>(x + m0, reinterpret_cast<const float *>(tab4 + ph * (RS_FILT / 4)), acc2);
			if (on) {
#pragma unroll
				for (int r = 0; r < RS_R; ++r)
					if (m0 + r < in_len) obuf[(m0 + r) * den + ph] = rs_word2int((r & 1) ? acc2[r / 2].y : acc2[r / 2].x);
			}
			if (lane < hq) { // new history = the last 47 samples of (history ++ input); the pad slot takes the zero slack
				const float *hx = x + in_len + 4 * lane;
				short4 h;
				h.x = (int16_t)hx[0], h.y = (int16_t)hx[1], h.z = (int16_t)hx[2], h.w = (int16_t)hx[3];
				*reinterpret_cast<short4 *>(hist + 4 * lane) = h;
			}
			wave_sync();
			mic_row = obuf;
		}
		int rlen = a.ref_len ? a.ref_len[s] : a.tick_len; // the far end's block may be missing or short this tick
		rlen = rlen < 0 ? 0 : (rlen > a.tick_len ? a.tick_len : rlen);
		mic_new = append(a.fmic, qm, mic_row, mlen);
		ref_new = append(a.fref, qr, a.ref_tick + (size_t)s * a.ref_tick_stride, rlen);
		if (!ref_new) rlen = 0;
		nf = (qm.y + (mic_new ? a.tick_len : 0)) / F;
		if (nf > a.max_frames) nf = a.max_frames;
		ref_frames = (qr.y + rlen) / F; // frames the far end can supply; the rest is silence
		if (ref_frames > nf) ref_frames = nf;
		if (qo.y + nf * F > a.fout.cap) { // no room for the results: nothing runs (counted), the inputs stay queued
			if (lane == 0) atomicAdd(a.fout.overflow, 1);
			nf = 0;
		}
		if (sched && lane == 0) {
			// This leg's place in the NEXT tick's lists: every leg is handed a block every tick, so the frames it will then have
			// follow from what it keeps now (a wrong guess -- a refused block -- only costs placement).  WHICH list: class
			// (c + i) % 8 for the leg at position i of class c -- a class's legs are dealt out over all eight classes every
			// tick, so legs that share a phase cannot stay together on one XCD however the slots were arranged (phase =
			// slot % 8 kept every tick's light legs on ONE XCD with per-class lists that never mixed: +4 % on the launch), and
			// every class keeps nstreams / 8 legs give or take fewer than 16.  Front of the list for two frames, back for
			// fewer: one 64-bit atomic on the destination class's line.
			const int keep = qm.y + (mic_new ? a.tick_len : 0) - nf * F;
			int next = (keep + a.tick_len) / F;
			if (next > a.max_frames) next = a.max_frames;
			const unsigned c = blockIdx.x & 7u, i = blockIdx.x >> 3, d = (c + i) & 7u;
			int *dst = a.order + (size_t)(par ^ 1) * 8 * a.cap8 + d * a.cap8;
			unsigned long long *placed = reinterpret_cast<unsigned long long *>(a.ctl + d * TickOrder::STRIDE + TickOrder::PLACED + 2 * (par ^ 1));
			const unsigned long long old = atomicAdd(placed, next >= 2 ? 1ull : (1ull << 32));
			const unsigned at = next >= 2 ? (unsigned)old : (unsigned)a.cap8 - 1u - (unsigned)(old >> 32);
			if (at < (unsigned)a.cap8) dst[at] = s; // (it always is: the bound at TickOrder::SLACK)
		}
		if (lane == 0) {
			if (a.count_out) a.count_out[s] = (uint8_t)nf;
			const int rf = nf ? ref_frames : 0;
			a.fmic.pos[s] = make_int2((qm.x + nf * F) % a.fmic.cap, qm.y + (mic_new ? a.tick_len : 0) - nf * F);
			a.fref.pos[s] = make_int2((qr.x + rf * F) % a.fref.cap, qr.y + rlen - rf * F);
			a.fout.pos[s] = make_int2(qo.x, qo.y + nf * F);
		}
	} else {
		nf = a.count ? (int)a.count[s] : ((a.run && !a.run[s]) ? 0 : 1);
		if (nf > a.max_frames) nf = a.max_frames;
	}
	if (nf <= 0) return;
	// sample e (< F) of frame f.  FIFO: position f * F + e of the queue = the ring while it is inside what was queued
	// before this tick, the tick's own block after that (read from the caller's row: nothing this wave wrote is read back)
	// Position p of a queue: the ring while p lies inside what was queued before this tick, the tick's own block after
	// that.  (No integer division: one add and two conditional subtractions, p < 2 F <= capacity.)
	auto ring_at = [&](const FifoView &q, int head, int p) -> int {
		unsigned at = (unsigned)head + (unsigned)p;
		if (at >= (unsigned)q.cap) at -= (unsigned)q.cap;
		if (at >= (unsigned)q.cap) at -= (unsigned)q.cap;
		return q.ring[(size_t)s * q.cap + at];
	};
	auto mic_at = [&](int f, int e) -> int {
		if (!fifo) return a.mic[(size_t)s * a.stride + f * F + e];
		const int p = f * F + e;
		// (up-sampled in this launch: the block exists nowhere but in the ring it was queued in)
		if (folded || p < qm.y) return ring_at(a.fmic, qm.x, p);
		return a.mic_tick[(size_t)s * a.mic_tick_stride + (p - qm.y)];
	};
	auto ref_at = [&](int f, int e) -> int {
		if (!fifo) return a.ref[(size_t)s * a.stride + f * F + e];
		if (f >= ref_frames) return 0; // speexec.c:261-272: a frame of zeros when the far end runs short
		const int p = f * F + e;
		if (p < qr.y) return ring_at(a.fref, qr.x, p);
		return a.ref_tick[(size_t)s * a.ref_tick_stride + (p - qr.y)];
	};
	auto out_ptr = [&](int f) -> int16_t * { // where this lane's K cleaned samples of frame f go
		if (!fifo) return a.out + (size_t)s * a.stride + f * F + e0;
		unsigned tail = (unsigned)(qo.x + qo.y) + (unsigned)(f * F); // the output FIFO is popped in ticks: its tail is what is F-aligned
		while (tail >= (unsigned)a.fout.cap) tail -= (unsigned)a.fout.cap;
		return a.fout.ring + (size_t)s * a.fout.cap + tail + e0; // K samples never wrap
	};
	const int M = a.M;
	float *sm = a.small + (size_t)s * a.small_stride;
	const rsrc_t rS = mk_rsrc(sm, (unsigned)a.small_stride * 4u);
	const rsrc_t rX = mk_rsrc(a.X + (size_t)s * (M + 1) * N, (unsigned)((M + 1) * N) * 4u);
	// The two filters of a stream lie back to back in two equal halves: ONE descriptor, a half is chosen by a scalar offset.
	// Which half holds the background and which the foreground is per-stream state (sc.wsel): it is how the library's two
	// filter copies cost nothing here (see pendingFG below).
	const rsrc_t rWF = mk_rsrc(a.WF + (size_t)s * 2 * M * N, (unsigned)(2 * M * N) * 4u);
	const unsigned HALF = (unsigned)(M * N) * 4u;
	const unsigned vb4 = (unsigned)e0 * 4u, vb8 = (unsigned)e0 * 8u; // this lane's byte offset into a float / a bin array
	AecScalars sc = a.scal[s];
	unsigned wo = sc.wsel ? HALF : 0u, fo = HALF - wo; // byte offset of the background / foreground half

	// ---- tables, per-block step / norm, the per-bin state that stays in registers for the whole tick
#pragma unroll
	for (int k = 0; k < K; ++k) {
		L.tw[e0 + k] = a.t.tw[e0 + k];
		L.super[e0 + k] = a.t.super[e0 + k];
		L.perm[e0 + k] = a.t.perm[e0 + k];
	}
	if (lane < M) {
		L.prop[lane] = sm[SL::PROP + lane];
		L.wnorm[lane] = sm[SL::WNORM + lane];
	}
	float2 Eprev[K];
	float p1[K];
	bload_bins<K>(rS, vb8, SL::E * 4, Eprev);
	bload_vec<K>(rS, vb4, SL::POWER1 * 4, p1);
	{ // each lane parks and later fetches its own K elements: no cross-lane traffic, program order suffices
		float t[K];
		bload_vec<K>(rS, vb4, SL::POWER * 4, t);
		store_vec<K>(L.pw + e0, t);
		bload_vec<K>(rS, vb4, SL::EH * 4, t);
		store_vec<K>(L.eh + e0, t);
		bload_vec<K>(rS, vb4, SL::YH * 4, t);
		store_vec<K>(L.yh + e0, t);
		bload_vec<K>(rS, vb4, (SL::LASTY + F) * 4, t);
		store_vec<K>(L.ly[0] + e0, t);
	}
	float pw_F = sm[SL::TAIL + 0], p1_F = sm[SL::TAIL + 1], eh_F = sm[SL::TAIL + 2], yh_F = sm[SL::TAIL + 3];
	bool prop_dirty = false;

	// far end of frame f: pre-emphasis, energy, spectrum of [previous frame | this frame]
	auto prep_far = [&](int f, const float (&xp)[K], float (&xn)[K], float2 (&X0)[K], float &Sxx) {
		float far[K];
#pragma unroll
		for (int k = 0; k < K; ++k) far[k] = (float)ref_at(f, e0 + k);
		float prev = __shfl_up(far[K - 1], 1);
		if (lane == 0) prev = sc.memX;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			xn[k] = far[k] - .9f * prev;
			prev = far[k];
		}
		sc.memX = rdlane(far[K - 1], 63);
		Sxx = WSeq<K>::inner_prod(xn, xn);
		WSYNC();
		store_vec<K>(L.tbuf + e0, xp);
		store_vec<K>(L.tbuf + F + e0, xn);
		w_rfft_forward<F>(L, a.t, X0);
	};

	float2 X0[K], X0B[K]; // far-end spectrum of the frame at hand / of the frame behind it
	float Sxx = 0, SxxB = 0;
	{
		float xp[K], xa[K], xb[K];
		bload_vec<K>(rS, vb4, SL::XPREV * 4, xp);
		prep_far(0, xp, xa, X0, Sxx);
		if (nf > 1) {
			prep_far(1, xa, xb, X0B, SxxB);
			bstore_vec<K>(rS, vb4, SL::XPREV * 4, xb);
		} else {
			bstore_vec<K>(rS, vb4, SL::XPREV * 4, xa);
#pragma unroll
			for (int k = 0; k < K; ++k) X0B[k] = make_float2(0, 0);
		}
	}

	float2 spec2[K]; // frame 2's foreground response, accumulated over frame 1's pass
#pragma unroll
	for (int k = 0; k < K; ++k) spec2[k] = make_float2(0, 0);
	// The library's two filter copies are never passes of their own, and neither moves a byte that was not moving anyway:
	//  * foreground := background (update_foreground).  The NEXT pass over the filter -- this tick's second frame's, or the
	//    first frame's of the next tick (the request then waits in the scalars) -- reads the background's half, which IS the
	//    foreground from now on, and writes the updated background into the OTHER half: the halves swap roles (sc.wsel).
	//    That pass reads one filter instead of two; nothing is copied.  (A pass of its own cost 98 KB per request and made the
	//    ticks in which many legs ask at once -- every ~12th with SURVEY 8(d)'s scene -- 10 % longer than the others.)
	//  * background := foreground (reset_background).  The next pass reads the foreground's half as the background too and
	//    writes the updated blocks to the background's half as it always does.  The block norms the proportional step needs
	//    in between are the foreground's, kept alongside (fgnorm: taken from wnorm whenever the foreground is set).
	// Until that pass one half in HBM is stale; fg_pending / bg_pending say so (mi_aec_get / export_state resolve it).
	bool pendingFG = sc.fg_pending != 0, pendingBG = sc.bg_pending != 0;
	float fgnorm = lane < M ? sm[SL::FGNORM + lane] : 0.f; // lane j: |foreground block j|^2 (one register for the whole tick)
	const bool postfilter = (a.flags & 1) != 0; // MI_AEC_POSTFILTER
	float leakf[2] = {sc.leak_estimate, sc.leak_estimate}; // leak estimate after each frame (post-filter input)
	bool resetf[2] = {false, false};                       // the frame reset the canceller: its echo estimate is zero
	// Frame 1 of a two-frame tick does not write its updated background: frame 2's pass reads W0(j) again and redoes frame
	// 1's gradient step on it (same operands, same operations: the same bits) before its own -- a block read and a few
	// multiply-adds instead of a block written and read (a written byte costs the memory system 1.75 read ones).  What
	// frame 1 still writes: the blocks it constrained (0 and jc) and the last one, whose far-end block leaves the ring.
	bool lazy1 = false;
	int jc1 = -1;
	float2 E1s[K];
	float p1s[K], p1s_F = 0.f;
#pragma unroll
	for (int k = 0; k < K; ++k) E1s[k] = make_float2(0, 0), p1s[k] = 0.f;

	for (int f = 0; f < nf; ++f) {
		// Frame 2's first blocks are asked for NOW (frame 1's pass wrote them long ago): the notch and the proportional step
		// below -- 5 us of serial chains that touch no HBM -- cover their round trip.  (Not frame 1's: with the far-end
		// spectra of both frames and the speculation accumulators live, the 24 registers spill; measured, no gain.)
		const int head = (sc.xhead + M) % (M + 1);
		auto xoff = [&](int j) { return (unsigned)((head + j) % (M + 1)) * (unsigned)(F * 8); };
		v2f pre0[K], pre1[K], pre2[K], pre3[K]; // frame 2's pass takes them over as its first far-end and background blocks
		if (f == 0) {
#pragma unroll
			for (int k = 0; k < K; ++k) pre0[k] = pre1[k] = pre2[k] = pre3[k] = (v2f){0.f, 0.f};
		} else {
			bload_bins<K>(rX, vb8, xoff(1), pre0);
			bload_bins<K>(rX, vb8, xoff(2 < M ? 2 : M), pre1);
			const unsigned wpre = pendingBG ? fo : wo; // frame 1 reset the background: its blocks are the foreground's
			bload_bins<K>(rWF, vb8, wpre, pre2);
			bload_bins<K>(rWF, vb8, wpre + (unsigned)(1 < M ? 1 : 0) * (unsigned)(F * 8), pre3);
		}
		// ---- near end: saturation flag, DC notch (serial IIR), pre-emphasis
		int any_sat;
		{
			float input[K];
			float fin[K];
			bool satl = false;
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const int m = mic_at(f, e0 + k);
				fin[k] = (float)m;
				satl |= (m <= -32000 || m >= 32000);
			}
			any_sat = __any(satl);
			const float radius = a.notch_radius;
			const float den2 = (float)(radius * radius + .7 * (1 - radius) * (1 - radius));
			float v[K];
			w_dc_notch<K>(fin, radius, den2, sc.notch0, sc.notch1, v);
			float vprev = __shfl_up(v[K - 1], 1);
			if (lane == 0) vprev = sc.memD;
#pragma unroll
			for (int k = 0; k < K; ++k) {
				input[k] = v[k] - .9f * vprev;
				vprev = v[k];
			}
			sc.memD = rdlane(v[K - 1], 63);
			store_vec<K>(L.input + e0, input);
		}
		sc.cancel_count++;
		sc.frames++;

		// ---- newest far-end spectrum into the ring; its power spectrum is all the rest of the frame needs of it
		sc.xhead = head;
		bstore_bins<K>(rX, vb8, (unsigned)head * (F * 8), X0);
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
		Xf_F = rdlane(Xf_F, 0);

		// ---- proportional step
		if (sc.adapted) {
			// mdf_adjust_prop, one block per lane: prop = sqrt(1 + |W_j|^2); the maximum is exact in any order; the sum
			// 1 + prop_0 + prop_1 + .. is the library's serial loop, run as a systolic chain over the first M lanes
			WSYNC();
			float p = sqrt_via_double(1.0f + (lane < M ? L.wnorm[lane] : 0.f));
			const float max_sum = wave_tree(lane < M ? p : 1.f, [](float x, float y) { return y > x ? y : x; });
			p = p + .1f * max_sum;
			float run = 1.f;
			for (int i = 0; i < M; ++i) run = dpp_shr1(1.f, run) + p;
			const float prop_sum = rdlane(run, M - 1);
			WSYNC();
			if (lane < M) L.prop[lane] = (.99f * p) / prop_sum;
			prop_dirty = true;
		}
		WSYNC();
		const bool do_grad = (sc.saturated == 0);
		if (!do_grad) sc.saturated--;

		// W += prop p1 conj(X) E, bin by bin (weighted_spectral_mul_conj); bin 0 = (DC, Nyquist): real products with their own steps
		auto grad_with = [&](auto (&w)[K], const auto (&x)[K], float prop, const float2 (&E)[K], const float (&pp)[K], float pp_F) {
#pragma unroll
			for (int k = 0; k < K; ++k) {
				const v2f xv = {x[k].x, x[k].y}, ev = {E[k].x, E[k].y};
				const float Wt = prop * pp[k];
				v2f st = pk_cmul_conj(xv, ev) * (v2f){Wt, Wt}; // Wt (x.x E.x + x.y E.y), Wt ((-x.y) E.x + x.x E.y)
				if (k == 0) {
					const v2f dc = (xv * ev) * (v2f){Wt, prop * pp_F}; // W0 (x.x E.x), WN (x.y E.y)
					if (e0 == 0) st = dc;
				}
				const v2f r = (v2f){w[k].x, w[k].y} + st;
				w[k].x = r.x, w[k].y = r.y;
			}
		};
		auto grad = [&](auto (&w)[K], const auto (&x)[K], float prop) { grad_with(w, x, prop, Eprev, p1, p1_F); };

		// ---- one streaming pass over the blocks (next block's loads in flight).  Block 0 and the round-robin block jc get
		// the AUMDF constraint (IFFT, zero the second half, FFT) where the pass meets them: the blocks are independent of each
		// other, only the sums over j have an order, and it stays ascending.
		const int jc = (M > 1) ? (sc.cancel_count % (M - 1)) + 1 : -1;
		auto constrain = [&](auto (&wv)[K]) {
			float2 w[K];
#pragma unroll
			for (int k = 0; k < K; ++k) w[k] = make_float2(wv[k].x, wv[k].y);
			w_rfft_inverse<F>(L, a.t, w);
			float z[K];
#pragma unroll
			for (int k = 0; k < K; ++k) z[k] = 0.f;
			store_vec<K>(w_time(L) + F + e0, z);
			w_rfft_forward<F>(L, a.t, w, w_time(L));
#pragma unroll
			for (int k = 0; k < K; ++k) wv[k].x = w[k].x, wv[k].y = w[k].y;
		};
		float2 yfg[K], ybgs[K];
#pragma unroll
		for (int k = 0; k < K; ++k) yfg[k] = ybgs[k] = make_float2(0, 0);
		auto norm_of = [&](const auto (&w)[K], int j) {
			float nn = 0;
#pragma unroll
			for (int k = 0; k < K; ++k) nn += w[k].x * w[k].x + w[k].y * w[k].y;
			nn = wave_tree(nn, [](float x, float y) { return x + y; });
			if (lane == 0) L.wnorm[j] = nn; // feeds the NEXT frame's proportional step
		};
		if (f == 0) {
			// frame 1: X, foreground and background; with a second frame behind it also that frame's foreground response
			const bool spec = nf > 1;
			v2f xj[K], xn[K], fg[K], wl[K], xm1[K]; // xm1: the block before xj = frame 2's block at this position (register pairs)
#pragma unroll
			for (int k = 0; k < K; ++k) xj[k] = (v2f){X0[k].x, X0[k].y}, xm1[k] = (v2f){X0B[k].x, X0B[k].y};
			// a filter copy the previous tick's last frame asked for (see pendingFG above): which half is read as what, and where
			// the updated background goes
			const bool carryFG = pendingFG, carryBG = pendingBG;
			pendingFG = pendingBG = false;
			const unsigned fsrc = carryFG ? wo : fo, wsrc = carryBG ? fo : wo, wdst = carryFG ? fo : wo;
			const bool lazy = spec && do_grad && !carryBG && !carryFG && M <= 32; // (the block weights wait in L.prop[32..])
			bload_bins<K>(rX, vb8, xoff(1), xn);
			bload_bins<K>(rWF, vb8, fsrc, fg);
			bload_bins<K>(rWF, vb8, wsrc, wl);
			for (int j = 0; j < M; ++j) {
				v2f xn2[K], fg2[K], wl2[K];
				{ // (behind the last block: its own again, dropped)
					const int jn = j + 1 < M ? j + 1 : j;
					bload_bins<K>(rX, vb8, xoff(jn + 1), xn2);
					bload_bins<K>(rWF, vb8, fsrc + (unsigned)jn * (F * 8), fg2);
					bload_bins<K>(rWF, vb8, wsrc + (unsigned)jn * (F * 8), wl2);
				}
				const bool aumdf = (j == 0 || j == jc);
				if (do_grad) grad(wl, xn, L.prop[j]);
				if (aumdf) constrain(wl);
				if (lazy ? (aumdf || j == M - 1) : (do_grad || aumdf || carryBG || carryFG)) bstore_bins<K>(rWF, vb8, wdst + (unsigned)j * (F * 8), wl);
				cmac_bins<K>(yfg, xj, fg, e0);
				cmac_bins<K>(ybgs, xj, wl, e0);
				if (spec) cmac_bins<K>(spec2, xm1, fg, e0);
				norm_of(wl, j);
#pragma unroll
				for (int k = 0; k < K; ++k) xm1[k] = xj[k], xj[k] = xn[k], xn[k] = xn2[k], fg[k] = fg2[k], wl[k] = wl2[k];
			}
			if (carryFG) { // the halves have swapped roles
				const unsigned t = wo;
				wo = fo, fo = t;
				sc.wsel ^= 1;
			}
			if (lazy) { // what frame 2's pass needs to redo this frame's step: the error spectrum, the steps, the block weights
				lazy1 = true;
				jc1 = jc;
				p1s_F = p1_F;
#pragma unroll
				for (int k = 0; k < K; ++k) E1s[k] = Eprev[k], p1s[k] = p1[k];
				WSYNC();
				if (lane < M) L.prop[32 + lane] = L.prop[lane];
				WSYNC();
			}
		} else {
			// frame 2: X and the background only.  alt = sum_j X(j) W1(j) with W1 the background as frame 1 left it: the
			// foreground response when frame 1 updated the foreground (its half then stays behind as the foreground and the
			// updated blocks go to the other half).
			// Two blocks per iteration: every iteration ends in ONE wait for everything in flight (loads and stores share a
			// counter and may complete out of order with each other, so with a store in flight the only wait is vmcnt(0)), a
			// full memory round trip that the iteration's arithmetic does not cover.  This pass has no foreground blocks in
			// flight, so it has the registers to take the blocks in pairs: half as many round trips.
			v2f xj[K], xn3[K]; // register pairs: a bin moves with one instruction
			v2f(&xn)[K] = pre0, (&xn2)[K] = pre1, (&wl)[K] = pre2, (&wl2)[K] = pre3;
#pragma unroll
			for (int k = 0; k < K; ++k) xj[k] = (v2f){X0[k].x, X0[k].y};
			auto xclamp = [&](int i) { return xoff(i < M ? i : M); };
			auto wclamp = [&](int i) { return (unsigned)(i < M ? i : M - 1) * (unsigned)(F * 8); };
			const bool carryBG = pendingBG, carryFG = pendingFG; // frame 1 reset the background / updated the foreground
			pendingBG = false;
			if (carryFG) { // the speculated foreground response is void: its registers take alt
#pragma unroll
				for (int k = 0; k < K; ++k) spec2[k] = make_float2(0, 0);
			}
			const unsigned wsrc = carryBG ? fo : wo, wdst = carryFG ? fo : wo;
			// a block's redo needs the far-end block two places on: four landed far-end blocks per pair, the next two in flight.
			// The fourth of the first pair is waited for HERE, once, so that no iteration waits in its middle.
			bload_bins<K>(rX, vb8, xclamp(3), xn3);
			__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
			const bool redo = lazy1 && !carryBG; // frame 1 left its updated blocks unwritten (all but 0, jc1 and the last)
			const bool keepW1 = redo && carryFG;  // ... and then made them the foreground: they must exist in that half
			auto block = [&](int j, v2f (&w)[K], const v2f (&xa)[K], const v2f (&xb)[K], const v2f (&xc)[K]) { // X(j), X(j+1), X(j+2)
				if (redo && j != 0 && j != jc1 && j != M - 1) {
					grad_with(w, xc, L.prop[32 + j], E1s, p1s, p1s_F);
					if (keepW1) bstore_bins<K>(rWF, vb8, wo + (unsigned)j * (F * 8), w);
				}
				if (carryFG) cmac_bins<K>(spec2, xa, w, e0);
				const bool aumdf = (j == 0 || j == jc);
				if (do_grad) grad(w, xb, L.prop[j]);
				if (aumdf) constrain(w);
				if (do_grad || aumdf || carryBG || carryFG || lazy1) bstore_bins<K>(rWF, vb8, wdst + (unsigned)j * (F * 8), w);
				cmac_bins<K>(ybgs, xa, w, e0);
				norm_of(w, j);
			};
			for (int j = 0; j < M; j += 2) { // an odd block count: the last round serves one block (no third copy of `block` in the code)
				v2f xa[K], xb[K], wa[K], wb[K]; // the next pair: X(j+3), X(j+4), W(j+2), W(j+3) (clamped at the end: dropped)
				bload_bins<K>(rX, vb8, xclamp(j + 4), xa);
				bload_bins<K>(rX, vb8, xclamp(j + 5), xb);
				bload_bins<K>(rWF, vb8, wsrc + wclamp(j + 2), wa);
				bload_bins<K>(rWF, vb8, wsrc + wclamp(j + 3), wb);
				block(j, wl, xj, xn, xn2);
				if (j + 1 < M) block(j + 1, wl2, xn, xn2, xn3);
#pragma unroll
				for (int k = 0; k < K; ++k) xj[k] = xn2[k], xn[k] = xn3[k], xn2[k] = xa[k], xn3[k] = xb[k], wl[k] = wa[k], wl2[k] = wb[k];
			}
#pragma unroll
			for (int k = 0; k < K; ++k) yfg[k] = spec2[k];
			pendingFG = false;
			lazy1 = false;
			if (carryFG) {
				const unsigned t = wo;
				wo = fo, fo = t;
				sc.wsel ^= 1;
			}
		}

		// ---- time-domain responses
		float efg[K], ybg[K], e1[K], e2[K], dresp[K], input[K];
		w_rfft_inverse<F>(L, a.t, yfg);
		load_vec<K>(w_time(L) + F + e0, efg);
		load_vec<K>(L.input + e0, input);
#pragma unroll
		for (int k = 0; k < K; ++k) e1[k] = input[k] - efg[k];
		w_rfft_inverse<F>(L, a.t, ybgs);
		load_vec<K>(w_time(L) + F + e0, ybg);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			e2[k] = input[k] - ybg[k];
			dresp[k] = efg[k] - ybg[k];
		}
		float Sff, Dbf, See;
		WSeq<K>::inner_prod3(e1, e1, dresp, dresp, e2, e2, Sff, Dbf, See);
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
		if (update_foreground) {
			sc.fg_updates++;
			sc.Davg1 = sc.Davg2 = 0;
			sc.Dvar1 = sc.Dvar2 = 0;
			pendingFG = true; // the next pass over the filter (this tick's or the next one's) carries the copy out
			WSYNC();
			if (lane < M) fgnorm = L.wnorm[lane];
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
				WSYNC();
				if (lane < M) L.wnorm[lane] = fgnorm;
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
			w_deemphasis<K>(d, sc.memE, tout);
#pragma unroll
			for (int k = 0; k < K; ++k) out_i[k] = word2int(tout[k]);
		}
		float Sey, Syy, Sdd;
		WSeq<K>::inner_prod3(e2, ybg, ybg, ybg, input, input, Sey, Syy, Sdd);
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
			w_rfft_forward<F>(L, a.t, Ecur);
			store_vec<K>(L.tbuf + e0, z);
			store_vec<K>(L.tbuf + F + e0, ybg);
			w_rfft_forward<F>(L, a.t, Ycur);
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
		// the Nyquist powers live in lane 0; everyone needs them for the ordered sums below
		Rf_F = rdlane(Rf_F, 0);
		Yf_F = rdlane(Yf_F, 0);
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
		int16_t *op = out_ptr(f);
		auto emit = [&](const int (&o)[K]) { // to the post-filter phase through LDS, else straight out
#pragma unroll
			for (int k = 0; k < K; ++k) {
				if (postfilter) L.outf[f][e0 + k] = (int16_t)o[k];
				else op[k] = (int16_t)o[k];
			}
		};
		if (sc.screwed_up >= 50) { // speex_echo_state_reset
			float z[K];
			float2 z2[K];
#pragma unroll
			for (int k = 0; k < K; ++k) z[k] = 0.f, z2[k] = make_float2(0, 0);
			for (int j = 0; j < M; ++j) {
				bstore_bins<K>(rWF, vb8, (unsigned)j * (F * 8), z2);
				bstore_bins<K>(rWF, vb8, HALF + (unsigned)j * (F * 8), z2);
			}
			for (int j = 0; j <= M; ++j) bstore_bins<K>(rX, vb8, (unsigned)j * (F * 8), z2);
#pragma unroll
			for (int k = 0; k < K; ++k) {
				p1[k] = 1.0f;
				Eprev[k] = make_float2(0, 0);
				spec2[k] = make_float2(0, 0);
			}
			lazy1 = false; // both halves are zero in HBM: nothing of frame 1's step is left to redo
			store_vec<K>(L.pw + e0, z);
			store_vec<K>(L.eh + e0, z);
			store_vec<K>(L.yh + e0, z);
			store_vec<K>(L.ly[f + 1] + e0, z);
			pw_F = eh_F = yh_F = 0.f;
			p1_F = 1.0f;
			WSYNC();
			if (lane < M) L.wnorm[lane] = 0;
			fgnorm = 0.f;
			bstore_vec<K>(rS, vb4, SL::LASTY * 4, z);
			bstore_vec<K>(rS, vb4, (SL::LASTY + F) * 4, z);
			bstore_vec<K>(rS, vb4, (SL::LASTY + (2) * F) * 4, z);
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
			if (f) resetf[1] = true, leakf[1] = sc.leak_estimate;
			else resetf[0] = true, leakf[0] = sc.leak_estimate;
			emit(out_i);
			if (f + 1 < nf) { // the next frame starts from the reset far-end state
				float xn[K];
				prep_far(f + 1, z, xn, X0, Sxx);
				bstore_vec<K>(rS, vb4, SL::XPREV * 4, xn);
			} else {
				bstore_vec<K>(rS, vb4, SL::XPREV * 4, z);
			}
			continue;
		}
		if (See < (float)(N * 100)) See = (float)(N * 100);
		float Sxx2 = Sxx + Sxx; // sic: the library accumulates the far-end energy a second time here

		// ---- far-end power, leak estimate
		float pw[K];
		load_vec<K>(L.pw + e0, pw);
#pragma unroll
		for (int k = 0; k < K; ++k) pw[k] = a.ss_1 * pw[k] + 1 + a.ss * Xf[k];
		store_vec<K>(L.pw + e0, pw);
		pw_F = a.ss_1 * pw_F + 1 + a.ss * Xf_F;
		float Ehd[K], Yhd[K], Ehd_F, Yhd_F;
		{
			float eh[K], yh[K];
			load_vec<K>(L.eh + e0, eh);
			load_vec<K>(L.yh + e0, yh);
#pragma unroll
			for (int k = 0; k < K; ++k) {
				Ehd[k] = Rf[k] - eh[k];
				Yhd[k] = Yf[k] - yh[k];
				eh[k] = (1 - a.spec_average) * eh[k] + a.spec_average * Rf[k];
				yh[k] = (1 - a.spec_average) * yh[k] + a.spec_average * Yf[k];
			}
			store_vec<K>(L.eh + e0, eh);
			store_vec<K>(L.yh + e0, yh);
		}
		Ehd_F = Rf_F - eh_F;
		Yhd_F = Yf_F - yh_F;
		eh_F = (1 - a.spec_average) * eh_F + a.spec_average * Rf_F;
		yh_F = (1 - a.spec_average) * yh_F + a.spec_average * Yf_F;
		float Pey = 1.0f, Pyy = 1.0f;
		Pey = Pey + Ehd_F * Yhd_F;
		Pyy = Pyy + Yhd_F * Yhd_F;
		WSeq<K>::dot_desc2(Pey, Ehd, Yhd, Pyy, Yhd, Yhd, Pey, Pyy);
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
		int m_here = M; // (converted HERE: hoisted to the kernel's entry the float lived -- spilled to scratch -- through the whole tick)
		asm volatile("" : "+s"(m_here));
		if (!sc.adapted && sc.sum_adapt > (float)m_here && sc.leak_estimate * Syy > .03f * Syy) sc.adapted = 1;

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

		// ---- echo estimate of this frame for the residual-echo stage (ring of three frames)
		{
			float ln[K], lprev[K];
			load_vec<K>(L.ly[f] + e0, lprev);
#pragma unroll
			for (int k = 0; k < K; ++k) ln[k] = sc.adapted ? (float)(mic_at(f, e0 + k) - out_i[k]) : lprev[k];
			store_vec<K>(L.ly[f + 1] + e0, ln);
		}
		if (f) leakf[1] = sc.leak_estimate;
		else leakf[0] = sc.leak_estimate;
		emit(out_i);
#pragma unroll
		for (int k = 0; k < K; ++k) X0[k] = X0B[k]; // the frame behind this one, if any
		Sxx = SxxB;
	}

	// ---- the tick's state back to HBM, once
	bstore_bins<K>(rS, vb8, SL::E * 4, Eprev);
	bstore_vec<K>(rS, vb4, SL::POWER1 * 4, p1);
	{
		float t[K];
		load_vec<K>(L.pw + e0, t);
		bstore_vec<K>(rS, vb4, SL::POWER * 4, t);
		load_vec<K>(L.eh + e0, t);
		bstore_vec<K>(rS, vb4, SL::EH * 4, t);
		load_vec<K>(L.yh + e0, t);
		bstore_vec<K>(rS, vb4, SL::YH * 4, t);
	}
	WSYNC();
	if (lane < M) {
		sm[SL::WNORM + lane] = L.wnorm[lane];
		sm[SL::FGNORM + lane] = fgnorm;
		if (prop_dirty) sm[SL::PROP + lane] = L.prop[lane];
	}
	{ // echo estimate of the last frame's pair [older | newest]
		float lo[K], ln[K];
		load_vec<K>(L.ly[nf - 1] + e0, lo);
		load_vec<K>(L.ly[nf] + e0, ln);
		if (nf > 1 ? resetf[1] : resetf[0]) {
#pragma unroll
			for (int k = 0; k < K; ++k) lo[k] = 0.f;
		}
		bstore_vec<K>(rS, vb4, SL::LASTY * 4, lo);
		bstore_vec<K>(rS, vb4, (SL::LASTY + F) * 4, ln);
	}
	if (lane == 0) {
		sm[SL::TAIL + 0] = pw_F;
		sm[SL::TAIL + 1] = p1_F;
		sm[SL::TAIL + 2] = eh_F;
		sm[SL::TAIL + 3] = yh_F;
	}
	if (!postfilter) {
		sc.fg_pending = pendingFG ? 1 : 0, sc.bg_pending = pendingBG ? 1 : 0;
		if (lane == 0) a.scal[s] = sc;
		return;
	}
	WSYNC(); // the parked arrays are re-used from here on

	// per-bin state of the tick, in registers
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
	float old_ps_b = 0, zeta_b = 0;
	if (lane < NB_BANDS) {
		old_ps_b = sm[SL::OLDPS_B + lane];
		zeta_b = sm[SL::ZETA_B + lane];
	}
	float *pl = L.spec, *pr = L.spec + F;
	float *bandv = L.band();
	float *lvec = L.vec();

	for (int f = 0; f < nf; ++f) {
		sc.nb_adapt++;
		if (sc.nb_adapt > 20000) sc.nb_adapt = 20000;
		sc.min_count++;
		float beta = 1.0f / sc.nb_adapt;
		if (beta < .03f) beta = .03f;
		const float beta_1 = 1.0f - beta;
		const float leak = f ? leakf[1] : leakf[0];
		const bool was_reset = f ? resetf[1] : resetf[0];

		// residual echo spectrum (speex_echo_get_residual)
		{
			float lo[K], ln[K];
			load_vec<K>(L.ly[f] + e0, lo);
			load_vec<K>(L.ly[f + 1] + e0, ln);
#pragma unroll
			for (int k = 0; k < K; ++k) lo[k] = h0[k] * (was_reset ? 0.f : lo[k]), ln[k] = h1[k] * ln[k];
			WSYNC();
			store_vec<K>(L.tbuf + e0, lo);
			store_vec<K>(L.tbuf + F + e0, ln);
		}
		float2 Yr[K];
		w_rfft_forward<F>(L, a.t, Yr);
		const float leak2 = (leak > .5) ? 1.f : 2 * leak;
		float res[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const float r = (e0 + k == 0) ? Yr[k].x * Yr[k].x : Yr[k].x * Yr[k].x + Yr[k].y * Yr[k].y;
			res[k] = (float)(int32_t)(leak2 * r);
		}
		const float res0 = rdlane(res[0], 0);
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
		int16_t *op = out_ptr(f);
		{
			float xcur[K], a0[K], a1[K];
#pragma unroll
			for (int k = 0; k < K; ++k) {
				xcur[k] = (float)L.outf[f][e0 + k];
				a0[k] = inb[k] * w0[k];
				a1[k] = xcur[k] * w1[k];
				inb[k] = xcur[k];
			}
			store_vec<K>(L.tbuf + e0, a0);
			store_vec<K>(L.tbuf + F + e0, a1);
		}
		float2 ft[K];
		w_rfft_forward<F>(L, a.t, ft);
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
			if (lane < NB_BANDS) band_sum3<F>(a.t, lane, L.spec, L.tbuf, np, bandv[lane], bandv[NB_BANDS + lane], bandv[2 * NB_BANDS + lane]);
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
		float post_b = 0, prior_b = 0, ps_b = 0;
		if (lane < NB_BANDS) {
			ps_b = bandv[NB_BANDS + lane];
			if (sc.nb_adapt == 1) old_ps_b = ps_b;
			snr(ps_b, bandv[2 * NB_BANDS + lane], bandv[lane], old_ps_b, post_b, prior_b);
		}
		WSYNC();
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int b = e0 + k;
			if (b == 0 || b >= F - 1) zeta[k] = .7f * zeta[k] + .3f * prior[k];
			else zeta[k] = .7f * zeta[k] + .15f * prior[k] + .075f * lvec[b - 1] + .075f * lvec[b + 1];
		}
		if (lane < NB_BANDS) zeta_b = .7f * zeta_b + .3f * prior_b;
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
		w_rfft_inverse<F>(L, a.t, ft);
		{
			float lo[K], hi[K];
			load_vec<K>(w_time(L) + e0, lo);
			load_vec<K>(w_time(L) + F + e0, hi);
#pragma unroll
			for (int k = 0; k < K; ++k) {
				op[k] = word2int(ob[k] + lo[k] * w0[k]);
				ob[k] = hi[k] * w1[k];
			}
		}
		WSYNC();
	}

	bstore_vec<K>(rS, vb4, SL::ECHON * 4, en);
	bstore_vec<K>(rS, vb4, SL::INBUF * 4, inb);
	bstore_vec<K>(rS, vb4, SL::S_ * 4, S);
	bstore_vec<K>(rS, vb4, SL::SMIN * 4, Smin);
	bstore_vec<K>(rS, vb4, SL::STMP * 4, Stmp);
	bstore_vec<K>(rS, vb4, SL::NOISE * 4, noise);
	bstore_vec<K>(rS, vb4, SL::OLDPS * 4, old_ps);
	bstore_vec<K>(rS, vb4, SL::ZETA * 4, zeta);
	bstore_vec<K>(rS, vb4, SL::OUTBUF * 4, ob);
	if (lane < NB_BANDS) {
		sm[SL::OLDPS_B + lane] = old_ps_b;
		sm[SL::ZETA_B + lane] = zeta_b;
	}

	sc.fg_pending = pendingFG ? 1 : 0, sc.bg_pending = pendingBG ? 1 : 0;
	if (lane == 0) a.scal[s] = sc;
}
