// resample_tile.hpp -- the tile FIR of the polyphase resampler kernels (included INSIDE the anonymous namespace of
// resample.hip, whose kernels are built from it, and of aec.hip, whose tick kernel runs the 16k -> 48k up-sampler of a
// call leg as its first phase: MSResample folded into MSSpeexEC's launch).
#pragma once

// WORD2INT of the library: floor(.5 + x) evaluated in double, clamped to int16 (x < -32767.5 -> -32768,
// x > 32766.5 -> 32767).  Exact float form: clamp to [-32768, 32767], then (floor(2x) + 1) >> 1 -- 2x is
// exact, and floor((floor(2x) + 1) / 2) == floor(x + .5) for every real x.
__device__ __forceinline__ int16_t rs_word2int(float x) {
	x = __builtin_amdgcn_fmed3f(x, -32768.f, 32767.f);
	return (int16_t)(((int)floorf(x + x) + 1) >> 1);
}

typedef float f2 __attribute__((ext_vector_type(2)));

// LDS hand-over inside ONE wavefront: the LDS unit executes a wave's instructions in order, so only the compiler has
// to be kept from moving accesses across this point (no s_barrier: the waves of a workgroup are independent here)
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// acc += splat(t.lo or t.hi) * w on both halves.  Written as asm so the FILT x R/2 issue order below is
// the one executed: left to itself the scheduler finishes one accumulator at a time and spills the window.
template <int HI>
__device__ __forceinline__ void pk_fma_splat(f2 &acc, const f2 t, const f2 w) {
	if (HI)
		asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(t), "v"(w));
	else
		asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(t), "v"(w));
}

// (a.hi, b.lo): the window pair at an odd offset, one issue slot
__device__ __forceinline__ f2 pk_odd_pair(const f2 a, const f2 b) {
	f2 d;
	asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
	return d;
}

// acc2[q] += sum_j t[j] * (xwin[j + 2q], xwin[j + 2q + 1]) for j < FILT: R = 8 consecutive positions of one polyphase
// row.  The window slides through registers 8 taps at a time (16 floats live, the next 8 in flight); v_pk_fma_f32 does
// two positions per issue slot: the tap is broadcast by op_sel, even-offset window pairs are register pairs as loaded
// (xwin is 16-byte aligned), odd-offset pairs cost one v_pk_mov_b32 each.  Reads xwin[0 .. FILT+R-1].
template <int FILT, int R>
__device__ __forceinline__ void fir_tile(const float *xwin, const f2 (&t2)[FILT / 2], f2 (&acc2)[R / 2]) {
	static_assert(R == 8 && FILT % 8 == 0, "window = FILT+R-1 samples read as 16-byte groups");
	const float4 *wp = reinterpret_cast<const float4 *>(xwin);
	float4 c0 = wp[0], c1 = wp[1], c2 = wp[2], c3 = wp[3];
	f2 od[7]; // odd-offset pairs (w[2i+1], w[2i+2]) of the 16 live floats
#pragma unroll
	for (int c = 0; c < FILT / 8; ++c) {
		const f2 ev[8] = {(f2){c0.x, c0.y}, (f2){c0.z, c0.w}, (f2){c1.x, c1.y}, (f2){c1.z, c1.w},
		                  (f2){c2.x, c2.y}, (f2){c2.z, c2.w}, (f2){c3.x, c3.y}, (f2){c3.z, c3.w}};
#pragma unroll
		for (int i = (c == 0 ? 0 : 3); i < 7; ++i) od[i] = pk_odd_pair(ev[i], ev[i + 1]);
#pragma unroll
		for (int jj = 0; jj < 8; ++jj) {
			const f2 tp2 = t2[(8 * c + jj) / 2];
#pragma unroll
			for (int q = 0; q < R / 2; ++q) {
				const int k = jj + 2 * q; // 0..13 within the 16 live floats
				const f2 wk = (k & 1) ? od[k / 2] : ev[k / 2];
				if (jj & 1)
					pk_fma_splat<1>(acc2[q], tp2, wk);
				else
					pk_fma_splat<0>(acc2[q], tp2, wk);
			}
		}
		od[0] = od[4], od[1] = od[5], od[2] = od[6];
		c0 = c2, c1 = c3;
		if (c + 1 < FILT / 8) c2 = wp[2 * c + 4], c3 = wp[2 * c + 5];
	}
}


// The same sums in the same order (bit-identical results), as a LOOP over the groups of eight taps with the taps read from
// LDS group by group instead of living in 24 register pairs: a fifth of the code and registers of fir_tile, for a caller
// whose time goes elsewhere (the canceller's tick kernel, which runs a leg's up-sampler as its first phase and has no
// instruction-cache room for 192 unrolled packed FMAs).  taps: this lane's polyphase row, 16-byte aligned, in LDS.
// Reads xwin[0 .. FILT+R+7] (the last group's look-ahead is never used).
template <int FILT, int R>
__device__ __forceinline__ void fir_tile_rolled(const float *xwin, const float *taps, f2 (&acc2)[R / 2]) {
	static_assert(R == 8 && FILT % 8 == 0, "window = FILT+R-1 samples read as 16-byte groups");
	const float4 *wp = reinterpret_cast<const float4 *>(xwin), *tp = reinterpret_cast<const float4 *>(taps);
	float4 c0 = wp[0], c1 = wp[1], c2 = wp[2], c3 = wp[3];
#pragma nounroll
	for (int c = 0; c < FILT / 8; ++c) {
		const float4 ta = tp[2 * c], tb = tp[2 * c + 1];
		const f2 t2[4] = {(f2){ta.x, ta.y}, (f2){ta.z, ta.w}, (f2){tb.x, tb.y}, (f2){tb.z, tb.w}};
		const f2 ev[8] = {(f2){c0.x, c0.y}, (f2){c0.z, c0.w}, (f2){c1.x, c1.y}, (f2){c1.z, c1.w},
		                  (f2){c2.x, c2.y}, (f2){c2.z, c2.w}, (f2){c3.x, c3.y}, (f2){c3.z, c3.w}};
		f2 od[7];
#pragma unroll
		for (int i = 0; i < 7; ++i) od[i] = pk_odd_pair(ev[i], ev[i + 1]);
#pragma unroll
		for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
			for (int q = 0; q < R / 2; ++q) {
				const int k = jj + 2 * q;
				const f2 wk = (k & 1) ? od[k / 2] : ev[k / 2];
				if (jj & 1)
					pk_fma_splat<1>(acc2[q], t2[jj / 2], wk);
				else
					pk_fma_splat<0>(acc2[q], t2[jj / 2], wk);
			}
		}
		c0 = c2, c1 = c3;
		c2 = wp[2 * c + 4], c3 = wp[2 * c + 5];
	}
}
