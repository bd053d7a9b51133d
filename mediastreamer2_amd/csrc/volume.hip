// volume.hip -- batched MSVolume chunk (meter, echo limiter, AGC, noise gate,
// Q12 gain) for gfx950.  Built with -ffp-contract=off: the control chain of
// src/audiofilters/msvolume.c is float32 evaluated unfused in source order on
// the reference's x86-64 build, and the integer gain (hence every output
// sample) depends on it bit for bit.
//
// Per stream and chunk this is volume_process's loop body (msvolume.c:480-513):
//   update_energy :388-407  sequential float32 sum of squares + integer peak
//   volume_echo_avoider_process :201-238 (peer energy), volume_agc_process :172-184,
//   volume_noise_gate_process :240-260, apply_gain :409-445 (ramp, Q12 integer
//   gain with C truncating division, symmetric +-32767 clamp, optional DC removal).
//
// Mapping: one wavefront owns SPB=4 streams (4096 streams -> one wave per SIMD).
// (A) all lanes stage the chunks into LDS with 16-byte coalesced loads and do
// everything that is order-free on the way: (float)(v*v) per sample into a
// second LDS array, integer peak and DC sum through LDS atomics; (B) one lane
// per stream adds the squares in sample order -- the float accumulation ORDER
// is part of the reference's result, so this is a chain of n
// dependent v_add_f32 fed by 16-byte LDS reads and nothing else -- and runs the
// scalar control chain; (C) all lanes apply the integer gain and store 16 bytes
// per lane.  Streams
// whose gain is exactly 1 (and no DC removal) are not written back, as in the
// reference (msvolume.c:440).  HBM traffic: 2 B/sample read, <= 2 B/sample
// written, + ~100 B of per-stream state.
#include "common.hpp"
#include <algorithm>
#include <vector>
#include <atomic>

#pragma clang fp contract(off)

namespace {

constexpr int SPB = 4;       // streams per block (one wavefront)
constexpr int VTHREADS = 64;

struct VolArgs {
	int16_t *samples;
	const int32_t *nsamples_per_stream; // or null
	const mi_volume_params *params;
	mi_volume_state *state;
	float *energy[2];  // double buffer: peers read the PREVIOUS launch's energy (msvolume.c:206-207 reads its peer's field)
	// a stream whose peer is MI_VOLUME_PEER_EXTERNAL reads the energy of the SAME index in another batch (mi_volume_set_peer_batch),
	// as that batch's last launch left it: the plugin's fused leg keeps volsend in one batch and meters volrecv in another
	const mi_volume_state *ext_state;
	const int *parity; // which of the two holds the previous launch's values; flipped on the device after each launch,
	                   // so a captured hipGraph replays correctly (a host-side flip would be frozen into the graph)
	int nstreams, nsamples, stride, sample_rate, pitch_dw, pitch_f;
	int first; // the launch serves streams [first, nstreams): block b owns SPB streams from first + b * SPB on
	// the one-second maximum of the smoothed energy behind MS_VOLUME_GET_MAX (ortp_extremum_record_max on every
	// process(), msvolume.c:115,:404): x = current maximum, y = ms since the window started (< 0: not started).  The
	// stream's own chunks are its clock, as the ticker's time is the filter's (one chunk per tick).
	float2 *win;
	// src.ring != NULL: the chunk is popped from a device FIFO (all-or-nothing, zeros when it holds less: what
	// mi_fifo_pop(..., zero_fill) delivers) and the result is written to `samples` -- no separate pop launch, no copy
	FifoView src;
	int src_dry_skips; // MI_VOLMIX_DRY_SKIPS: a stream whose queue holds less than the chunk is left alone (no meter update, no output)
};

__device__ __forceinline__ int sat16(int v) { return (v > 32767) ? 32767 : ((v < -32767) ? -32767 : v); }

// What update_energy leaves behind and everything volume_process derives from it for one chunk (msvolume.c:388-407,
// :172-260, :409-445's gain ramp): the smoothed energy, the echo limiter / AGC / noise gate targets, the ramped gain as the
// Q12 integer the samples are scaled with, the DC estimate, the one-second maximum.  acc = the float32 sum of the squares
// in sample order, pk / dcsum = integer peak and sum of the chunk.  One lane per stream; shared by volume_kernel and the
// fused volume + conference mix kernel below.
struct VolCtl {
	int intgain, dcoff, mode; // mode 0: samples untouched (gain exactly 1), 1: gain, 2: DC removal + gain
};
__device__ __forceinline__ VolCtl volume_control(const mi_volume_params &p, mi_volume_state &st, float peer_energy, float acc, int n, int pk,
                                                 int dcsum, int sample_rate, float2 &win) {
	const float max_e = (32768 * 0.7f);
	VolCtl o;
	const float en = (float)((sqrt((double)(acc / n)) + 1) / (double)max_e);
	st.energy = (en * 0.2f) + st.energy * (1.0f - 0.2f);
	st.level_pk = (float)pk / max_e;
	st.instant_energy = en;

	float target = p.static_gain;
	if (p.peer != -1) { // echo limiter
		const float peer_e = peer_energy, peer_pk = peer_e;
		if (peer_pk > st.lt_speaker_en) st.lt_speaker_en = peer_pk;
		else st.lt_speaker_en = (0.005f * peer_pk) + (0.995f * st.lt_speaker_en);
		const float ratio = (st.energy / (st.lt_speaker_en + p.ea_thres));
		if (peer_e > p.ea_thres) {
			if (ratio > p.ea_transmit_thres) {
				st.target_gain = p.static_gain;
				st.fast_upramp = 1;
			} else {
				st.target_gain = p.static_gain / (1 + (peer_e * p.force));
				st.sustain_dur = p.sustain_time;
			}
		} else {
			if (st.sustain_dur > 0) {
				st.sustain_dur -= (n * 1000) / sample_rate;
			} else {
				st.target_gain = p.static_gain;
				st.fast_upramp = 1;
			}
		}
		target = st.target_gain;
	}
	if (p.agc_enabled) target /= (0.5f + st.level_pk) / 1;
	if (p.noise_gate_enabled) {
		float tgain = p.ng_floorgain;
		if (st.instant_energy > p.ng_threshold) {
			st.ng_noise_dur = p.ng_cut_time;
			tgain = 1.0f;
		} else if (st.ng_noise_dur > 0) {
			st.ng_noise_dur -= (n * 1000) / sample_rate;
			tgain = 1.0f;
		}
		st.ng_gain = st.ng_gain * 0.75f + tgain * 0.25f;
	}
	// apply_gain: multiplicative ramp toward target
	if (st.gain < target) {
		if (st.gain < p.ng_floorgain) st.gain = p.ng_floorgain;
		st.gain *= 1 + (st.fast_upramp ? p.vol_fast_upramp : p.vol_upramp);
		if (st.gain > target) st.gain = target;
	} else if (st.gain > target) {
		st.gain *= 1 - p.vol_downramp;
		if (st.gain < target) st.gain = target;
		st.fast_upramp = 0;
	}
	const float gain = st.gain * st.ng_gain;
	o.intgain = (int32_t)(gain * 4096);
	o.dcoff = st.dc_offset;
	if (p.remove_dc) {
		o.mode = 2;
		st.dc_offset = (st.dc_offset * 7 + dcsum * 2 / (2 * n)) / 8;
	} else {
		o.mode = (gain != 1) ? 1 : 0;
	}
	{ // ortp_extremum_record_max(&v->max, curtime, v->energy), period 1000 ms
		float2 w = win;
		if (w.y >= 0) {
			w.y += (float)((n * 1000) / sample_rate);
			if (w.y > 1000.f) w.y = -1.f; // (int)(now - start) > period: the old maximum is dropped
		}
		if (w.y < 0) w = make_float2(st.energy, 0.f);
		if (st.energy > w.x) w.x = st.energy;
		win = w;
	}
	return o;
}

__global__ __launch_bounds__(VTHREADS) void volume_kernel(VolArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint32_t *rows = reinterpret_cast<uint32_t *>(smem);                              // [SPB][pitch_dw] packed int16
	float *sq = reinterpret_cast<float *>(smem + (size_t)SPB * a.pitch_dw * 4);       // [SPB][pitch_f] (float)(v*v)
	__shared__ int s_intgain[SPB], s_dcoff[SPB], s_mode[SPB], s_n[SPB], s_pk[SPB], s_dc[SPB];
	__shared__ int s_head[SPB]; // FIFO source: index of the chunk's first sample in the stream's ring, -1 = not enough queued
	const bool from_fifo = a.src.ring != nullptr;

	const int tid = threadIdx.x;
	const int s0 = a.first + blockIdx.x * SPB;
	const int nloc = min(SPB, a.nstreams - s0);
	// per-stream parameters and state: requested first, needed only in phase B
	mi_volume_params p;
	mi_volume_state st;
	float peer_energy = 0;
	if (tid < nloc) {
		p = a.params[s0 + tid];
		st = a.state[s0 + tid];
		if (p.peer >= 0) peer_energy = a.energy[*a.parity][p.peer];
		else if (p.peer == MI_VOLUME_PEER_EXTERNAL && a.ext_state) peer_energy = a.ext_state[s0 + tid].energy;
		else if (p.peer == MI_VOLUME_PEER_EXTERNAL) p.peer = -1; // (no peer batch set, or it has been destroyed: no peer, no limiter)
	}
	if (tid < SPB) {
		int n = 0;
		if (tid < nloc) n = a.nsamples_per_stream ? a.nsamples_per_stream[s0 + tid] : a.nsamples;
		s_n[tid] = min(max(n, 0), a.nsamples);
		s_pk[tid] = 0;
		s_dc[tid] = 0;
		s_head[tid] = -1;
		if (from_fifo && tid < nloc) { // ms_bufferizer_read, all-or-nothing (msqueue.c:83); the position moves on at once
			const int2 q = a.src.pos[s0 + tid];
			if (q.y >= a.nsamples) {
				s_head[tid] = q.x;
				a.src.pos[s0 + tid] = make_int2((q.x + a.nsamples) % a.src.cap, q.y - a.nsamples);
			} else if (a.src_dry_skips) {
				s_n[tid] = 0; // volume_process finds no whole chunk in its bufferizer and does nothing (msvolume.c:480-486)
			}
		}
	}
	__syncthreads();

	// ---- (A) stage chunks
	const bool vec = ((a.stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.samples) & 15) == 0);
	if (vec) {
		const int oct = (a.nsamples + 7) >> 3; // 16-byte groups per row
		constexpr int NB = 4;                  // loads in flight per lane
		for (int base = 0; base < nloc * oct; base += NB * VTHREADS) {
			uint4 v[NB];
			int sl[NB], q[NB];
			bool on[NB];
			const int last = nloc * oct - 1;
#pragma unroll
			for (int u = 0; u < NB; ++u) { // straight-line loads (index clamped, rows are `stride` long): all in flight at once
				const int i = base + u * VTHREADS + tid;
				const int ic = min(i, last);
				sl[u] = ic / oct;
				q[u] = ic - sl[u] * oct;
				on[u] = i <= last;
				if (!from_fifo) {
					v[u] = *reinterpret_cast<const uint4 *>(a.samples + (size_t)(s0 + sl[u]) * a.stride + 8 * q[u]);
				} else {
					// the ring's capacity is a multiple of 8 (checked by the caller); while its head is one too -- every FIFO that is
					// only ever popped in chunks like this one -- a group is one aligned load that never wraps.  A head left
					// elsewhere by a pop of another size (mi_fifo_pop / _pop_frames allow any) takes the samples one by one.
					const int h = s_head[sl[u]];
					unsigned at = (unsigned)(h < 0 ? 0 : h) + 8u * (unsigned)q[u];
					if (at >= (unsigned)a.src.cap) at -= (unsigned)a.src.cap;
					const int16_t *ring = a.src.ring + (size_t)(s0 + sl[u]) * a.src.cap;
					if ((h & 7) == 0 || h < 0) {
						v[u] = *reinterpret_cast<const uint4 *>(ring + (at & ~7u));
					} else {
						unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
						for (int k = 0; k < 8; ++k) {
							unsigned p = at + (unsigned)k;
							if (p >= (unsigned)a.src.cap) p -= (unsigned)a.src.cap;
							w[k >> 1] |= (unsigned)(uint16_t)ring[p] << (16 * (k & 1));
						}
						v[u] = make_uint4(w[0], w[1], w[2], w[3]);
					}
					if (h < 0) v[u] = make_uint4(0, 0, 0, 0);
				}
			}
#pragma unroll
			for (int u = 0; u < NB; ++u) on[u] = on[u] && 8 * q[u] < s_n[sl[u]];
#pragma unroll
			for (int u = 0; u < NB; ++u) {
				if (!on[u]) continue;
				const int n = s_n[sl[u]];
				*reinterpret_cast<uint4 *>(rows + sl[u] * a.pitch_dw + 4 * q[u]) = v[u];
				const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
				float f[8];
				int pk = 0, dc = 0;
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const int x = (int)(short)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu));
					f[k] = (float)(x * x);
					if (8 * q[u] + k < n) {
						const int av = x < 0 ? -x : x;
						pk = max(pk, av);
						dc += x;
					}
				}
				float *d = sq + sl[u] * a.pitch_f + 8 * q[u];
				*reinterpret_cast<float4 *>(d) = make_float4(f[0], f[1], f[2], f[3]);
				*reinterpret_cast<float4 *>(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
				atomicMax(&s_pk[sl[u]], pk);
				atomicAdd(&s_dc[sl[u]], dc);
			}
		}
	} else {
		int16_t *rows16 = reinterpret_cast<int16_t *>(rows);
		for (int i = tid; i < nloc * a.nsamples; i += VTHREADS) {
			const int sl = i / a.nsamples, q = i - sl * a.nsamples;
			if (q < s_n[sl]) {
				const int x = a.samples[(size_t)(s0 + sl) * a.stride + q];
				rows16[sl * a.pitch_dw * 2 + q] = (int16_t)x;
				sq[sl * a.pitch_f + q] = (float)(x * x);
				atomicMax(&s_pk[sl], x < 0 ? -x : x);
				atomicAdd(&s_dc[sl], x);
			}
		}
	}
	__syncthreads();

	// ---- (B) one lane per stream: meter + control chain
	if (tid < nloc && s_n[tid] > 0) {
		const int s = s0 + tid;
		const int n = s_n[tid];
		const float *x2 = sq + tid * a.pitch_f;

		// same additions in the same order as update_energy's loop; 32 samples per stage, the next stage's
		// LDS reads in flight while this one is added
		float acc = 0;
		int i = 0;
		const float4 *p4 = reinterpret_cast<const float4 *>(x2);
		const int nb = n >> 5;
		if (nb > 0) {
			float4 c[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) c[k] = p4[k];
			for (int b = 0; b < nb; ++b) {
				float4 nx[8];
				const float4 *pn = p4 + 8 * min(b + 1, nb - 1);
#pragma unroll
				for (int k = 0; k < 8; ++k) nx[k] = pn[k];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					acc += c[k].x;
					acc += c[k].y;
					acc += c[k].z;
					acc += c[k].w;
				}
#pragma unroll
				for (int k = 0; k < 8; ++k) c[k] = nx[k];
			}
			i = nb << 5;
		}
		for (; i < n; ++i) acc += x2[i];
		float2 win = a.win[s];
		const VolCtl o = volume_control(p, st, peer_energy, acc, n, s_pk[tid], s_dc[tid], a.sample_rate, win);
		s_intgain[tid] = o.intgain, s_dcoff[tid] = o.dcoff, s_mode[tid] = o.mode;
		a.state[s] = st;
		a.energy[*a.parity ^ 1][s] = st.energy;
		a.win[s] = win;
	} else if (tid < SPB) {
		s_mode[tid] = 0;
		if (tid < nloc) a.energy[*a.parity ^ 1][s0 + tid] = a.state[s0 + tid].energy;
	}
	__syncthreads();

	// ---- (C) integer gain, coalesced write-back
	if (vec) {
		const int oct = (a.nsamples + 7) >> 3;
		for (int i = tid; i < nloc * oct; i += VTHREADS) {
			const int sl = i / oct, q = i - sl * oct;
			const int mode = s_mode[sl], n = s_n[sl];
			if (8 * q >= n) continue;
			const int16_t *r = reinterpret_cast<const int16_t *>(rows + sl * a.pitch_dw) + 8 * q;
			int16_t *dst = a.samples + (size_t)(s0 + sl) * a.stride + 8 * q;
			if (mode == 0) { // gain 1: in place nothing to do (msvolume.c:440); popped from a FIFO the chunk still has to land
				if (from_fifo) {
					if (8 * q + 8 <= n) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(r);
					else
						for (int k = 0; 8 * q + k < n; ++k) dst[k] = r[k];
				}
				continue;
			}
			const int ig = s_intgain[sl], dc = (mode == 2) ? s_dcoff[sl] : 0;
			if (8 * q + 8 <= n) {
				const uint4 rv = *reinterpret_cast<const uint4 *>(r);
				const unsigned w[4] = {rv.x, rv.y, rv.z, rv.w};
				unsigned o[4];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int lo = sat16((((int)(short)(w[k] & 0xffffu) - dc) * ig) / 4096);
					const int hi = sat16((((int)(short)(w[k] >> 16) - dc) * ig) / 4096);
					o[k] = (unsigned)(lo & 0xffff) | ((unsigned)hi << 16);
				}
				*reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
			} else {
				for (int k = 0; 8 * q + k < n; ++k) dst[k] = (int16_t)sat16(((r[k] - dc) * ig) / 4096);
			}
		}
	} else {
		const int16_t *rows16 = reinterpret_cast<const int16_t *>(rows);
		for (int i = tid; i < nloc * a.nsamples; i += VTHREADS) {
			const int sl = i / a.nsamples, q = i - sl * a.nsamples;
			const int mode = s_mode[sl];
			if (mode == 0 || q >= s_n[sl]) continue;
			const int dc = (mode == 2) ? s_dcoff[sl] : 0;
			a.samples[(size_t)(s0 + sl) * a.stride + q] =
			    (int16_t)sat16(((rows16[sl * a.pitch_dw * 2 + q] - dc) * s_intgain[sl]) / 4096);
		}
	}
}

__global__ void volume_flip_kernel(int *parity) { *parity ^= 1; }

// ---- MSVolume + MSAudioMixer of a conference in ONE kernel (the chain's last two filters: every leg's chunk is popped
// from its canceller's output FIFO, metered and levelled (volume_process, msvolume.c:471-514), and the conference is mixed
// from the levelled chunks (mixer_process in conference mode, audiomixer.c:288-346) without the levelled audio ever
// leaving the chip: a leg's tick crosses HBM twice (FIFO read, mix write) instead of four times.
// One workgroup = one conference; the members' chunks live in LDS as packed int16 rows (pitch: an odd number of 8-byte
// words, so the lanes that walk one row each in the serial meter read disjoint banks).
//   (0) lane m < members: parameters, state, peer energy, mixer controls, ms_bufferizer_read of its FIFO (all or nothing);
//   (A) eight lanes per member: the chunks into LDS, 16 bytes at a time; integer peak and DC sum in packed 16-bit
//       arithmetic, reduced over the eight lanes in registers;
//   (B) lane m: the float32 sum of squares IN SAMPLE ORDER (the order is part of the reference's result) and the control
//       chain (volume_control) -- one member per lane, all members at once;
//   (C + D) lane = four columns: the Q12 gain, then the mixer's input stage (channel_process_in: pin active? input gain)
//       in place and the int32 sum over the members; then for every pin with its output enabled saturate(sum - own)
//       (channel_process_out), 8 bytes per lane and row, rows contiguous.
struct VolMixArgs {
	VolArgs v;            // the volume batch (params, state, energy double buffer, window) and the source FIFO
	const uint8_t *flags; // mixer controls [nconf][mm]
	const float *gain;
	int16_t *out;         // [nconf][mm][ns]
	int mm, row_w;        // members per conference; row pitch in 8-byte words (odd)
	int dry_skips;        // MI_VOLMIX_DRY_SKIPS: a leg whose queue holds less than a tick is not metered at all (MSVolume gets no chunk)
	const uint8_t *run;   // nullable [nconf]: 0 = the conference does not tick in this launch (nothing popped, nothing written)
};
constexpr int VM_THREADS = 256, VM_MAXM = MI_MIXER_MAX_CHANNELS;
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(VM_THREADS) void volmix_kernel(VolMixArgs va) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const VolArgs &a = va.v;
	uint2 *rows = reinterpret_cast<uint2 *>(smem); // [mm][row_w] four samples per word
	__shared__ int s_pk[VM_MAXM], s_dc[VM_MAXM];
	__shared__ int4 s_par[VM_MAXM]; // what (C + D) needs of a member, one 16-byte broadcast read: (flags | mode << 8, Q12 gain, DC offset, pin gain)
	const int t = threadIdx.x, c = blockIdx.x, mm = va.mm, ns = a.nsamples, nw = ns >> 2, ng = ns >> 3;
	if (va.run && !va.run[c]) return;
	const int s0 = a.first + c * mm;

	mi_volume_params p;
	mi_volume_state st;
	float peer_energy = 0;
	unsigned mflag = 0; // the member's mixer controls
	int mgain_bits = 0;
	int2 qpos = make_int2(0, 0); // its queue (head, level) before this tick
	float2 win = make_float2(0, 0);
	if (t < mm) {
		const int s = s0 + t;
		p = a.params[s];
		st = a.state[s];
		if (p.peer >= 0) peer_energy = a.energy[*a.parity][p.peer];
		else if (p.peer == MI_VOLUME_PEER_EXTERNAL && a.ext_state) peer_energy = a.ext_state[s].energy;
		else if (p.peer == MI_VOLUME_PEER_EXTERNAL) p.peer = -1;
		mflag = va.flags[c * mm + t];
		mgain_bits = __float_as_int(va.gain[c * mm + t]);
		win = a.win[s];
		qpos = a.src.pos[s]; // ms_bufferizer_read, all-or-nothing (msqueue.c:83); a leg that runs dry hears and meters silence
	}

	// ---- (A) eight lanes per member (lane q takes the 16-byte groups q, q + 8, ..: 128 contiguous bytes per member and
	// round), every lane reading its member's queue position itself (the pop is written back in (B), behind the barrier)
	// and asking for all of its groups at once -- up to eight loads in flight per lane instead of one round trip per
	// group --; peak and DC sum in packed 16-bit arithmetic -- the peak as max(max x, -min x), exact for -32768 too --,
	// reduced over the eight lanes in registers (DPP): one plain LDS word per member and quantity, no LDS atomics
	// (two per group, most of a wave on ONE address, were a fifth of the kernel)
	for (int mb = 0; mb < mm; mb += VM_THREADS / 8) {
		const int m = mb + (t >> 3), q = t & 7;
		const bool valid = m < mm;
		s16x2 pmax = {0, 0}, pmin = {0, 0};
		int dc = 0;
		auto take = [&](int g, uint4 v) { // group g of member m: into its row, into the statistics
			rows[m * va.row_w + 2 * g] = make_uint2(v.x, v.y);
			rows[m * va.row_w + 2 * g + 1] = make_uint2(v.z, v.w);
			const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const s16x2 x = __builtin_bit_cast(s16x2, w[k]);
				pmax = __builtin_elementwise_max(pmax, x);
				pmin = __builtin_elementwise_min(pmin, x);
				dc = __builtin_amdgcn_sdot2(x, (s16x2){1, 1}, dc, false);
			}
		};
		if (valid) {
			const int2 qp = a.src.pos[s0 + m];
			const int h = qp.y >= ns ? qp.x : -1;
			const int16_t *ring = a.src.ring + (size_t)(s0 + m) * a.src.cap;
			if (h < 0 || (h & 7) == 0) {
				for (int g = q; g < ng; g += 64) { // eight groups per lane and round: one round at 480 samples
					uint4 v[8];
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						v[i] = make_uint4(0, 0, 0, 0);
						if (h >= 0 && g + 8 * i < ng) {
							unsigned at = (unsigned)h + 8u * (unsigned)(g + 8 * i);
							if (at >= (unsigned)a.src.cap) at -= (unsigned)a.src.cap;
							v[i] = *reinterpret_cast<const uint4 *>(ring + at);
						}
					}
#pragma unroll
					for (int i = 0; i < 8; ++i)
						if (g + 8 * i < ng) take(g + 8 * i, v[i]);
				}
			} else { // a head some other reader left off the 16-byte grid: sample by sample, wrap-aware
				for (int g = q; g < ng; g += 8) {
					unsigned at = (unsigned)h + 8u * (unsigned)g;
					if (at >= (unsigned)a.src.cap) at -= (unsigned)a.src.cap;
					unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						unsigned qq = at + (unsigned)k;
						if (qq >= (unsigned)a.src.cap) qq -= (unsigned)a.src.cap;
						w[k >> 1] |= (unsigned)(uint16_t)ring[qq] << (16 * (k & 1));
					}
					take(g, make_uint4(w[0], w[1], w[2], w[3]));
				}
			}
		}
		unsigned umax = __builtin_bit_cast(unsigned, pmax), umin = __builtin_bit_cast(unsigned, pmin);
#define VM_DPP(x, ctrl) __builtin_amdgcn_update_dpp((int)(x), (int)(x), ctrl, 0xf, 0xf, false)
#define VM_STEP(ctrl)                                                                                                                            \
	umax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, umax), __builtin_bit_cast(s16x2, (unsigned)VM_DPP(umax, ctrl)))); \
	umin = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s16x2, umin), __builtin_bit_cast(s16x2, (unsigned)VM_DPP(umin, ctrl)))); \
	dc += VM_DPP(dc, ctrl);
		VM_STEP(0xB1)  // quad_perm [1,0,3,2]
		VM_STEP(0x4E)  // quad_perm [2,3,0,1]
		VM_STEP(0x141) // row_half_mirror: the other quad of the eight
#undef VM_STEP
#undef VM_DPP
		if (valid && q == 0) {
			const s16x2 hi = __builtin_bit_cast(s16x2, umax), lo = __builtin_bit_cast(s16x2, umin);
			s_pk[m] = max(max((int)hi.x, (int)hi.y), -min((int)lo.x, (int)lo.y));
			s_dc[m] = dc;
		}
	}
	__syncthreads();

	// ---- (B)
	if (t < mm) {
		const uint2 *r = rows + t * va.row_w;
		float acc = 0;
		// same additions in the same order as update_energy's loop; a sample's square as the product of two floats -- the
		// exact integer below 2^31 rounded once, what the conversion of the integer product gives -- eight samples a round
		auto sq = [&](unsigned w) {
			const float lo = (float)(int)(short)(w & 0xffffu), hi = (float)(int)(short)(w >> 16);
			acc += lo * lo;
			acc += hi * hi;
		};
		// The next round's four words are asked for BEFORE this round's 16 dependent operations and waited for behind them: by
		// hand, because the compiler sinks the read to its first use and then waits at once -- an LDS round trip per round, a
		// third of this phase, and this phase is the longest stretch of a conference's workgroup (one wave busy, three waiting).
		typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
		const unsigned r_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)r;
		u32x4 cur, nxt;
		asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cur) : "v"(r_lds));
		for (int i = 0; i < nw; i += 2) {
			const int in = min(i + 2, nw - 2);
			asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=&v"(nxt) : "v"(r_lds + 8u * (unsigned)in));
			__builtin_amdgcn_sched_barrier(0);
			sq(cur.x), sq(cur.y), sq(cur.z), sq(cur.w);
			__builtin_amdgcn_sched_barrier(0);
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt));
			cur = nxt;
		}
		const int s = s0 + t;
		const bool dry = qpos.y < ns;
		if (!dry) a.src.pos[s] = make_int2((qpos.x + ns) % a.src.cap, qpos.y - ns); // (every lane of (A) has read the old position: the barrier)
		if (va.dry_skips && dry) {
			// the plugin's chain: volume_process (msvolume.c:480-486) finds no whole 10 ms chunk in its bufferizer and does
			// nothing -- no meter update, no gain ramp --, the mixer reads zeros for the pin (audiomixer.c:88)
			s_par[t] = make_int4((int)mflag, 4096, 0, mgain_bits);
			a.energy[*a.parity ^ 1][s] = st.energy;
		} else {
			const VolCtl o = volume_control(p, st, peer_energy, acc, ns, s_pk[t], s_dc[t], a.sample_rate, win);
			s_par[t] = make_int4((int)mflag | (o.mode << 8), o.intgain, o.dcoff, mgain_bits);
			a.state[s] = st;
			a.energy[*a.parity ^ 1][s] = st.energy;
			a.win[s] = win;
		}
	}
	__syncthreads();

	// ---- (C + D) lane = four columns.  First loop over the members: the Q12 gain (apply_gain), then the pin's contribution as
	// channel_process_in leaves it (0 unless linked and active) -- back into the row, in place -- and the int32 sum over
	// the members; second loop: for every pin with its output enabled saturate(sum - own) (channel_process_out), 8 bytes
	// per lane and row, rows contiguous.  A lane only re-reads what it wrote itself: no barrier in between, no index
	// arithmetic per element (the two phases as loops over (member, word) items divided by a run-time width per item).
	// a member's record as scalars: lane m of every wave holds member m's (mm <= 50 < 64), v_readlane hands it to the wave --
	// no LDS round trip per member in front of the row's, uniform branches, scalar operands
	const int wl = t & 63;
	const int4 mine = wl < mm ? s_par[wl] : make_int4(0, 0, 0, 0);
	typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
	const unsigned rows_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)rows;
	const unsigned pitch = 8u * (unsigned)va.row_w;
	for (int j = t; j < nw; j += VM_THREADS) {
		// (the next member's word is asked for before this member's arithmetic and waited for behind it, by hand as in (B))
		const unsigned col = rows_lds + 8u * (unsigned)j;
		int sum[4] = {0, 0, 0, 0};
		u32x2 cur, nxt;
		asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cur) : "v"(col));
		nxt = cur;
		for (int m = 0; m < mm; ++m) {
			if (m + 1 < mm) asm volatile("ds_read_b64 %0, %1" : "=&v"(nxt) : "v"(col + (unsigned)(m + 1) * pitch));
			const int px = __builtin_amdgcn_readlane(mine.x, m);
			const unsigned f = (unsigned)px & 0xffu;
			uint2 o = make_uint2(0, 0);
			if ((f & MI_MIX_LINKED) && (f & MI_MIX_ACTIVE)) {
				int x[4] = {(int)(short)(cur.x & 0xffffu), (int)(short)(cur.x >> 16), (int)(short)(cur.y & 0xffffu), (int)(short)(cur.y >> 16)};
				const int mode = px >> 8;
				if (mode != 0) {
					const int ig = __builtin_amdgcn_readlane(mine.y, m), dc = (mode == 2) ? __builtin_amdgcn_readlane(mine.z, m) : 0;
#pragma unroll
					for (int k = 0; k < 4; ++k) x[k] = sat16(((x[k] - dc) * ig) / 4096);
				}
				const float gn = __int_as_float(__builtin_amdgcn_readlane(mine.w, m));
				if (gn != 1.0f) {
#pragma unroll
					for (int k = 0; k < 4; ++k) x[k] = sat16((int)(gn * (float)x[k]));
				}
#pragma unroll
				for (int k = 0; k < 4; ++k) sum[k] += x[k];
				o.x = (unsigned)(x[0] & 0xffff) | ((unsigned)x[1] << 16);
				o.y = (unsigned)(x[2] & 0xffff) | ((unsigned)x[3] << 16);
			}
			rows[m * va.row_w + j] = o;
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt));
			cur = nxt;
		}
		asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cur) : "v"(col) : "memory");
		nxt = cur;
		for (int m = 0; m < mm; ++m) {
			if (m + 1 < mm) asm volatile("ds_read_b64 %0, %1" : "=&v"(nxt) : "v"(col + (unsigned)(m + 1) * pitch));
			if ((unsigned)__builtin_amdgcn_readlane(mine.x, m) & MI_MIX_OUTPUT) {
				const int o0 = sat16(sum[0] - (int)(short)(cur.x & 0xffffu)), o1 = sat16(sum[1] - (int)(short)(cur.x >> 16));
				const int o2 = sat16(sum[2] - (int)(short)(cur.y & 0xffffu)), o3 = sat16(sum[3] - (int)(short)(cur.y >> 16));
				*reinterpret_cast<uint2 *>(va.out + ((size_t)(c * mm + m) * ns) + 4 * j) =
				    make_uint2((unsigned)(o0 & 0xffff) | ((unsigned)o1 << 16), (unsigned)(o2 & 0xffff) | ((unsigned)o3 << 16));
			}
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt));
			cur = nxt;
		}
	}
}

} // namespace

struct mi_volume {
	mi_ctx *ctx = nullptr;
	int nstreams = 0, sample_rate = 0;
	mi_volume_params *d_params = nullptr;
	mi_volume_state *d_state = nullptr;
	float2 *d_win = nullptr;
	float *d_energy[2] = {nullptr, nullptr};
	int *d_parity = nullptr;
	bool has_peers = false; // conservative: set once any stream names a peer
	const mi_volume *ext = nullptr; // mi_volume_set_peer_batch
	std::vector<mi_volume *> ext_users; // the batches whose `ext` is this one: told when it goes (mi_volume_destroy)
};

extern "C" {

void mi_volume_default_params(mi_volume_params *p) {
	if (!p) return;
	memset(p, 0, sizeof(*p));
	p->static_gain = 1;
	p->vol_upramp = 0.4f;
	p->vol_fast_upramp = 0.4f * 3;
	p->vol_downramp = 0.4f;
	p->ea_thres = 0.1f;
	p->ea_transmit_thres = 4;
	p->force = 4.0f;
	p->sustain_time = 200;
	p->ng_cut_time = 400;
	p->ng_threshold = 0.1f;
	p->ng_floorgain = 0.005f;
	p->peer = -1;
}

int mi_volume_create(mi_ctx *ctx, int nstreams, int sample_rate, mi_volume **out) {
	MI_CHECK_ARG(ctx && out && nstreams > 0 && sample_rate > 0);
	*out = nullptr;
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	mi_volume *v = new mi_volume();
	v->ctx = ctx;
	v->nstreams = nstreams;
	v->sample_rate = sample_rate;
	if (hipMalloc((void **)&v->d_params, sizeof(mi_volume_params) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_state, sizeof(mi_volume_state) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_energy[0], sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_energy[1], sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMalloc((void **)&v->d_parity, sizeof(int)) != hipSuccess ||
	    hipMalloc((void **)&v->d_win, sizeof(float2) * (size_t)nstreams) != hipSuccess) {
		mi::set_error("hipMalloc failed for volume state");
		mi_volume_destroy(v);
		return MI_ENOMEM;
	}
	mi_volume_params dp;
	mi_volume_default_params(&dp);
	std::vector<mi_volume_params> hp((size_t)nstreams, dp);
	mi_volume_state ds;
	memset(&ds, 0, sizeof(ds));
	ds.gain = ds.target_gain = 1; // volume_init msvolume.c:92
	ds.ng_gain = 1;               // :112
	std::vector<mi_volume_state> hs((size_t)nstreams, ds);
	if (hipMemcpy(v->d_params, hp.data(), sizeof(dp) * hp.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(v->d_state, hs.data(), sizeof(ds) * hs.size(), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemset(v->d_energy[0], 0, sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMemset(v->d_energy[1], 0, sizeof(float) * (size_t)nstreams) != hipSuccess ||
	    hipMemset(v->d_parity, 0, sizeof(int)) != hipSuccess || mi_volume_reset_max(v, 0, nstreams) != MI_OK) {
		mi::set_error("volume state upload failed");
		mi_volume_destroy(v);
		return MI_ENODEV;
	}
	*out = v;
	return MI_OK;
}

void mi_volume_destroy(mi_volume *v) {
	if (!v) return;
	// (no launch may read a destroyed batch's state: whoever named this batch as its peers' loses the link -- its EXTERNAL streams then have
	// no peer -- and this batch leaves the list of the one it named.  Calls on batches of one context come from one thread at a time, as for
	// every object of the library)
	for (mi_volume *u : v->ext_users) u->ext = nullptr;
	if (v->ext) {
		auto &l = const_cast<mi_volume *>(v->ext)->ext_users;
		l.erase(std::remove(l.begin(), l.end(), v), l.end());
	}
	(void)hipSetDevice(v->ctx->device);
	if (v->d_params) (void)hipFree(v->d_params);
	if (v->d_state) (void)hipFree(v->d_state);
	if (v->d_energy[0]) (void)hipFree(v->d_energy[0]);
	if (v->d_energy[1]) (void)hipFree(v->d_energy[1]);
	if (v->d_parity) (void)hipFree(v->d_parity);
	if (v->d_win) (void)hipFree(v->d_win);
	delete v;
}

int mi_volume_set_params(mi_volume *v, int first, int count, const mi_volume_params *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	for (int i = 0; i < count; ++i) MI_CHECK_ARG(h[i].peer < v->nstreams && h[i].peer >= MI_VOLUME_PEER_EXTERNAL);
	for (int i = 0; i < count; ++i)
		if (h[i].peer >= 0) v->has_peers = true;
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(v->d_params + first, h, sizeof(*h) * (size_t)count, hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_volume_set_peer_batch(mi_volume *v, mi_volume *peers) {
	MI_CHECK_ARG(v && (!peers || (peers->nstreams >= v->nstreams && peers->ctx == v->ctx && peers != v)));
	if (v->ext) {
		auto &l = const_cast<mi_volume *>(v->ext)->ext_users;
		l.erase(std::remove(l.begin(), l.end(), v), l.end());
	}
	v->ext = peers;
	if (peers) peers->ext_users.push_back(v);
	return MI_OK;
}

int mi_volume_get_state(mi_volume *v, int first, int count, mi_volume_state *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(h, v->d_state + first, sizeof(*h) * (size_t)count, hipMemcpyDeviceToHost));
	return MI_OK;
}

int mi_volume_get_state_async(mi_volume *v, int first, int count, mi_volume_state *h_pinned) {
	MI_CHECK_ARG(v && h_pinned && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	return mi_copy_d2h_pinned(v->ctx, h_pinned, v->d_state + first, sizeof(*h_pinned) * (size_t)count);
}

int mi_volume_set_state(mi_volume *v, int first, int count, const mi_volume_state *h) {
	MI_CHECK_ARG(v && h && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	MI_HIP(hipMemcpy(v->d_state + first, h, sizeof(*h) * (size_t)count, hipMemcpyHostToDevice));
	std::vector<float> e((size_t)count);
	for (int i = 0; i < count; ++i) e[(size_t)i] = h[i].energy;
	// both halves: whichever the device-side parity says is "previous" must show these energies to the peers
	MI_HIP(hipMemcpy(v->d_energy[0] + first, e.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice));
	MI_HIP(hipMemcpy(v->d_energy[1] + first, e.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice));
	return MI_OK;
}

int mi_volume_get_max(mi_volume *v, int first, int count, float *h_max) {
	MI_CHECK_ARG(v && h_max && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	std::vector<float2> w((size_t)count);
	MI_HIP(hipMemcpy(w.data(), v->d_win + first, sizeof(float2) * (size_t)count, hipMemcpyDeviceToHost));
	for (int i = 0; i < count; ++i) h_max[i] = w[(size_t)i].y < 0 ? 0.f : w[(size_t)i].x; // ortp_extremum_init: 0 before the first record
	return MI_OK;
}

int mi_volume_reset_max(mi_volume *v, int first, int count) {
	MI_CHECK_ARG(v && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_HIP(hipStreamSynchronize(v->ctx->stream));
	std::vector<float2> w((size_t)count, make_float2(0.f, -1.f));
	MI_HIP(hipMemcpy(v->d_win + first, w.data(), sizeof(float2) * (size_t)count, hipMemcpyHostToDevice));
	return MI_OK;
}

static int volume_launch(mi_volume *v, int16_t *d_samples, int nsamples, int stride, const int32_t *d_nsamples, const mi_fifo *src,
                         int first = 0, int count = -1, unsigned flags = 0);

int mi_volume_process(mi_volume *v, int16_t *d_samples, int nsamples, int stride, const int32_t *d_nsamples) {
	return volume_launch(v, d_samples, nsamples, stride, d_nsamples, nullptr);
}

int mi_volume_process_fifo(mi_volume *v, mi_fifo *f_src, int16_t *d_out, int nsamples, int stride) {
	MI_CHECK_ARG(v && f_src && f_src->nstreams == v->nstreams);
	if ((f_src->capacity & 7) || (nsamples & 7) || (stride & 7) || (reinterpret_cast<uintptr_t>(d_out) & 15)) {
		mi::set_error("mi_volume_process_fifo: capacity, chunk and stride must be multiples of 8 samples, rows 16-byte aligned");
		return MI_ENOTSUP;
	}
	return volume_launch(v, d_out, nsamples, stride, nullptr, f_src);
}

int mi_volume_process_fifo_flags(mi_volume *v, mi_fifo *f_src, int16_t *d_out, int nsamples, int stride, unsigned flags) {
	MI_CHECK_ARG(v && f_src && f_src->nstreams == v->nstreams);
	if ((f_src->capacity & 7) || (nsamples & 7) || (stride & 7) || (reinterpret_cast<uintptr_t>(d_out) & 15)) {
		mi::set_error("mi_volume_process_fifo_flags: capacity, chunk and stride must be multiples of 8 samples, rows 16-byte aligned");
		return MI_ENOTSUP;
	}
	return volume_launch(v, d_out, nsamples, stride, nullptr, f_src, 0, -1, flags);
}

int mi_volume_process_fifo_range(mi_volume *v, mi_fifo *f_src, int16_t *d_out, int nsamples, int stride, int first, int count) {
	MI_CHECK_ARG(v && f_src && f_src->nstreams == v->nstreams && first >= 0 && count >= 0 && first + count <= v->nstreams);
	if ((f_src->capacity & 7) || (nsamples & 7) || (stride & 7) || (reinterpret_cast<uintptr_t>(d_out) & 15)) {
		mi::set_error("mi_volume_process_fifo_range: capacity, chunk and stride must be multiples of 8 samples, rows 16-byte aligned");
		return MI_ENOTSUP;
	}
	if (count == 0) return MI_OK;
	if (v->has_peers && count != v->nstreams) {
		mi::set_error("mi_volume_process_fifo_range: a batch with echo-limiter peers is processed whole (the peers read each other's "
		              "energy of the previous launch, msvolume.c:206-207)");
		return MI_ENOTSUP;
	}
	return volume_launch(v, d_out, nsamples, stride, nullptr, f_src, first, count);
}

int mi_mixer_process_volume_fifo(mi_mixer *m, mi_volume *v, int first_stream, mi_fifo *f_src, int16_t *d_out) {
	return mi_mixer_process_volume_fifo_flags(m, v, first_stream, f_src, d_out, 0u, nullptr);
}

int mi_mixer_process_volume_fifo_flags(mi_mixer *m, mi_volume *v, int first_stream, mi_fifo *f_src, int16_t *d_out, unsigned flags,
                                       const uint8_t *d_run) {
	MI_CHECK_ARG(m && v && f_src && d_out && first_stream >= 0);
	MixerView mv;
	mi_mixer_view(m, &mv);
	MI_CHECK_ARG(f_src->nstreams == v->nstreams && first_stream + mv.nconf * mv.mm <= v->nstreams && mv.device == v->ctx->device);
	const int row_w = (mv.ns >> 2) | 1; // 8-byte words per row, odd
	const size_t lds = (size_t)mv.mm * row_w * 8;
	if ((mv.ns & 7) || (f_src->capacity & 7) || (reinterpret_cast<uintptr_t>(d_out) & 7) || lds > 120 * 1024) {
		mi::set_error("mi_mixer_process_volume_fifo: ticks and FIFO capacities must be multiples of 8 samples and a conference's tick must "
		              "fit the LDS (%d members x %d samples)", mv.mm, mv.ns);
		return MI_ENOTSUP;
	}
	if (v->has_peers && mv.nconf * mv.mm != v->nstreams) {
		mi::set_error("mi_mixer_process_volume_fifo: a volume batch with echo-limiter peers must be covered by the conferences entirely");
		return MI_ENOTSUP;
	}
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	VolMixArgs a;
	a.v.samples = nullptr;
	a.v.nsamples_per_stream = nullptr;
	a.v.params = v->d_params;
	a.v.state = v->d_state;
	a.v.win = v->d_win;
	a.v.ext_state = v->ext ? v->ext->d_state : nullptr;
	a.v.energy[0] = v->d_energy[0];
	a.v.energy[1] = v->d_energy[1];
	a.v.parity = v->d_parity;
	a.v.nstreams = v->nstreams;
	a.v.nsamples = mv.ns;
	a.v.stride = mv.ns;
	a.v.sample_rate = v->sample_rate;
	a.v.pitch_dw = a.v.pitch_f = 0;
	a.v.first = first_stream;
	a.v.src = fifo_view(f_src);
	a.v.src_dry_skips = 0;
	a.flags = mv.flags;
	a.gain = mv.gain;
	a.out = d_out;
	a.mm = mv.mm;
	a.row_w = row_w;
	a.dry_skips = (flags & MI_VOLMIX_DRY_SKIPS) ? 1 : 0;
	a.run = d_run;
	static std::atomic<uint64_t> big_lds{0}; // more than 64 KB of dynamic LDS needs the attribute once per device
	const uint64_t dev_bit = 1ull << (v->ctx->device & 63);
	if (lds > 64 * 1024 && !(big_lds.load(std::memory_order_relaxed) & dev_bit)) {
		MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(volmix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
		big_lds.fetch_or(dev_bit, std::memory_order_relaxed);
	}
	hipLaunchKernelGGL(volmix_kernel, dim3(mv.nconf), dim3(VM_THREADS), lds, v->ctx->stream, a);
	MI_LAUNCH_CHECK();
	if (v->has_peers) {
		hipLaunchKernelGGL(volume_flip_kernel, dim3(1), dim3(1), 0, v->ctx->stream, v->d_parity);
		MI_LAUNCH_CHECK();
	}
	return MI_OK;
}

static int volume_launch(mi_volume *v, int16_t *d_samples, int nsamples, int stride, const int32_t *d_nsamples, const mi_fifo *src,
                         int first, int count, unsigned flags) {
	MI_CHECK_ARG(v && d_samples && nsamples > 0 && stride >= nsamples);
	if (nsamples > 3840) {
		mi::set_error("chunk of %d samples exceeds the volume kernel's LDS staging (max 3840)", nsamples);
		return MI_ENOTSUP;
	}
	if (v->ctx->activate() != MI_OK) return MI_ENODEV;
	VolArgs a;
	a.samples = d_samples;
	a.nsamples_per_stream = d_nsamples;
	a.params = v->d_params;
	a.state = v->d_state;
	a.win = v->d_win;
	a.ext_state = v->ext ? v->ext->d_state : nullptr;
	a.energy[0] = v->d_energy[0];
	a.energy[1] = v->d_energy[1];
	a.parity = v->d_parity;
	if (count < 0) count = v->nstreams - first;
	a.first = first;
	a.nstreams = first + count; // the kernel's upper bound
	a.nsamples = nsamples;
	a.stride = stride;
	a.sample_rate = v->sample_rate;
	a.src = fifo_view(src);
	a.src_dry_skips = (flags & MI_VOLMIX_DRY_SKIPS) ? 1 : 0;
	// packed rows: whole 16-byte groups.  Float rows: whole groups of 8, and an odd number of 16-byte
	// groups per row so the SPB lanes of phase B read disjoint banks.
	const int pitch = ((nsamples + 7) >> 3) * 4;
	int pitch_f = ((nsamples + 7) >> 3) * 8;
	if (((pitch_f >> 2) & 1) == 0) pitch_f += 4;
	a.pitch_dw = pitch;
	a.pitch_f = pitch_f;
	const size_t lds = (size_t)SPB * pitch * sizeof(uint32_t) + (size_t)SPB * pitch_f * sizeof(float);
	if (lds > 60 * 1024) {
		mi::set_error("volume chunk of %d samples does not fit the LDS staging (max ~2500)", nsamples);
		return MI_ENOTSUP;
	}
	hipLaunchKernelGGL(volume_kernel, dim3(mi::ceil_div(count, SPB)), dim3(VTHREADS), lds, v->ctx->stream, a);
	MI_LAUNCH_CHECK();
	if (v->has_peers) { // without peers nobody reads the previous energies: no flip, no extra launch
		hipLaunchKernelGGL(volume_flip_kernel, dim3(1), dim3(1), 0, v->ctx->stream, v->d_parity);
		MI_LAUNCH_CHECK();
	}
	return MI_OK;
}

int mi_volume_process_host(mi_volume *v, int16_t *h_samples, int nsamples, int stride, const int32_t *h_nsamples) {
	MI_CHECK_ARG(v && h_samples);
	mi_ctx *c = v->ctx;
	if (c->activate() != MI_OK) return MI_ENODEV;
	const size_t b = (size_t)v->nstreams * stride * sizeof(int16_t);
	void *d, *dn = nullptr;
	int rc;
	if ((rc = c->ensure_scratch(0, b, &d)) != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(d, h_samples, b, hipMemcpyHostToDevice, c->stream));
	if (h_nsamples) {
		if ((rc = c->ensure_scratch(2, sizeof(int32_t) * (size_t)v->nstreams, &dn)) != MI_OK) return rc;
		MI_HIP(hipMemcpyAsync(dn, h_nsamples, sizeof(int32_t) * (size_t)v->nstreams, hipMemcpyHostToDevice, c->stream));
	}
	rc = mi_volume_process(v, (int16_t *)d, nsamples, stride, (const int32_t *)dn);
	if (rc != MI_OK) return rc;
	MI_HIP(hipMemcpyAsync(h_samples, d, b, hipMemcpyDeviceToHost, c->stream));
	MI_HIP(hipStreamSynchronize(c->stream));
	return MI_OK;
}

} // extern "C"

// (mi_warmup, ctx.hip: this unit's code object is loaded when the library is, not under a tick's first launch)
static const mi::WarmEntry g_warm_volume(reinterpret_cast<const void *>(&volume_flip_kernel));
