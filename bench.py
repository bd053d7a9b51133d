#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched DSP backend.

Contract (see DESIGN.md "Measurement"):
  python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0.

Workload: the north_star hot path, chained on the device, per call leg and
10 ms tick (src/base/msticker.c:46): MSResample 16k->48k -> MSSpeexEC (48 kHz,
256-sample frames, 128 ms tail, canceller + post-filter, its bufferizers
folded into the kernel) -> MSVolume (AGC) -> MSAudioMixer (conferences of 32).
A "step" is one tick of every leg on the GPU; inputs are resident in HBM.
Input: SURVEY 8(d)'s echo scene -- the microphone is the far end through a room
(0.5 x, 20 ms, 64 taps) plus near-end noise -- and the cancellers are in STEADY
STATE: converged on their scenes before anything is timed (class Converged).

value = concurrent 48 kHz legs the job sustains: the largest leg count per GPU
(capacity sweep) at which no tick of --worst-ticks (3000) CONSECUTIVE single
ticks reaches the 10 ms interval (p50 / p99 / p99.9 / max in the line; no tick
is discarded; a series is run again, once, only when EVERY late tick in it was a
stall of the submitting host thread, measured beside each tick: TickTimes,
host_stalls_only -- both series stay in the line), none of --paced-ticks (3000) ticks fired one per
10 ms of wall time does (the deployed cadence: config.paced_ticks), and none
does either when every leg starts from reset at once; summed over the ranks.  ms_per_step = the average tick at that
count over the timed region (whole 16-tick scene periods, at least 0.5 s,
replayed from a hipGraph so the host's launch cost is not what is timed).
Every tick streams the cancellers' resident state (~180 KB per leg, gigabytes
per GPU), so nothing of the working set survives in the 256 MiB Infinity Cache
between ticks.  The legs' re-framing phases are the product's own
(mi_aec_stagger_fifos); config.legs_in_phase shows the same count without.

N > 1 (one rank per GPU, torch.distributed.run): legs and whole conferences are
sharded statically (no collective); in addition 64 conferences are split over
ALL ranks and mixed through the path's one exchange step every tick:
mi_mixer_partial_sum -> mi_exchange_allreduce_i32 (C ABI, straight on RCCL, on
the kernel stream: torch.distributed only launches the ranks and carries the
128-byte id) -> mi_mixer_finalize, checked bit for bit against the single-GPU
mix on rank 0.  RCCL failure = non-zero exit.

roofline = the canceller's tick kernel (canceller + post-filter + FIFOs, one
launch) INSIDE the running chain at the headline leg count, HIP events on the
launch stream around that launch; cpu_baseline = the oracle's same chain on the
host's cores (bounded sample).  other_kernels: BASELINE configs[1]-[4], the
small-frame cancellers and the adjacent stages, each with its own roofline object.
plugin_path: full call legs through the DROP-IN PLUGIN (tests/host/plugin_bench:
filters by id from the factory, 16 ticker threads paced at 10 ms; never part of
`value`); video_pcie_inclusive: config 5 from and to host memory.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
TICKS_PER_S = 100.0    # MSTicker interval 10 ms (src/base/msticker.c:46)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--streams", type=int, default=0, help="call legs per GPU; 0 = capacity sweep (largest count whose worst tick < 10 ms)")
    ap.add_argument("--sweep-lo", type=int, default=98304)
    ap.add_argument("--sweep-hi", type=int, default=163840)
    ap.add_argument("--min-timed-s", type=float, default=0.0, help="(off by default: EXACTLY --steps steps are timed) make the timed region at least this long; the line then carries steps_requested beside steps")
    ap.add_argument("--worst-ticks", type=int, default=3000, help="consecutive single ticks the worst tick is taken over")
    ap.add_argument("--zero-ticks", type=int, default=256, help="ticks of the from-reset test at the chosen count (0 = skip)")
    ap.add_argument("--paced-ticks", type=int, default=3000, help="ticks of the series run at the 10 ms cadence of an MSTicker, one per 10 ms of wall time: part of `value`'s criterion (0 = skip)")
    ap.add_argument("--accept-seconds", type=float, default=150.0, help="wall time the step-downs of the acceptance series may take before the next count is chosen with room for the largest machine event seen on this hardware (1.8 ms)")
    ap.add_argument("--no-plugin-path", action="store_true", help="skip the rate of full call legs through the drop-in plugin (tests/host/plugin_bench)")
    ap.add_argument("--plugin-legs", type=int, default=32768, help="full call legs the plugin path is first tried with (config[3]'s count: 1024 conferences x 32)")
    ap.add_argument("--no-video-host", action="store_true", help="skip the PCIe-inclusive video probe (config 5)")
    ap.add_argument("--roofline-ticks", type=int, default=32, help="eager ticks with HIP events around the canceller's launch")
    ap.add_argument("--from-reset", action="store_true", help="measure cancellers that start from reset instead of steady state")
    ap.add_argument("--no-session", action="store_true", help="skip the PCIe-inclusive mi_session probes")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel roofline table")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (everything the short stdout line leaves out)")
    return ap.parse_args()


def synth_pcm_batch(nstreams, n, rate, seed0=0x5EED, sigma=3000.0):
    """N(0, 3000) + -20 dBFS 1 kHz tone, one RNG per 64-stream group (cheap, still distinct)."""
    out = np.empty((nstreams, n), np.int16)
    t = np.arange(n) / rate
    tone = 3276.7 * np.sin(2 * np.pi * 1000.0 * t)
    for g in range(0, nstreams, 64):
        rng = np.random.default_rng(seed0 + g)
        k = min(64, nstreams - g)
        out[g:g + k] = np.clip(np.round(rng.normal(0.0, sigma, (k, n)) + tone), -32767, 32767).astype(np.int16)
    return out


class Leg:
    """One kernel measured as K graph-replayed launches over a ring of tick buffers."""

    def __init__(self, ctx, name, launch, ring, alg_bytes, units, unit_name):
        self.ctx, self.name, self.launch, self.ring = ctx, name, launch, ring
        self.alg_bytes, self.units, self.unit_name = alg_bytes, units, unit_name

    def run(self, steps, warmup, use_graph=True):
        import torch
        torch.cuda.synchronize()  # buffers were filled on torch's stream; launches go to the context's stream
        for i in range(warmup):
            self.launch(i % self.ring)
        self.ctx.sync()
        graph = None
        if use_graph:
            self.ctx.capture_begin()
            for i in range(steps):
                self.launch(i % self.ring)
            graph = self.ctx.capture_end()
            graph.launch()  # untimed: uploads the graph, K more warm steps
            self.ctx.sync()
        return graph

    def timed(self, steps, graph):
        self.ctx.timer_start()
        if graph is not None:
            graph.launch()
        else:
            for i in range(steps):
                self.launch(i % self.ring)
        return self.ctx.timer_stop()  # ms over the K launches, HIP events on the launch stream


def roofline(ev_ms, steps, alg_bytes, traffic=None):
    dur_s = ev_ms * 1e-3 / steps
    gbs = alg_bytes / dur_s / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_us": round(dur_s * 1e6, 3), "algorithmic_bytes_per_launch": int(alg_bytes)}


def pmc_traffic(kernel, streams=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if there is one ('a+b' = both kernels of a leg).
    With `streams`: only an entry collected AT that leg count with the current kernel counts (at_streams), never the older
    top-level figure."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        d = json.load(open(p))
        tot = 0
        for k in kernel.split("+"):
            e = d[k.split("<")[0]]
            tot += (e["at_streams"][str(streams)] if streams else e)["hbm_bytes_per_launch"]
        return tot
    except Exception:
        return None


def pmc_traffic_at(kernel, streams):
    """(HBM bytes per launch, provenance) of the canceller at `streams` legs from the committed counter passes: the entry
    collected AT that leg count if there is one (scripts/r03_profile.sh), else the nearest one scaled by the leg count."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        d = json.load(open(p))[kernel]
        at = d.get("at_streams") or {}
        if str(streams) in at:
            return int(at[str(streams)]["hbm_bytes_per_launch"]), f"profiles/pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at {streams} legs"
        if at:
            k = min(at, key=lambda v: abs(int(v) - streams))
            return (int(at[k]["hbm_bytes_per_launch"] / int(k) * streams),
                    f"profiles/pmc_summary.json: counter passes at {k} legs, scaled to {streams} (not measured by this run)")
        return (int(d["hbm_bytes_per_launch"] / 4096 * streams),
                f"profiles/pmc_summary.json: counter passes over an 8-tick cycle of 4096 legs, scaled to {streams} (not measured by this run)")
    except Exception:
        return None


def make_resample_leg(ms, torch, ctx, nstreams):
    in_len, out_len = 160, 480
    per_tick = nstreams * (in_len + out_len) * 2
    ring = max(2, min(128, -(-(320 << 20) // per_tick)))
    rs = ms.ResamplerBatch(ctx, nstreams, 16000, 48000)
    host = synth_pcm_batch(nstreams, in_len * 4, 16000)
    ins = []
    for r in range(ring):
        o = (r % 4) * in_len
        ins.append(torch.from_numpy(np.ascontiguousarray(host[:, o:o + in_len])).cuda())
    outs = [torch.zeros((nstreams, out_len), dtype=torch.int16, device="cuda") for _ in range(ring)]
    torch.cuda.synchronize()

    def launch(i):
        rs.process(ins[i], out=outs[i])

    leg = Leg(ctx, "resample_up_kernel<3,48,8,false,false>", launch, ring, per_tick, nstreams, "stream-ticks")
    leg.keep = (rs, ins, outs, host)
    # the FIR is the work: 48 taps x 2 flop per output sample, issued as v_pk_fma_f32 (measured 1.96 ns per wave-instruction
    # per SIMD on this part: scripts/ubench/valu_rate.hip) -- the kernel is VALU-issue-bound at scale, not HBM-bound
    leg.valu_flop = 2.0 * 48 * nstreams * out_len
    leg.valu_peak_tflops = 1024 * 64 * 4 / 1.96e-9 / 1e12
    leg.valu_peak_name = "peak_packed_fma_fp32_tflops"
    return leg


def make_mixer_leg(ms, torch, ctx, nconf=128, mm=32, ns=480):
    per_tick = 2 * nconf * mm * ns * 2
    ring = max(2, min(64, -(-(320 << 20) // per_tick)))
    mx = ms.MixerBatch(ctx, nconf, mm, ns)
    base = synth_pcm_batch(nconf * mm, ns, 48000, sigma=1500.0).reshape(nconf, mm, ns)
    ins = [torch.from_numpy(np.roll(base, r, axis=2).copy()).cuda() for r in range(ring)]
    outs = [torch.zeros((nconf, mm, ns), dtype=torch.int16, device="cuda") for _ in range(ring)]

    def launch(i):
        mx.process(ins[i], None, 1, out=outs[i])

    leg = Leg(ctx, "mixer_members_kernel", launch, ring, per_tick, nconf, "conference-ticks")
    leg.keep = (mx, ins, outs)
    return leg


def make_volume_leg(ms, torch, ctx, nstreams=4096, ns=480):
    per_tick = nstreams * ns * 2 * 2
    ring = max(2, min(64, -(-(320 << 20) // per_tick)))
    vb = ms.VolumeBatch(ctx, nstreams, 48000)
    p = vb.default_params()
    p.agc_enabled = 1
    vb.set_params([p] * nstreams)
    base = synth_pcm_batch(nstreams, ns, 48000)
    bufs = [torch.from_numpy(np.roll(base, r, axis=1).copy()).cuda() for r in range(ring)]

    def launch(i):
        vb.process(bufs[i])

    leg = Leg(ctx, "volume_kernel", launch, ring, per_tick, nstreams, "stream-ticks")
    leg.keep = (vb, bufs)
    return leg


def make_equalizer_leg(ms, torch, ctx, nstreams=4096, ns=480):
    per_tick = nstreams * ns * 2 * 2
    ring = max(2, min(64, -(-(320 << 20) // per_tick)))
    eq = ms.EqualizerBatch(ctx, nstreams, 48000)
    eq.set_gain(0, 1000.0, 2.0, 500.0)
    taps = eq.taps(0)
    for s in range(1, nstreams):
        eq.set_taps(s, taps)
    base = synth_pcm_batch(nstreams, ns, 48000)
    bufs = [torch.from_numpy(np.roll(base, r, axis=1).copy()).cuda() for r in range(ring)]

    def launch(i):
        eq.process(bufs[i])

    leg = Leg(ctx, "equalizer_pk_kernel<512>", launch, ring, per_tick, nstreams, "stream-ticks")
    leg.keep = (eq, bufs)
    # this one is VALU-bound, not HBM-bound: 512 taps x 480 samples, multiply and add issued separately because the
    # reference's x86 build rounds the product (bit-exact output).  Peak = the unfused PACKED fp32 issue rate measured
    # on this part (scripts/ubench/valu_rate.hip: v_pk_*_f32 1.96 ns per wave-instruction per SIMD, two flops per lane
    # -> 66.9 Tflop/s over 1024 SIMDs; the scalar forms give 48.9).
    leg.valu_flop = 2.0 * nstreams * ns * 512
    leg.valu_peak_tflops = 1024 * 64 * 2 / 1.96e-9 / 1e12
    return leg


def make_scaler_leg(ms, torch, ctx, nframes=256, fmt=None):
    """1080p I420 -> 720p, to RGB24 (BASELINE configs[4]) or to I420 (fmt=MI_PIX_I420: what MSSizeConv asks of the scaler,
    sizeconv.c:97-184 -- libyuv ignores the destination format for I420 sources, msvideo.c:547-551).  256 frames per launch:
    configs[4]'s 2048 concurrent 1080p streams over 8 GPUs, one frame of each per launch (a launch per 33 ms frame period)."""
    sw, sh, dw, dh = 1920, 1080, 1280, 720
    fmt = ms.MI_PIX_RGB24 if fmt is None else fmt
    sc = ms.ScalerBatch(ctx, sw, sh, dw, dh, fmt)
    per_step = nframes * (sc.src_bytes + sc.dst_bytes)
    ring = 2  # 2 x 1.5 GB: far beyond the Infinity Cache
    rng = np.random.default_rng(0x5EED)
    yy, xx = np.mgrid[0:sh, 0:sw]
    y = (16 + 200 * (xx + yy) / (sw + sh)).astype(np.float32)
    frames = []
    for f in range(4):
        yf = (y + rng.normal(0, 12, y.shape)).clip(0, 255).astype(np.uint8)
        u = (128 + 100 * np.sin(2 * np.pi * np.arange(sw // 2) / (sw // 2) + f))[None, :].repeat(sh // 2, 0)
        v = (128 + 100 * np.cos(2 * np.pi * np.arange(sh // 2) / (sh // 2) + f))[:, None].repeat(sw // 2, 1)
        frames.append(np.concatenate([yf.ravel(), u.clip(0, 255).astype(np.uint8).ravel(),
                                      v.clip(0, 255).astype(np.uint8).ravel()]))
    four = torch.from_numpy(np.stack(frames)).cuda()
    ins = [four[torch.arange(nframes, device="cuda") % 4].contiguous() for _ in range(ring)]
    outs = [torch.zeros((nframes, sc.dst_bytes), dtype=torch.uint8, device="cuda") for _ in range(ring)]

    def launch(i):
        sc.process(ins[i], out=outs[i])

    leg = Leg(ctx, "scaler_wave_kernel<true>" if fmt == ms.MI_PIX_RGB24 else "scaler_wave_kernel<false>", launch, ring, per_step,
              nframes, "frames" if fmt == ms.MI_PIX_RGB24 else "frames (1080p I420 -> 720p I420)")
    leg.keep = (sc, ins, outs)
    leg.mpix_in = nframes * sw * sh / 1e6
    return leg


def make_pixconv_leg(ms, torch, ctx, nframes=64, fmt=None, w=1920, h=1080):
    """MSPixConv: packed YUY2 1080p frames -> I420 (the capture-side conversion, pixconv.c:62-94)."""
    fmt = ms.MI_PIX_YUY2 if fmt is None else fmt
    pc = ms.PixConvBatch(ctx, w, h, fmt)
    per_step = nframes * (pc.src_bytes + pc.dst_bytes)
    ring = 2
    rng = np.random.default_rng(7)
    host = rng.integers(0, 256, (4, pc.src_bytes), dtype=np.uint8)
    ins = [torch.from_numpy(np.ascontiguousarray(host[np.arange(nframes) % 4])).cuda() for _ in range(ring)]
    outs = [torch.zeros((nframes, pc.dst_bytes), dtype=torch.uint8, device="cuda") for _ in range(ring)]

    def launch(i):
        pc.process(ins[i], out=outs[i])

    leg = Leg(ctx, "pixconv_kernel<2>", launch, ring, per_step, nframes, "frames (1080p YUY2 -> I420)")
    leg.keep = (pc, ins, outs)
    leg.mpix_in = nframes * w * h / 1e6
    return leg


def make_g711_leg(ms, torch, ctx, nstreams=65536, n=480, law=None, encode=False):
    """MSAlawDec / MSUlawDec (alaw.c:208-221) or the encoders' sample loop (alaw.c:77-82): one block of n samples per
    stream per launch, 1 B of code word <-> 2 B of PCM per sample."""
    law = ms.MI_LAW_PCMA if law is None else law
    ring = 2
    g = torch.Generator(device="cpu").manual_seed(11)
    codes = [torch.randint(0, 256, (nstreams, n), dtype=torch.uint8, generator=g).cuda() for _ in range(ring)]
    pcm = [torch.zeros((nstreams, n), dtype=torch.int16, device="cuda") for _ in range(ring)]
    if encode:
        for i in range(ring):
            ms.g711_decode(ctx, law, codes[i], pcm[i])
        ctx.sync()

    def launch(i):
        if encode:
            ms.g711_encode(ctx, law, pcm[i], codes[i])
        else:
            ms.g711_decode(ctx, law, codes[i], pcm[i])

    name = ("g711_encode_kernel<%d>" if encode else "g711_decode_kernel<%d>") % law
    leg = Leg(ctx, name, launch, ring, nstreams * n * 3, nstreams, "stream-blocks (%d samples)" % n)
    leg.keep = (codes, pcm)
    return leg


def make_plc_leg(ms, torch, ctx, nstreams=65536, rate=8000, loss=0.05):
    """MSGenericPLC (msgenericplc.c:59-167) behind a G.711 decoder: one 10 ms block per leg and launch, a fraction `loss`
    of the legs missing theirs (alternating sets, so every loss is a FIRST loss: window, FFT nb, IFFT 2 nb)."""
    n = rate // 100
    plc = ms.PlcBatch(ctx, nstreams, rate, max_block=n)
    rng = np.random.default_rng(5)
    ring = 2
    rows = [torch.from_numpy(synth_pcm_batch(nstreams, n, rate)).cuda() for _ in range(ring)]
    lens = torch.full((nstreams,), n, dtype=torch.int32, device="cuda")
    pick = rng.random(nstreams) < 2 * loss
    half = rng.random(nstreams) < 0.5
    modes = [torch.from_numpy(np.where(pick & (half == bool(i)), ms.MI_PLC_CONCEAL, ms.MI_PLC_RECEIVED).astype(np.uint8)).cuda()
             for i in range(ring)]

    def launch(i):
        plc.process(rows[i], lens, modes[i])

    # per leg: the block in and out, the history write, the continuity buffer both ways
    leg = Leg(ctx, "plc_list_kernel+plc_received_kernel+plc_conceal_kernel", launch, ring, nstreams * (2 * n * 2 + n * 2 + 2 * (2 * rate * 5 // 1000) * 2), nstreams,
              "leg-ticks (%d samples, %.0f %% lost)" % (n, 100 * loss))
    leg.keep = (plc, rows, lens, modes)
    return leg


def make_aec_leg(ms, torch, ctx, nstreams=4096):
    """BASELINE configs[2] geometry: 48 kHz, 256-sample frames, 128 ms tail (M=24, N=512), post-filter on, in the form the
    chain runs it: ONE launch per 10 ms tick serving the one or two frames every leg has ready (15 frames per 8 ticks,
    speexec.c:256), canceller + post-filter in the same wavefront.  The leg's ring is one such 8-tick cycle."""
    rate, F = 48000, 256
    aec = ms.AecBatch(ctx, nstreams, rate, frame_size=F, filter_length=128 * rate // 1000)
    rng = np.random.default_rng(0x5EED)
    nfull, nb = nstreams, min(nstreams, 4096)  # distinct signals for 4096 streams, repeated for the rest
    far = synth_pcm_batch(nb, F * 8, rate)
    ir = rng.normal(0, 1, 64) * np.exp(-np.arange(64) / 12.0)
    ir /= np.sqrt((ir ** 2).sum())
    ring = 8
    reps = -(-nfull // nb)
    mics, refs = [], []
    for r in range(4):  # four distinct two-frame rows, used twice per cycle
        f = far[:, r * 2 * F:(r + 1) * 2 * F].astype(np.float32)
        echo = 0.5 * np.apply_along_axis(lambda v: np.convolve(v, ir)[:2 * F], 1, f[:256])
        mic = np.tile(echo, (nb // 256 + 1, 1))[:nb] + rng.normal(0, 300, (nb, 2 * F))
        mics.append(torch.from_numpy(np.clip(np.round(mic), -32767, 32767).astype(np.int16)).cuda().repeat(reps, 1)[:nfull].contiguous())
        refs.append(torch.from_numpy(np.ascontiguousarray(far[:, r * 2 * F:(r + 1) * 2 * F])).cuda().repeat(reps, 1)[:nfull].contiguous())
    out = torch.zeros((nfull, 2 * F), dtype=torch.int16, device="cuda")
    two = torch.full((nfull,), 2, dtype=torch.uint8, device="cuda")
    one = torch.full((nfull,), 1, dtype=torch.uint8, device="cuda")
    flags = ms.MI_AEC_POSTFILTER if os.environ.get("AEC_PROBE_POST", "1") != "0" else 0  # dev probes only (scripts/aec_phase_probe.sh)

    def launch(i):
        aec.process_frames(mics[i % 4], refs[i % 4], out, one if i % 8 == 7 else two, max_frames=2, flags=flags)

    # per 256-sample frame mic+ref+out 1536 B, W read+write 2x49152, foreground 49152, X history read 51200,
    # newest X block 2048 = 202 240 B; a launch (tick) carries 15/8 frames per leg on average
    leg = Leg(ctx, "aec_tick_kernel<256>", launch, ring, int(nfull * AEC_FRAMES_PER_TICK * AEC_FRAME_BYTES), nfull,
              "leg-ticks (15/8 frames of 256 samples each)")
    leg.keep = (aec, mics, refs, out, two, one)
    leg.state_bytes = aec.state_bytes() * nfull
    return leg


def make_aec_small_leg(ms, torch, ctx, rate, F, nstreams=65536):
    """The canceller at the small frame sizes (speexec.c:171-180: 8 kHz -> 64, 16 kHz -> 128 samples), 128 ms tail, post-filter on,
    one frame per launch handed in as rows (mi_aec_process: what the plugin's MSSpeexEC bank launches for such legs): several
    legs per wavefront (aec_group.hpp).  Bytes per frame as SURVEY 8(d) counts them: io + W read / write + foreground + X
    history + the newest block."""
    M = (128 * rate // 1000 + F - 1) // F
    N = 2 * F
    aec = ms.AecBatch(ctx, nstreams, rate, frame_size=F, filter_length=128 * rate // 1000)
    mic = torch.from_numpy(synth_pcm_batch(nstreams, F, rate)).cuda()
    ref = torch.from_numpy(synth_pcm_batch(nstreams, F, rate, sigma=2000.0)).cuda()
    out = torch.zeros_like(mic)
    torch.cuda.synchronize()
    per_frame = 3 * F * 2 + (3 * M * N + (M + 1) * N + N) * 4
    leg = Leg(ctx, f"aec_group_kernel<{F}>", lambda i: aec.process(mic, ref, out=out, flags=ms.MI_AEC_POSTFILTER), 1, nstreams * per_frame, nstreams,
              f"stream-frames ({F} samples at {rate} Hz, {M} filter blocks; {256 // F} legs per wavefront)")
    leg.keep = (aec, mic, ref, out)
    leg.state_bytes = aec.state_bytes() * nstreams
    return leg


def make_aec_small_fifo_leg(ms, torch, ctx, rate, F, nstreams=65536):
    """The same canceller through its FIFO ENTRY (mi_aec_process_fifos: what the plugin's fused legs and mi_session launch for 8 / 16 kHz
    legs): a whole 10 ms tick per launch sequence -- blocks queued, every whole frame popped, cancelled by aec_group_kernel, results
    queued (aec.hip: aec_fifos_group).  ns / F = 1.25 frames per leg and tick on average, the legs' re-framing phases spread."""
    M = (128 * rate // 1000 + F - 1) // F
    N, ns = 2 * F, rate // 100
    aec = ms.AecBatch(ctx, nstreams, rate, frame_size=F, filter_length=128 * rate // 1000)
    cap = (4 * ns + 4 * F + F - 1) // F * F
    fm, fr, fo = (ms.FifoBatch(ctx, nstreams, cap) for _ in range(3))
    aec.stagger_fifos(fm, fr, ns) if hasattr(aec, "stagger_fifos") else None
    ring = 8
    mic = [torch.from_numpy(synth_pcm_batch(nstreams, ns, rate, seed0=0x5EED + i)).cuda() for i in range(ring)]
    ref = [torch.from_numpy(synth_pcm_batch(nstreams, ns, rate, seed0=0xFA2 + i, sigma=2000.0)).cuda() for i in range(ring)]
    sink = torch.zeros((nstreams, ns), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    per_frame = 3 * F * 2 + (3 * M * N + (M + 1) * N + N) * 4

    def launch(i):
        aec.process_fifos(fm, mic[i % ring], fr, ref[i % ring], fo, tick_len=ns, max_frames=2, flags=ms.MI_AEC_POSTFILTER)
        fo.pop(ns, sink, zero_fill=True)   # (the mixer's side of the queue: keeps it from filling up)

    leg = Leg(ctx, f"aec_fifo_tick<{F}>", launch, ring, nstreams * per_frame * ns / F, nstreams,
              f"leg-ticks ({ns} samples at {rate} Hz = {ns / F:.2f} frames of {F}; FIFO entry -> aec_group_kernel)")
    leg.keep = (aec, fm, fr, fo, mic, ref, sink)
    leg.state_bytes = aec.state_bytes() * nstreams
    return leg


def copy_ceiling(torch):
    """Achievable HBM rate on this box: a 1 GiB device-to-device copy (read + write bytes / time), the
    'measured ceiling' BASELINE.md section 4 asks to report beside the 8 TB/s vendor peak."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(5):
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1)
        best = ms_ if best is None else min(best, ms_)
    del src, dst
    torch.cuda.empty_cache()
    return round(2 * n / (best * 1e-3) / 1e9, 1)


CHAIN_DESC = ("MSResample 16k->48k -> device FIFO (480-sample ticks -> 256-sample frames) -> MSSpeexEC (128 ms tail, "
              "canceller + post-filter; 15 frames per 8 ticks) -> device FIFO (frames -> ticks) -> MSVolume (AGC) -> "
              "MSAudioMixer (conferences of 32), device resident; TWO launches per tick: resampler + FIFO appends + canceller + "
              "post-filter, then volume + conference mix")
WORKLOAD_SHORT = ("north_star chain per call leg and 10 ms tick: MSResample 16k->48k -> MSSpeexEC 48 kHz 128 ms tail + post-filter -> "
                  "MSVolume AGC -> MSAudioMixer 32-party (configs[1]+[2]+[3] chained, echo scene, steady state)")
AEC_FRAME_BYTES = 202240  # mic+ref+out 1536, W read+write 2 x 49152, foreground 49152, X history 51200, newest X 2048
AEC_FRAMES_PER_TICK = 1.875  # 480 / 256 (speexec.c:171-180: 256-sample frames at 48 kHz)
SPLIT_CONFERENCES = 64  # at N > 1: conferences whose 32 members are spread over all ranks (the RCCL exchange step)
SCENE_BASE = 4096       # distinct echo scenes; leg s plays scene s % SCENE_BASE
SCENE_TICKS = 16        # period of a scene: 160 ms (longer than the 128 ms tail), two 8-tick framing cycles

_scene_cache = {}


def echo_scene(rank=0):
    """SURVEY 8(d)'s input for the canceller, SCENE_BASE distinct scenes of SCENE_TICKS ticks, periodic (the echo and the
    decimation wrap around, so a ring of SCENE_TICKS tick buffers plays for ever without a seam):
      far end at 48 kHz: N(0, 3000) + a 1 kHz tone at -20 dBFS;
      microphone: 0.5 x the far end, 20 ms late, through a fixed 64-tap decaying random impulse response, + independent
      near-end noise (sigma 300) -- low-passed and decimated to the 16 kHz the leg's resampler is fed with.
    Returns (mic16 [base][ticks * 160] int16, ref48 [base][ticks * 480] int16)."""
    if rank in _scene_cache:
        return _scene_cache[rank]
    n48 = SCENE_TICKS * 480
    ref = synth_pcm_batch(SCENE_BASE, n48, 48000, seed0=0x5EED + 7919 * rank).astype(np.float32)
    rng = np.random.default_rng(0xEC0 + rank)
    ir = rng.normal(0, 1, 64) * np.exp(-np.arange(64) / 12.0)
    ir /= np.sqrt((ir ** 2).sum())
    h = np.zeros(n48)
    h[960:960 + 64] = 0.5 * ir                                   # 20 ms of delay, then the room
    lp = np.sinc((np.arange(-48, 49)) * (7200.0 / 24000.0)) * (7200.0 / 24000.0) * np.hamming(97)  # anti-alias for / 3
    g = np.zeros(n48)
    g[:97] = lp
    g = np.roll(g, -48)                                          # zero phase
    H = np.fft.rfft(h)
    G = np.fft.rfft(g)
    mic16 = np.empty((SCENE_BASE, n48 // 3), np.int16)
    for i in range(0, SCENE_BASE, 256):
        R = np.fft.rfft(ref[i:i + 256], axis=1)
        near = rng.normal(0, 300.0, (min(256, SCENE_BASE - i), n48))
        mic48 = np.fft.irfft((R * H + np.fft.rfft(near, axis=1)) * G, n=n48, axis=1)
        mic16[i:i + 256] = np.clip(np.round(mic48[:, ::3]), -32767, 32767).astype(np.int16)
    _scene_cache[rank] = (mic16, ref.astype(np.int16))
    return _scene_cache[rank]


class ChainRig:
    """The north_star hot path for `n` concurrent 48 kHz call legs on ONE GPU, device resident from end to end
    (tests/test_gpu_pipeline.py checks the same chain stage by stage against the oracle).  One tick() = one 10 ms
    MSTicker interval (src/base/msticker.c:46) of every leg:
      MSResample 16k->48k (msresample.c:122-179) -> device FIFO -> MSSpeexEC at 256-sample frames, 128 ms tail,
      canceller + post-filter (speexec.c:171-180,223-305; one launch per tick for the one or two frames a leg has ready) -> device FIFO -> MSVolume with AGC (msvolume.c:471-514) -> MSAudioMixer,
      conferences of 32 (audiomixer.c:288-346).
    Input: SURVEY 8(d)'s echo scene (echo_scene above), leg s playing scene s % SCENE_BASE from a ring of SCENE_TICKS
    tick buffers.  stagger: the legs start with the product's re-framing lead (mi_aec_stagger_fifos), so every tick carries
    15/8 frames per leg; without it all legs cancel two frames in seven ticks of eight and one in the eighth.
    With world > 1 the last `nsplit` conferences of every rank are SPLIT ones: their 32 members are spread over all
    ranks (32 / world local members each), the rank computes int32 partial sums (mi_mixer_partial_sum), the caller
    all-reduces them (RCCL) and finalize() writes the local members' outputs (audiomixer.c:304-314 across GPUs)."""

    F, RATE, MEMBERS, RING = 256, 48000, 32, SCENE_TICKS

    def __init__(self, ms, torch, ctx, nstreams, world=1, rank=0, nsplit=0, stagger=True):
        self.ms, self.torch, self.ctx = ms, torch, ctx
        F, rate, mm = self.F, self.RATE, self.MEMBERS
        self.mloc = mm // world if nsplit else 0
        self.nsplit = nsplit
        self.staggered = bool(stagger)
        nsplit_streams = nsplit * self.mloc
        self.nconf = max(1, (nstreams - nsplit_streams) // mm)
        self.n = n = self.nconf * mm + nsplit_streams
        self.rs = ms.ResamplerBatch(ctx, n, 16000, rate)
        self.aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
        self.vol = ms.VolumeBatch(ctx, n, rate)
        p = self.vol.default_params()
        p.agc_enabled = 1
        self.vol.set_params([p] * n)
        self.mix = ms.MixerBatch(ctx, self.nconf, mm, 480)
        # whole frames.  Legs out of phase with each other can miss one pop at start-up and then run one frame fuller for
        # good (ms_bufferizer_read is all-or-nothing, msqueue.c:83): their output ring gets two frames more
        self.f_mic, self.f_ref = (ms.FifoBatch(ctx, n, 1024) for _ in range(2))
        self.f_out = ms.FifoBatch(ctx, n, 1536 if stagger else 1024)
        ring = self.RING
        mic16, ref48 = echo_scene(rank)
        base = min(n, SCENE_BASE)
        reps = -(-n // base)

        def spread(a, r, w):
            t = torch.from_numpy(np.ascontiguousarray(a[:base, r * w:(r + 1) * w])).cuda()
            return t.repeat(reps, 1)[:n].contiguous() if reps > 1 else t
        self.d_mic = [spread(mic16, r, 160) for r in range(ring)]
        self.d_ref = [spread(ref48, r, 480) for r in range(ring)]
        z = lambda *shape, dt=torch.int16: torch.zeros(shape, dtype=dt, device="cuda")
        self.cnt = z(n, dt=torch.uint8)  # frames each leg cancelled in the last tick
        self.tick_buf = z(n, 480)
        self.mixed = z(n, 480)
        nw = self.nconf * mm
        self.whole_in = self.tick_buf[:nw].view(self.nconf, mm, 480)
        self.whole_out = self.mixed[:nw].view(self.nconf, mm, 480)
        torch.cuda.synchronize()
        if stagger:
            self.aec.stagger_fifos(self.f_mic, self.f_ref, 480)
            ctx.sync()
        if nsplit:
            self.mixs = ms.MixerBatch(ctx, nsplit, self.mloc, 480)
            self.split_in = self.tick_buf[nw:].view(nsplit, self.mloc, 480)
            self.split_out = self.mixed[nw:].view(nsplit, self.mloc, 480)
            self.d_sum = z(nsplit, 480, dt=torch.int32)
        torch.cuda.synchronize()

    def tick(self, t, parts=None):
        """parts: a callback(stage) called before / after the canceller's launch (the roofline's HIP events)"""
        r = t % self.RING
        # MSResample + MSSpeexEC for the tick in ONE launch: the leg's wavefront up-samples its 16 kHz block (the resampler's
        # own tile FIR, history and table), queues it and the far-end block, cancels + post-filters the one or two whole frames
        # the leg then holds and queues the cleaned frames towards the mixer
        if parts:
            parts("aec_begin")
        self.aec.process_fifos_resampled(self.rs, self.d_mic[r], self.f_mic, self.f_ref, self.d_ref[r], self.f_out, max_frames=2,
                                         count_out=self.cnt)
        if parts:
            parts("aec_end")
        # MSVolume + MSAudioMixer of the whole conferences: one launch (ticks popped from the output FIFO, levelled, mixed)
        self.mix.process_volume_fifo(self.vol, self.f_out, self.whole_out)
        if self.nsplit:  # the split conferences' local members: levelled here, summed here, mixed after the exchange
            self.vol.process_fifo(self.f_out, self.tick_buf, first=self.nconf * self.MEMBERS)
            self.mixs.partial_sum(self.split_in, self.d_sum)

    def finalize(self):
        """after the all-reduce of d_sum: the split conferences' local outputs"""
        if self.nsplit:
            self.mixs.finalize(self.split_in, self.d_sum, self.split_out)

    def capture(self, ticks):
        self.ctx.capture_begin()
        for t in ticks:
            self.tick(t)
            if self.nsplit and len(ticks) > 1:
                raise RuntimeError("a tick with a collective in it is captured alone")
        return self.ctx.capture_end()

    def warm(self, nt=None):
        for t in range(nt or self.RING):  # whole scene periods: input ring and FIFO levels return to where they started
            self.tick(t)
            self.finalize()
        self.ctx.sync()

    def seed_from(self, base_rig):
        """every leg's canceller starts from the converged state of the base rig's leg playing the same scene
        (mi_aec_copy_state, on the device).  Call between whole scene periods of both rigs."""
        for first in range(0, self.n, base_rig.n):
            self.aec.copy_state_from(base_rig.aec, 0, first, min(base_rig.n, self.n - first))
        self.ctx.sync()

    def canceller_stats(self, sample=64):
        """(adapted fraction, foreground updates, frames) over `sample` legs spread over the batch"""
        ids = np.unique(np.linspace(0, self.n - 1, sample).astype(int))
        ad = [self.aec.get(int(i), "scalars", 16)[8] for i in ids]
        c = np.array([self.aec.get(int(i), "counters", 4) for i in ids])
        return float(np.mean(ad)), float(c[:, 0].sum()), float(c[:, 3].sum()), len(ids)

    def overflows(self):
        return self.f_mic.overflows() + self.f_ref.overflows() + self.f_out.overflows()

    def state_bytes(self):
        return int(self.aec.state_bytes() * self.n)

    def close(self):
        for o in ("rs", "aec", "vol", "mix", "f_mic", "f_ref", "f_out", "mixs"):
            if hasattr(self, o):
                getattr(self, o).close()
        self.__dict__.clear()


CONVERGE_TICKS = 20 * SCENE_TICKS  # 3.2 s of audio from reset: the cancellers of the base rig are `adapted` long before
SETTLE_TICKS = 4 * SCENE_TICKS     # after seeding: the far-end history of a seeded leg refills (24 blocks) and the two-path logic settles


class Converged:
    """The canceller in steady state for every leg of a rig: SCENE_BASE legs (one per distinct scene, product stagger)
    are run from reset for CONVERGE_TICKS on this GPU; seed() copies their state into every leg of a rig that plays the
    same scenes and runs SETTLE_TICKS more, so that what is timed afterwards is a converged, adapted canceller tracking
    a live echo -- the proportional step, the adapted step-size formula and the two-path updates at their steady rates."""

    def __init__(self, ms, torch, ctx, rank=0):
        self.base = ChainRig(ms, torch, ctx, SCENE_BASE, rank=rank)
        self.base.warm(CONVERGE_TICKS)
        self.adapted_fraction = self.base.canceller_stats()[0]

    def seed(self, rig):
        rig.seed_from(self.base)
        rig.warm(SETTLE_TICKS)

    def close(self):
        self.base.close()


class TickTimes(np.ndarray):
    """a series of tick durations (ms, HIP events on the launch stream); .submit: beside every tick the HOST's own time from
    before the first event's record to the return of the tick's launch call (perf_counter) -- a thread that is descheduled
    between the two lengthens the interval the events measure, and this says so (a tick the DEVICE took long over shows ~0.02 ms)"""

    def __new__(cls, n):
        obj = np.empty(n).view(cls)
        obj.submit = np.zeros(n)
        return obj

    def __array_finalize__(self, obj):
        self.submit = getattr(obj, "submit", None)


class quiet_interpreter:
    """around a timed series: Python's cyclic garbage collector stays off inside (a collection that drops a device buffer in the
    middle of a series frees it there and then -- the device is held while the driver unmaps it: 35-45 ms, seen).  It runs
    BEFORE the series instead, and `warm` (one untimed scene period of the same ticks) runs after it: the collection takes the
    host tens of ms, and a device that idled that long pays for it in its next two ticks (+1.7 / +0.6 ms, LOG.md)."""

    def __init__(self, warm=None):
        self.warm = warm

    def __enter__(self):
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        if self.warm:
            self.warm()

    def __exit__(self, *exc):
        if self.was:
            gc.enable()


def tick_series(ctx, graphs, nticks, after=None):
    """`nticks` CONSECUTIVE single ticks, each timed on its own with HIP events on the launch stream (the GPU drains after
    every tick: conservative).  No tick is discarded or repeated."""
    v = TickTimes(nticks)

    def period():  # (whole scene periods leave the rig's queues and input ring where they were)
        for g in graphs:
            g.launch()
    with quiet_interpreter(period):
        for t in range(nticks):
            t0 = time.perf_counter()
            ctx.timer_start()
            graphs[t % len(graphs)].launch()
            if after:
                after()
            v.submit[t] = (time.perf_counter() - t0) * 1e3
            v[t] = ctx.timer_stop()
    return v


def host_stalls_only(v):
    """True when EVERY late tick of the series is the submitting thread's doing: the host's own time between the first event's
    record and the return of the launch call (TickTimes.submit) covers at least 90 % of what the tick took beyond the series'
    median.  Such a tick says nothing about the leg count -- the same stall makes a tick of one leg late."""
    sub = getattr(v, "submit", None)
    a = np.asarray(v)
    late = np.flatnonzero(a >= 10.0)
    if sub is None or late.size == 0:
        return False
    p50, sub50 = float(np.median(a)), float(np.median(sub))
    return all(sub[i] - sub50 >= 0.9 * (a[i] - p50) for i in late)


def rig_period(head):
    return head.rig.RING


def series_stats(v):
    """late = ticks that reached the interval; slowest = [position in the series, ms] of the four longest (a machine
    event -- profiles/r03_outlier_probe.txt -- shows as one tick ~1 ms over the median and the next 0.2-0.4 over)"""
    sub = getattr(v, "submit", None)
    v = np.asarray(v)
    top = np.argsort(v)[-4:][::-1]
    out = {"ticks": int(v.size), "p50_ms": round(float(np.percentile(v, 50)), 4), "p99_ms": round(float(np.percentile(v, 99)), 4),
           "p99_9_ms": round(float(np.percentile(v, 99.9)), 4), "max_ms": round(float(v.max()), 4), "mean_ms": round(float(v.mean()), 4),
           "late": int((v >= 10.0).sum()), "slowest": [[int(i), round(float(v[i]), 3)] for i in top]}
    if sub is not None and len(sub) == v.size:
        # the host's share of the slowest ticks (TickTimes.submit): ~0.02 ms when the device took long, the excess itself when the
        # submitting thread was held up
        out["slowest_host_submit_ms"] = [round(float(sub[i]), 3) for i in top]
        out["host_submit_max_ms"] = round(float(np.max(sub)), 3)
    return out


def chain_capacity_point(ms, torch, ctx, nstreams, min_s=0.25, stagger=True, converged=None, worst_ticks=64):
    """avg and worst tick of the chain at `nstreams` legs on this GPU.  avg: one hipGraph of a scene period (16 ticks)
    replayed for at least `min_s` seconds; worst: `worst_ticks` consecutive single-tick graphs timed one by one (no
    overlap between ticks, the GPU drains after each: conservative; nothing discarded).  converged: a Converged to seed
    the legs' cancellers from (steady state); None = every leg starts from reset."""
    rig = ChainRig(ms, torch, ctx, nstreams, stagger=stagger)
    try:
        if converged is not None:
            converged.seed(rig)
        else:
            rig.warm()
        P = rig.RING
        gp = rig.capture(range(P))
        gp.launch()
        ctx.sync()
        ctx.timer_start()
        gp.launch()
        one = ctx.timer_stop()
        reps = max(1, int(np.ceil(min_s * 1e3 / max(one, 1e-3))))
        ctx.timer_start()
        for _ in range(reps):
            gp.launch()
        avg = ctx.timer_stop() / (P * reps)
        g1 = [rig.capture([t]) for t in range(P)]
        per = tick_series(ctx, g1, worst_ticks)  # (one untimed scene period first: quiet_interpreter)
        stalled = None
        a = np.asarray(per)
        # a point's series is measured once more when its late ticks say nothing about the count: every one of them the submitting
        # thread's stall (TickTimes.submit), or ONE tick more than 5 ms over the median -- no event of the power controller is that
        # long (<= 2.1 ms seen); some boxes hold the device for 35-45 ms once, early in a process's first sustained load
        # (profiles/r04_bench_low_sweep.json).  The sweep only proposes: the acceptance series decide, and excuse host stalls only.
        if host_stalls_only(per) or (int((a >= 10.0).sum()) == 1 and float(a.max() - np.median(a)) > 5.0):
            stalled = {"tick_ms_worst": round(float(a.max()), 4), "at": int(np.argmax(a)), "host_submit_ms": round(float(per.submit[int(np.argmax(a))]), 3)}
            per = tick_series(ctx, g1, worst_ticks)
        out = {"streams": rig.n, "conferences": rig.nconf, "tick_ms_avg": round(avg, 4), "tick_ms_worst": round(float(per.max()), 4),
               "tick_ms_single_median": round(float(np.median(per)), 4), "fits": bool(per.max() < 10.0),
               "fifo_overflows": int(rig.overflows()), "aec_resident_state_bytes": rig.state_bytes(),
               "state": "steady" if converged is not None else "from reset", "staggered": bool(stagger)}
        if stalled:
            out["first_series_held_a_stall"] = stalled
        del gp, g1
    finally:
        rig.close()
        PLATFORM.release(torch)
    return out


def find_capacity(ms, torch, ctx, lo=32768, hi=131072, coarse=8192, fine=2048, log=None, converged=None):
    """Largest stream count (multiple of `fine`) whose WORST tick of the chain stays under the 10 ms MSTicker interval.
    Coarse steps up from `lo` while the tick fits, then bisection down to `fine`.  Returns (streams, points measured)."""
    pts = []
    def fits(n):
        try:
            p = chain_capacity_point(ms, torch, ctx, n, converged=converged)
        except Exception as e:  # out of memory counts as "does not fit"
            p = {"streams": n, "fits": False, "error": str(e)[:160]}
        pts.append(p)
        if log:
            log(p)
        return p["fits"]
    good, bad = None, None
    n = lo
    while n <= hi:
        if fits(n):
            good = n
            n += coarse
        else:
            bad = n
            break
    if good is None:  # even `lo` is too many for this device: walk down
        n = lo // 2
        while n >= fine and not fits(n):
            bad = n
            n //= 2
        good = n if n >= fine else 0
    if bad is None:
        return good, pts
    while bad - good > fine:
        mid = (good + bad) // 2 // fine * fine
        if mid <= good or mid >= bad:
            break
        if fits(mid):
            good = mid
        else:
            bad = mid
    return good, pts


def pipeline_probe(ms, torch, ctx, nstreams):
    """one point of the capacity curve (dev tools: scripts/pipe_probe.py)"""
    out = chain_capacity_point(ms, torch, ctx, nstreams)
    out["chain"] = CHAIN_DESC
    return out


def session_probe(ms, ctx, nstreams, ticks=40, trunk=False):
    """mi_session end to end: host buffers in, host buffers out, uploads / kernels / downloads overlapped on three HIP
    streams, three ticks in flight -- the PCIe-inclusive rate of the chained path (never part of `value`).
    trunk: G.711 at 8 kHz in and out, far-end reference looped back on the device (160 B per leg and tick over PCIe)."""
    if trunk:
        se = ms.Session(ctx, nstreams, in_rate=8000, use_graphs=False, mic_codec=ms.MI_SESSION_PCMA, out_rate=8000,
                        out_codec=ms.MI_SESSION_PCMA, ref_loopback=True, ref_delay_ms=40)
        mic = np.random.default_rng(1).integers(0, 256, (nstreams, 80), dtype=np.uint8)
        ref = None
    else:
        se = ms.Session(ctx, nstreams, use_graphs=False)
        mic = synth_pcm_batch(nstreams, 160, 16000)
        ref = synth_pcm_batch(nstreams, 480, 48000, sigma=2000.0)
    per_tick = sum(se.tick_bytes()) * nstreams
    for _ in range(3):
        m, r = se.acquire()
        m[:] = mic
        if r is not None:
            r[:] = ref
        se.submit()
    for _ in range(3):
        se.collect()
    t0 = time.perf_counter()
    for _ in range(ticks):
        if se.in_flight() == 3:
            se.collect()
        se.acquire()
        se.submit()
    while se.in_flight():
        se.collect()
    dt = (time.perf_counter() - t0) / ticks
    se.close()
    note = "mi_session: pinned host buffers -> H2D | kernels | D2H on three streams, 3 ticks in flight"
    if trunk:
        note += "; trunk mode: PCMA 8 kHz in -> 48 kHz chain -> PCMA 8 kHz out, far-end reference looped back on the device"
    return {"streams": nstreams, "tick_ms_end_to_end": round(dt * 1e3, 4), "pcie_bytes_per_tick": int(per_tick),
            "fits": bool(dt < 0.010), "note": note}


def _host_cores():
    """threads the process may really use: affinity mask, clipped by the cgroup CPU quota"""
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max") and txt[0] != "max":
                quota = float(txt[0]) / float(txt[1])
            elif path.endswith("cfs_quota_us") and int(txt[0]) > 0:
                quota = int(txt[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota:  # a container may see every host CPU and still be allowed only a few cores' worth of time
        ncores = max(1, min(ncores, int(quota + 0.5)))
    return ncores, quota


def video_host_probe(ms, ctx, seconds=2.0, batch=32, depth=3):
    """BASELINE config 5 end to end on ONE GPU, PCIe included: 1080p I420 frames in pinned host buffers -> 720p RGB24 frames in
    pinned host buffers through mi_scaler_pipe (batches of `batch` frames, `depth` in flight; upload | kernel | download on
    three streams).  config 5's share per GPU is 2048 / 8 = 256 streams x 30 fps = 7 680 frames/s.  The producer writes into
    the staging in place (acquire hands out the pinned buffer: a decoder's output buffer), so no host copy is timed."""
    sc = ms.ScalerBatch(ctx, 1920, 1080, 1280, 720, ms.MI_PIX_RGB24)
    pipe = ms.ScalerPipe(sc, batch, depth)
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, sc.src_bytes, dtype=np.uint8)
    for _ in range(depth):  # every staging buffer of the ring holds frames; then the ring drains
        buf = pipe.acquire()
        buf[:, :sc.src_bytes] = frame
        pipe.submit(batch)
    first = None
    while pipe.in_flight():
        out = pipe.collect()
        if first is None:
            first = out[0, :sc.dst_bytes].copy()
    want = sc.process(frame[None, :])[0]
    exact = bool(np.array_equal(first, np.asarray(want)))
    nb = 0
    for _ in range(depth - 1):
        pipe.acquire()
        pipe.submit(batch)
    t0 = time.perf_counter()
    while True:
        pipe.acquire()
        pipe.submit(batch)
        pipe.collect()
        nb += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    while pipe.in_flight():
        pipe.collect()
    fps = nb * batch / dt
    pipe.close()
    sc.close()
    return {"frames_per_s": round(fps, 1), "h2d_GBps": round(fps * sc.src_bytes / 1e9, 2), "d2h_GBps": round(fps * sc.dst_bytes / 1e9, 2),
            "streams_1080p30": int(fps / 30), "fits_256x30fps": bool(fps >= 7680.0), "needed_frames_per_s": 7680,
            "batch_frames": batch, "batches_in_flight": depth, "bit_exact_vs_synchronous": exact, "mpix_per_s_in": round(fps * 1920 * 1080 / 1e6, 1),
            "what": "1080p I420 in pinned host memory -> 720p RGB24 in pinned host memory per frame, mi_scaler_pipe: upload | kernel | download "
                    "on three HIP streams; the producer fills the staging in place (no host copy timed); config 5 needs 7 680 frames/s per GPU "
                    "(23.9 GB/s up + 21.2 GB/s down)"}


def search_counts(measure, counts, start):
    """the largest of `counts` (ascending) that fits, by plugin_path_probe's rule: from counts[start] up a step at a time while the
    count fits (three more at most), else the counts below are bisected.  measure(count) -> {"fits": bool, ..}; -> the best run or None"""
    best, d = None, measure(counts[start])
    if d["fits"]:
        best = d
        for i in range(start + 1, min(start + 4, len(counts))):
            d = measure(counts[i])
            if not d["fits"]:
                break
            best = d
    else:  # (a noisy host must not cost a run per step)
        lo, hi = -1, start
        while hi - lo > 1:
            mid = (lo + hi) // 2
            d = measure(counts[mid])
            if d["fits"]:
                best, lo = d, mid
            else:
                hi = mid
    return best


def plugin_path_probe(first_legs, ticks=600, warmup=40, log=None, shape="", step_legs=8192, max_legs=98304, extras=True):
    """How many FULL call legs a mediastreamer2-shaped process carries through the DROP-IN PLUGIN (never part of `value`):
    tests/host/plugin_bench builds N legs of  source -> MSResample 16k->48k -> MSSpeexEC (128 ms) -> MSVolume (AGC) ->
    MSAudioMixer (conferences of 32)  from the factory's ids after libmsmi355xfilters_init (audiostream.c:1798-1810 in
    front of a conference mixer) on T ticker threads of the test runtime (one thread per MSTicker, as the reference runs
    them; T = the cores this process is granted, at most 16), every ticker PACED at one tick per 10 ms of wall time.

    The rules, all of them:
      * a run = `warmup` paced ticks from the attach on (reported apart as from_attach: a start-up stall is visible, not folded
        into capacity), then `ticks` (600) paced ticks; a tick costs what the slowest ticker needs;
      * a count FITS when no tick of the run reaches 10 ms (ticks_over_10ms == 0) and no ticker ever started a step a whole
        interval late (max_backlog_ms < 10, msticker_late_events == 0).  Strict: p99 is reported, not the criterion;
      * every count is measured the same way whether the search comes from below or from above: one run; the count the search STARTS
        at gets a second one if the first did not fit (the host's CPUs are shared; both runs are listed) -- it fits if either did
        (up to round 5's end every count got the second run: on a busy host the probe alone then took four minutes);
      * the search starts at `first_legs`, steps of 8192 legs (16 conferences of 32 per ticker): up a step at a time while the
        count fits (three more at most), else the counts below are bisected; `legs` is the largest count that fit -- ONE criterion
        (rounds 4 and 5 printed two more readings of the same runs beside it);
      * the tickers fire SPREAD over the interval (ticker k at k x 10 ms / T: a server's MSTickers start with their conferences and
        pace themselves from their own start, msticker.c:419-443) -- `phases` in every run's record says so;
      * `us_per_leg_tick_by_load`: what the boundary costs a host core, the figure that does not depend on the neighbours' ticks: the
        ticker threads' mean busy time per leg and tick at 1 024 / 2 048 / 3 072 legs per ticker, median of three paced runs each;
        `legs_per_host_core` = 10 ms over the figure at 2 048."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "plugin_bench")
    plugin = os.path.join(ROOT, "mediastreamer2_amd", "libmsmi355xfilters.so")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host"), "plugin_bench"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ncores, quota = _host_cores()
    tickers = max(1, min(16, ncores))
    keep = ("paced", "phases", "legs", "tickers", "ticks", "p50_ms", "p99_ms", "p99_9_ms", "max_ms", "ticks_over_10ms", "max_backlog_ms", "msticker_late_events", "fits",
            "us_per_leg_tick", "ticker_flush_ms", "ticker_graph_walk_ms", "launches_per_tick_and_ticker", "flush_rounds_per_tick_and_ticker",
            "fused_legs", "late_events", "worst_tick", "slow_ticks", "from_attach")

    def run(legs, nt, extra_env=None, paced=True):
        env = dict(os.environ)
        env.pop("MSMI355X_NO_FUSE", None)
        if shape:
            env["PLUGIN_BENCH_SHAPE"] = shape
        if paced:
            env["PLUGIN_BENCH_PACED"] = "1"  # every ticker fires at t0 + k x 10 ms of wall time, as an MSTicker does (msticker.c:419-443)
        env.update(extra_env or {})
        r = subprocess.run([exe, plugin, str(legs), str(tickers), str(nt), str(warmup)], capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0 or not r.stdout.strip():
            raise RuntimeError(f"plugin_bench exit {r.returncode}: {r.stderr[-300:]}")
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["ticks_over_10ms"] = d["late"]
        d["fits"] = bool(d["late"] == 0 and d["max_backlog_ms"] < 10.0 and d["msticker_late_events"] == 0)
        return d

    tried = []

    def measure(legs):
        """one count by the rule: a run, and -- for the count the search starts at -- a second one if the first did not fit"""
        best = None
        for _ in range(2 if not tried else 1):
            d = run(legs, ticks)
            tried.append({k: d.get(k) for k in keep})
            if log:
                log({"plugin_path" + ("_" + shape if shape else ""): {k: d.get(k) for k in keep if k != "slow_ticks"}})
            best = d if best is None or d["fits"] else best
            if d["fits"]:
                break
        return best

    step = max(tickers * 32, step_legs // (tickers * 32) * (tickers * 32))
    counts = list(range(step, max_legs + 1, step))
    start = min(range(len(counts)), key=lambda i: abs(counts[i] - first_legs))
    best = search_counts(measure, counts, start)
    out = {"cadence": "paced: one tick per 10 ms of wall time on every ticker, warm-up included",
           "what": "full call legs through the drop-in plugin, PCIe included: source -> MSResample 16k->48k -> MSSpeexEC (128 ms tail) -> MSVolume (AGC) "
                   "-> MSAudioMixer (32-party conference mode) + the far end into MSSpeexEC pin 0, filters created by id from the factory after "
                   "libmsmi355xfilters_init, one ticker thread per MSTicker in the test runtime (tests/host/plugin_bench.c); the plugin runs each "
                   "ticker's conferences as one device-resident batch (host/filters/leg_chain.inl); the test program grows glibc's arenas in 32 MB steps "
                   "(mallopt M_TOP_PAD: with the default 128 KB the first ticks after the attach are 65 % mprotect, profiles/r06_first_ticks.txt)",
           "fits_definition": f"no tick of {ticks} paced ticks reaches 10 ms, no step starts a whole interval late; tickers' phases spread over the interval; see the function's docstring",
           "host_cores_granted": ncores, "cgroup_cpu_quota_cores": quota, "ticks": ticks, "tried": tried}
    if extras:
        try:  # the cost per leg and tick at three loads per ticker thread: median of three paced runs each
            by_load = {}
            for lpt in (1024, 2048, 3072):
                us = sorted(run(tickers * lpt, 250)["us_per_leg_tick"] for _ in range(3))
                by_load[str(lpt)] = round(us[1], 4)
            out["us_per_leg_tick_by_load"] = by_load
            out["legs_per_host_core"] = int(10000.0 / max(by_load["2048"], 1e-3))
        except Exception as e:
            out["us_per_leg_tick_by_load"] = {"error": str(e)[:200]}
        try:  # conferences re-plumbed under load, as MSAudioConference does around every join and leave (audioconference.c:322-374)
            ch = run(tickers * 2048, 400, {"PLUGIN_BENCH_CHURN": "20"})
            out["churn"] = {"legs": ch["legs"], "legs_per_ticker": 2048, "ticks": ch["ticks"], "p50_ms": ch["p50_ms"], "p99_ms": ch["p99_ms"], "max_ms": ch["max_ms"],
                            "ticks_over_10ms": ch["late"], "replumbing": ch.get("churn"), "late_events": ch["late_events"], "fused_legs_at_the_end": ch.get("fused_legs"),
                            "what": "every ticker has one whole conference graph detached and attached again (32 legs: postprocess + preprocess of every filter, the fused "
                                    "batch left and joined again) 20 times a second, by an APPLICATION thread while the tickers run, as the reference's callers do: the detach "
                                    "waits for the ticker's lock (the tick in progress), postprocess and preprocess run outside it (msticker.c:153-221,:462-493).  "
                                    "replumbing.*_ms: what one took, that wait included"}
        except Exception as e:
            out["churn"] = {"error": str(e)[:200]}
    out["legs_strict"] = int(best["legs"]) if best is not None else 0
    if best is None:
        out.update({"fits": False, "legs": 0})
        return out
    out.update({k: best[k] for k in ("legs", "tickers", "p50_ms", "p99_ms", "p99_9_ms", "max_ms", "ticks_over_10ms", "max_backlog_ms", "msticker_late_events",
                                      "slow_ticks", "us_per_leg_tick", "from_attach")})
    out.update({"launches_per_tick": best["launches_per_tick_and_ticker"], "syncs_per_tick": best["flush_rounds_per_tick_and_ticker"], "fits": True,
                "where_the_time_goes": {"per_ticker_mean_ms": {"plugin_flush": best["ticker_flush_ms"], "graph_walk": best["ticker_graph_walk_ms"]}}})
    if "legs_per_host_core" not in out:  # (no by-load figure: from the run that fit -- the ticker threads' mean busy time per tick against the interval)
        busy = best["ticker_flush_ms"] + best["ticker_graph_walk_ms"]
        out["legs_per_host_core"] = int(best["legs"] / tickers * 10.0 / max(busy, 1e-3))
    if not extras:
        return out
    try:  # the walk by filter id (MS2SHIM_PROFILE: a timer around every process()): the plugin's facades vs the test runtime's sources and sinks
        pr = run(best["legs"], 300, {"MS2SHIM_PROFILE": "1"})
        by = pr.get("walk_us_per_leg_tick_by_filter_id", {})
        harness = sum(v for k, v in by.items() if k in ("9001", "9002"))
        out["walk_split_us_per_leg_tick"] = {"plugin_facades": round(sum(by.values()) - harness, 4), "harness_sources_and_sinks": round(harness, 4),
                                             "plugin_flush": round(pr["ticker_flush_ms"] * 1e3 * tickers / pr["legs"], 4), "by_filter_id": by,
                                             "ids": "41 MSResample, 28 MSSpeexEC, 43 MSVolume, 68 MSAudioMixer (+ the bank's enqueue), 9001 / 9002 the test runtime's sources / sinks"}
    except Exception as e:
        out["walk_split_us_per_leg_tick"] = {"error": str(e)[:200]}
    try:  # config[3]'s count fired back to back (every tick as soon as the slowest ticker is done: a throughput figure)
        b = run(32768, 600, paced=False)
        out["back_to_back"] = {k: b[k] for k in ("legs", "ticks", "p50_ms", "p99_ms", "max_ms", "ticks_over_10ms", "ticker_flush_ms",
                                                  "ticker_graph_walk_ms", "us_per_leg_tick", "slow_ticks")}
    except Exception as e:
        out["back_to_back"] = {"error": str(e)[:200]}
    try:  # the same graph with every facade on its own bank (MSMI355X_NO_FUSE=1: four uploads, launches and waits per chain), for scale
        d = run(4096, 200, {"MSMI355X_NO_FUSE": "1"}, paced=False)
        out["facades_one_by_one_4096_legs"] = {k: d[k] for k in ("legs", "tickers", "p50_ms", "max_ms", "us_per_leg_tick", "flush_rounds_per_tick_and_ticker")}
        d = run(4096, 200, paced=False)
        out["fused_4096_legs"] = {k: d[k] for k in ("legs", "tickers", "p50_ms", "max_ms", "us_per_leg_tick", "flush_rounds_per_tick_and_ticker")}
        # ... and that the two are the same audio: every leg's mix and speaker frames of a whole run folded into one number each
        a_, b_ = run(4096, 200, {"PLUGIN_BENCH_CHECKSUM": "1"}, paced=False), run(4096, 200, {"PLUGIN_BENCH_CHECKSUM": "1", "MSMI355X_NO_FUSE": "1"}, paced=False)
        out["fused_equals_one_by_one_4096_legs"] = {"equal": a_["mix_checksum"] == b_["mix_checksum"] and a_["speaker_checksum"] == b_["speaker_checksum"],
                                                    "mix_checksum": [a_["mix_checksum"], b_["mix_checksum"]], "mix_bytes": a_["mix_bytes"],
                                                    "what": "FNV-1a over every leg's mix (and speaker audio) of 240 ticks, byte for byte and in order, summed over the legs"}
    except Exception as e:
        out["facades_one_by_one_4096_legs"] = {"error": str(e)[:200]}
    return out


def plugin_shape_point(shape, legs=32768, ticks=250, warmup=40):
    """one paced run of tests/host/plugin_bench in another leg shape at a fixed count (detail file only: what the shape costs, not a capacity)"""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "plugin_bench")
    plugin = os.path.join(ROOT, "mediastreamer2_amd", "libmsmi355xfilters.so")
    ncores, _ = _host_cores()
    tickers = max(1, min(16, ncores))
    env = dict(os.environ, PLUGIN_BENCH_SHAPE=shape, PLUGIN_BENCH_PACED="1")
    env.pop("MSMI355X_NO_FUSE", None)
    r = subprocess.run([exe, plugin, str(legs), str(tickers), str(ticks), str(warmup)], capture_output=True, text=True, timeout=300, env=env)
    if r.returncode != 0 or not r.stdout.strip():
        raise RuntimeError(f"plugin_bench exit {r.returncode}: {r.stderr[-300:]}")
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out = {k: d.get(k) for k in ("legs", "tickers", "ticks", "fused_legs", "p50_ms", "p99_ms", "max_ms", "late", "us_per_leg_tick", "ticker_graph_walk_ms",
                                 "ticker_flush_ms", "launches_per_tick_and_ticker", "flush_rounds_per_tick_and_ticker", "late_events")}
    if isinstance(d.get("from_attach"), dict):
        out["from_attach"] = {k: d["from_attach"].get(k) for k in ("ticks", "p50_ms", "max_ms", "ticks_over_10ms", "first_ms")}
    return out


def cpu_baseline_chain(seconds, threads=1):
    """The oracle's CHAIN (CPU restatement of the reference path: one resampler, canceller + post-filter, volume object
    per call leg, one mixer per conference of 32, driven tick by tick as an MSTicker thread would) on this host's
    cores: a bounded sample of the headline workload."""
    import oracle
    oracle.build()
    L = oracle.lib()
    i16p = C.POINTER(C.c_int16)
    L.orc_bench_chain_mt.restype = C.c_double
    L.orc_bench_chain_mt.argtypes = [C.c_int] * 4 + [i16p, i16p, C.c_int, C.POINTER(C.c_longlong)]
    nconf = max(1, threads)
    mic = synth_pcm_batch(nconf * 32, 160, 16000)
    ref = synth_pcm_batch(nconf * 32, 480, 48000, seed0=0xFA2, sigma=2000.0)
    run = lambda nt: L.orc_bench_chain_mt(nconf, 32, nt, 128, mic.ctypes.data_as(i16p), ref.ctypes.data_as(i16p), threads, None)
    t = run(4)
    nticks = max(8, int(seconds / (t / 4)))
    t = run(nticks)
    per = t / (nconf * 32 * nticks) * threads  # core-seconds per stream-tick
    return {"value": round(nconf * 32 * nticks / t / TICKS_PER_S, 1),
            "unit": "concurrent 48 kHz streams", "cores": threads, "kind": "port",
            "sample_short": f"{nconf}x32 legs x {nticks} ticks of the oracle chain, {t:.1f} s wall",
            "sample": f"{nconf} conference(s) x 32 legs x {nticks} ticks of the chain (resample 16k->48k, MSSpeexEC 128 ms + "
                      f"post-filter, AGC, 32-party mix), oracle/*.c, {t:.1f} s wall on {threads} of {os.cpu_count()} host CPUs",
            "us_per_stream_tick_per_core": round(per * 1e6, 2)}


def cpu_reference_times():
    """The oracle (CPU restatement of the reference's process() bodies) timed per unit of work on ONE host core,
    small bounded samples (about a second each): what the same tick costs on the reference's CPU path."""
    import oracle
    oracle.build()
    L = oracle.lib()
    i16p, u8p, llp = C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.POINTER(C.c_longlong)
    for fn in ("orc_bench_mixer", "orc_bench_volume", "orc_bench_equalizer", "orc_bench_aec", "orc_bench_scaler"):
        getattr(L, fn).restype = C.c_double
    L.orc_bench_mixer.argtypes = [C.c_int] * 4 + [i16p, llp]
    L.orc_bench_volume.argtypes = [C.c_int] * 5 + [i16p, llp]
    L.orc_bench_equalizer.argtypes = [C.c_int] * 4 + [i16p, llp]
    L.orc_bench_aec.argtypes = [C.c_int] * 5 + [i16p, i16p, llp]
    L.orc_bench_scaler.argtypes = [C.c_int] * 5 + [u8p, llp]
    out = {}
    x = synth_pcm_batch(256, 480, 48000)
    p = lambda a_, t=C.c_int16: a_.ctypes.data_as(C.POINTER(t))
    t = L.orc_bench_mixer(8, 32, 480, 40, p(x), None)
    out["mixer_members_kernel"] = {"cpu_us_per_unit": round(t / (8 * 40) * 1e6, 2), "unit": "conference-tick (32 x 480)"}
    t = L.orc_bench_volume(256, 480, 40, 48000, 1, p(x), None)
    out["volume_kernel"] = {"cpu_us_per_unit": round(t / (256 * 40) * 1e6, 3), "unit": "stream-tick (480 samples, AGC)"}
    t = L.orc_bench_equalizer(32, 480, 10, 48000, p(x), None)
    out["equalizer_pk_kernel<512>"] = {"cpu_us_per_unit": round(t / (32 * 10) * 1e6, 2), "unit": "stream-tick (480 samples, 512 taps)"}
    mic = synth_pcm_batch(8, 256, 48000)
    ref = synth_pcm_batch(8, 256, 48000, sigma=2000.0)
    t = L.orc_bench_aec(8, 256, 128 * 48, 48000, 60, p(mic), p(ref), None)
    out["aec_tick_kernel<256>"] = {"cpu_us_per_unit": round(t / (8 * 60) * 1e6 * AEC_FRAMES_PER_TICK, 2),
                                   "unit": "leg-tick (15/8 frames of 256 samples, M=24, canceller + post-filter)"}
    # G.711: the reference's OWN conversions where oracle/_ref was built (kind "reference"), else the oracle's
    R = oracle.g711_ref()
    L.orc_bench_g711_decode.restype = L.orc_bench_g711_encode.restype = C.c_double
    L.orc_bench_g711_decode.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_int, llp]
    L.orc_bench_g711_encode.argtypes = [C.c_void_p, i16p, C.c_size_t, C.c_int, llp]
    codes = np.random.default_rng(2).integers(0, 256, 480 * 1024, dtype=np.uint8)
    pcm = oracle.g711_decode(0, codes)
    fd = C.cast(R.Snack_Alaw2Lin if R is not None else L.orc_alaw2lin, C.c_void_p)
    fe = C.cast(R.Snack_Lin2Alaw if R is not None else L.orc_lin2alaw, C.c_void_p)
    kind = "reference (src/audiofilters/g711.c compiled unmodified)" if R is not None else "port"
    t = L.orc_bench_g711_decode(fd, p(codes, C.c_uint8), codes.size, 20, None)
    out["g711_decode_kernel<0>"] = {"cpu_us_per_unit": round(t / (1024 * 20) * 1e6, 3), "unit": "stream-block (480 samples)", "kind": kind}
    t = L.orc_bench_g711_encode(fe, p(pcm), pcm.size, 20, None)
    out["g711_encode_kernel<0>"] = {"cpu_us_per_unit": round(t / (1024 * 20) * 1e6, 3), "unit": "stream-block (480 samples)", "kind": kind}
    frame = np.random.default_rng(1).integers(0, 256, 1920 * 1080 * 3 // 2, dtype=np.uint8)
    t = L.orc_bench_scaler(4, 1920, 1080, 1280, 720, p(frame, C.c_uint8), None)
    out["scaler_wave_kernel<true>"] = {"cpu_us_per_unit": round(t / 4 * 1e6, 1), "unit": "frame (1080p I420 -> 720p RGB24)"}
    return out


class HipPlatform:
    """The device-specific calls of the headline's control flow, in one place.  The product path is this class: HIP
    device, libmsmi355x kernels, RCCL exchange.  tests/bench_cpu_double.py substitutes a double (and a stand-in for
    ChainRig) to drive main()'s multi-rank control flow -- capacity agreement, step counts, barriers, the per-tick
    exchange, the split-mix check, exit codes -- over gloo on a box without a GPU."""
    device = "cuda"
    backend = "nccl"

    def available(self, torch):
        return torch.cuda.is_available()

    def select(self, torch, local):
        torch.cuda.set_device(local)

    def sync(self, torch):
        torch.cuda.synchronize()

    def release(self, torch):
        torch.cuda.empty_cache()

    def load(self):
        import mediastreamer2_amd as ms
        return ms

    def exchange(self, ctx, local, dist, rank, world, backend):
        """the split conferences' all-reduce: mi_exchange (C ABI, straight on RCCL, enqueued on the context's stream).
        torch.distributed only carries the 128-byte id from rank 0 to the others.  There is no substitute transport: if any
        rank cannot bring it up, every rank raises (main() exits 3).  Only the TEST backend (MSMI355X_BENCH_BACKEND=gloo:
        ranks sharing one GPU, which RCCL refuses) goes through torch, and the line says so."""
        if backend != "nccl":
            from mediastreamer2_amd.sharding import PartialSumExchange
            ex = PartialSumExchange(ctx.stream, local)
            ex.label = backend + " all-reduce (TEST BACKEND, not RCCL)"
            return ex
        import torch
        ms = self.load()
        ex, why = None, ""
        idt = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if rank == 0:
            try:
                idt.copy_(torch.frombuffer(bytearray(ms.Exchange.unique_id(ctx)), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001 -- the others must not wait for an id that never comes: zeros travel
                why = str(e)[:300]
        dist.broadcast(idt, 0)
        self.sync(torch)
        def agreed(flag):  # a MIN over the ranks (torch's communicator): every rank learns whether ALL of them got this far
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        try:
            if not bool(idt.any().item()):
                raise RuntimeError(why or "rank 0 could not make the communicator id")
            ex = ms.Exchange(ctx, world, rank, idt.cpu().numpy().tobytes())
        except Exception as e:  # noqa: BLE001 -- reported below, by every rank
            ex, why = None, str(e)[:300]
        if agreed(ex is not None):  # (a rank whose communicator did not start cannot take part in the probe: agree first)
            try:
                probe = torch.ones(4, dtype=torch.int32, device=self.device)  # every rank contributes 1: the sum is the world size
                self.sync(torch)
                ex(probe)
                ctx.sync()
                if probe.tolist() != [world] * 4:
                    raise RuntimeError(f"mi_exchange probe returned {probe.tolist()}, expected {world}")
            except Exception as e:  # noqa: BLE001
                ex.close()
                ex, why = None, str(e)[:300]
        elif ex is not None:
            ex.close()
            ex = None
        # ONE contract: the exchange is mi_exchange on RCCL or the run fails -- on every rank together (so that nobody waits in
        # a collective for a rank that has left): main() exits 3, RCCL's reason on stderr.  No substitute transport.
        if agreed(ex is not None):
            ex.label = "mi_exchange_allreduce_i32 (C ABI, RCCL over xGMI, on the kernel stream)"
            ex.label_short = "mi_exchange_allreduce_i32 (RCCL)"
            return ex
        if ex is not None:
            ex.close()
        raise RuntimeError(why or "another rank could not bring mi_exchange up")

    def converged(self, ms, torch, ctx, rank):
        return Converged(ms, torch, ctx, rank)


PLATFORM = HipPlatform()


def init_distributed(torch, rank, world, local):
    """One process per GPU.  The data path's collective is RCCL (backend "nccl"); there is no fallback: if RCCL cannot
    be brought up the run fails.  MSMI355X_BENCH_BACKEND=gloo exists only to exercise this control flow on a box with
    fewer GPUs than ranks (tests), and is named in config.parallelism."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("MSMI355X_BENCH_BACKEND", PLATFORM.backend)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        probe = torch.ones(1, dtype=torch.int32, device=PLATFORM.device)
        dist.all_reduce(probe)
        PLATFORM.sync(torch)
        if int(probe.item()) != world:
            raise RuntimeError(f"all-reduce probe returned {int(probe.item())}, expected {world}")
    except Exception as e:
        print(f"bench.py: rank {rank}: {backend} process group failed: {str(e)[:300]}", file=sys.stderr)
        sys.exit(3)
    return dist, backend


class Headline:
    """The timed region: K ticks of the chain at `nstreams` legs per GPU."""

    def __init__(self, ms, torch, ctx, nstreams, world, rank, dist, local, exchange=None):
        self.ms, self.torch, self.ctx, self.world, self.dist = ms, torch, ctx, world, dist
        self.rig = ChainRig(ms, torch, ctx, nstreams, world=world, rank=rank, nsplit=SPLIT_CONFERENCES if world > 1 else 0)
        self.exchange = exchange  # N > 1: the callable that all-reduces the split conferences' partial sums in place

    def prepare(self, warmup, converged=None):
        rig, P = self.rig, self.rig.RING
        if converged is not None:  # every canceller starts converged on its scene (steady state)
            rig.seed_from(converged.base)
            warmup = max(warmup, SETTLE_TICKS)
        for t in range(max(P, -(-warmup // P) * P)):  # whole scene periods: input ring and FIFO levels return to where they started
            self.eager_tick(t)
        self.ctx.sync()
        self.g1 = [rig.capture([t]) for t in range(P)]
        if self.world == 1:
            self.gp = rig.capture(range(P))
            self.gp.launch()  # untimed: uploads the graph, one more warm period
        else:
            self.ctx.capture_begin()
            rig.finalize()
            self.gfin = self.ctx.capture_end()
            for t in range(P):
                self.graph_tick(t)
        self.ctx.sync()

    def eager_tick(self, t, parts=None):
        self.rig.tick(t, parts) if parts else self.rig.tick(t)
        if self.exchange:
            self.exchange(self.rig.d_sum)
        self.rig.finalize()

    def graph_tick(self, t):
        self.g1[t % self.rig.RING].launch()
        self.exchange(self.rig.d_sum)
        self.gfin.launch()

    def run(self, steps):
        """EXACTLY `steps` ticks: whole scene periods as the period's graph, what is left tick by tick; returns HIP-event ms on the
        launch stream.  (A count that is no multiple of the period leaves the input ring mid-period: finish_period() runs the rest.)"""
        P = self.rig.RING
        self.ctx.timer_start()
        if self.world == 1:
            for _ in range(steps // P):
                self.gp.launch()
            for t in range(steps % P):
                self.g1[t].launch()
        else:
            for t in range(steps):
                self.graph_tick(t)
        return self.ctx.timer_stop()

    def finish_period(self, steps):
        """untimed: the ticks that complete the scene period a run of `steps` left open"""
        P = self.rig.RING
        for t in range(steps % P, P if steps % P else 0):
            if self.world == 1:
                self.g1[t].launch()
            else:
                self.graph_tick(t)
        self.ctx.sync()

    def tick_series(self, nticks):
        """`nticks` consecutive deployed ticks (with the exchange and the finalize launch at N > 1), each timed alone"""
        if self.world == 1:
            return tick_series(self.ctx, self.g1, nticks)
        v = TickTimes(nticks)

        def period():
            for t in range(rig_period(self)):
                self.graph_tick(t)
        with quiet_interpreter(period):
            for t in range(nticks):
                t0 = time.perf_counter()
                self.ctx.timer_start()
                self.graph_tick(t)
                v.submit[t] = (time.perf_counter() - t0) * 1e3
                v[t] = self.ctx.timer_stop()
        return v

    def paced_series(self, nticks, between=None):
        """`nticks` deployed ticks at an MSTicker's cadence: one per 10 ms of wall time (src/base/msticker.c:419-443,496-515),
        the device idle for the rest of each interval.  Each tick timed alone.  between(): called right before a tick's
        launches (scripts/paced_probe.py tries ways of keeping the device warm across the gap with it)."""
        v = TickTimes(nticks)

        def period():
            for t in range(rig_period(self)):
                if self.world == 1:
                    self.g1[t % len(self.g1)].launch()
                else:
                    self.graph_tick(t)
        with quiet_interpreter(period):
            nxt = time.perf_counter()
            for t in range(nticks):
                while time.perf_counter() < nxt:
                    pass
                nxt = max(nxt + 0.010, time.perf_counter() - 0.050)  # (a late tick is followed at once, like wait_next_tick does)
                if between:
                    between()
                t0 = time.perf_counter()
                self.ctx.timer_start()
                if self.world == 1:
                    self.g1[t % len(self.g1)].launch()
                else:
                    self.graph_tick(t)
                v.submit[t] = (time.perf_counter() - t0) * 1e3
                v[t] = self.ctx.timer_stop()
        return v

    def canceller_launches(self, nticks):
        """the canceller's launch inside the running chain, HIP events on the launch stream around that launch alone:
        (total ms, launches, frames cancelled).  Eager ticks (the events cannot sit inside a captured graph); the only host
        wait per tick is the one on the event behind the canceller's launch, the volume + mix launch that follows runs while
        the host queues the next tick, so the canceller starts right behind it and its clocks never drop.
        Frames: over whole 8-tick re-framing cycles every leg cancels exactly 15 (480 / 256 samples), whatever its phase;
        checked once against the per-leg counts the kernel reports before anything is timed."""
        rig, ctx = self.rig, self.ctx
        nticks = max(8, nticks // 8 * 8)
        counted = 0
        for t in range(8):
            rig.tick(t)
            rig.finalize()
            ctx.sync()
            counted += int(rig.cnt.sum().item())
        if counted != 15 * rig.n:
            raise RuntimeError(f"{counted} frames cancelled in an 8-tick cycle of {rig.n} legs, expected {15 * rig.n}")
        acc = []

        def parts(stage):
            if stage == "aec_begin":
                ctx.timer_start()
            else:
                acc.append(ctx.timer_stop())  # waits for the event behind the launch
        def cycle():  # (one untimed re-framing cycle: the device is busy again when the first timed launch starts)
            for t in range(8):
                rig.tick(t)
                rig.finalize()
        with quiet_interpreter(cycle):
            for t in range(nticks):
                rig.tick(t, parts)
                rig.finalize()
            ctx.sync()
        return float(sum(acc)), nticks, 15 * rig.n * (nticks // 8)

    def allreduce_alone_us(self, reps=200):
        """the exchange step by itself: event -> all-reduce of the [split conferences][480] int32 sums -> event"""
        if not self.exchange:
            return None
        for _ in range(10):
            self.exchange(self.rig.d_sum)
        self.ctx.sync()
        self.ctx.timer_start()
        for _ in range(reps):
            self.exchange(self.rig.d_sum)
        return self.ctx.timer_stop() * 1e3 / reps

    def check_split_mix(self, rank):
        """One more tick, then rank 0 gathers every rank's split-conference inputs, mixes the 32 members on ONE GPU
        (mi_mixer_process) and compares its own local outputs bit for bit."""
        torch, dist, rig = self.torch, self.dist, self.rig
        self.eager_tick(0)
        self.ctx.sync()
        mine = rig.split_in.contiguous().view(torch.uint8)  # NCCL has no int16: ship bytes
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        PLATFORM.sync(torch)
        ok = 1
        if rank == 0:
            full = torch.cat([p.view(torch.int16).view(rig.nsplit, rig.mloc, 480) for p in parts], dim=1).contiguous()
            mx = self.ms.MixerBatch(self.ctx, rig.nsplit, rig.MEMBERS, 480)
            ref = torch.zeros_like(full)
            PLATFORM.sync(torch)  # `full` and `ref` were produced on torch's stream, the mixer runs on the context's
            mx.process(full, out=ref)
            self.ctx.sync()
            ok = int(torch.equal(ref[:, :rig.mloc].contiguous(), rig.split_out.contiguous()))
            mx.close()
        flag = torch.tensor([ok], dtype=torch.int32, device=PLATFORM.device)
        dist.broadcast(flag, 0)
        return bool(flag.item())

    def close(self):
        for g in ("gp", "gfin"):
            if hasattr(self, g):
                getattr(self, g).close()
        for g in getattr(self, "g1", []):
            g.close()
        self.rig.close()
        PLATFORM.release(self.torch)

LINE_LIMIT = 6000  # the driver keeps the last 7 999 characters of stdout: the line stays well inside that


def _pick(d, *keys):
    """the named keys of a dict that are there and are numbers, booleans or short identifiers"""
    out = {}
    for k in keys:
        v = (d or {}).get(k)
        if isinstance(v, (bool, int, float)) or v is None and k in (d or {}):
            out[k] = v
        elif isinstance(v, str):
            out[k] = v[:96]
    return out


def short_line(full, detail_name):
    """The ONE stdout line: numbers and short identifiers only.  `full` (every series, every rejected count, every kernel's
    table row, every note) goes to the detail file; nothing is measured here."""
    cfg = full.get("config", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    if full.get("steps_requested") not in (None, full.get("steps")):
        out["steps_requested"] = full["steps_requested"]
    out["unit"] = "concurrent 48 kHz streams"
    c = {"workload": WORKLOAD_SHORT}
    c.update(_pick(cfg, "streams_per_gpu", "conferences_per_gpu", "worst_tick_ms", "single_tick_median_ms", "fits",
                   "tick_budget_used", "fifo_overflows", "launch", "device", "cu_count"))
    c["parallelism"] = cfg.get("parallelism_short", cfg.get("parallelism", ""))[:96]
    if "consecutive_ticks" in cfg:
        c["consecutive"] = _pick(cfg["consecutive_ticks"], "ticks", "p50_ms", "p99_ms", "p99_9_ms", "max_ms", "late")
    if "paced_ticks" in cfg:
        c["paced"] = _pick(cfg["paced_ticks"], "ticks", "p50_ms", "p99_ms", "max_ms", "late")
    if "from_reset" in cfg:
        c["from_reset"] = _pick(cfg["from_reset"], "worst_tick_ms", "ticks")
    if "steady_state" in cfg:
        c["steady_state"] = _pick(cfg["steady_state"], "adapted_fraction", "fg_updates_per_s")
    if cfg.get("consecutive_ticks_failed_at"):
        c["rejected_counts"] = [t.get("streams") for t in cfg["consecutive_ticks_failed_at"]][:12]
    if "split_conferences" in cfg:
        c["split_conferences"] = _pick(cfg["split_conferences"], "count", "members_per_rank", "allreduce_bytes_per_tick",
                                       "allreduce_alone_us", "mix_bit_exact_vs_single_gpu", "backend")
    c["detail"] = detail_name
    out["config"] = c
    if "roofline" in full:
        r = full["roofline"]
        out["roofline"] = _pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_src", "avg_launch_us",
                                "algorithmic_bytes_per_launch", "kernel", "timed_launches", "measured_copy_GBps", "error")
        out["roofline"]["units_per_launch"] = r.get("units_short")
        if "tick" in r:
            out["roofline"]["tick_frac"] = r["tick"].get("frac")
    if "cpu_baseline" in full:
        out["cpu_baseline"] = _pick(full["cpu_baseline"], "value", "unit", "cores", "kind", "us_per_stream_tick_per_core")
        out["cpu_baseline"]["sample"] = full["cpu_baseline"].get("sample_short", "")
        if "cpu_baseline_all_cores" in full:
            out["cpu_baseline"]["all_cores"] = _pick(full["cpu_baseline_all_cores"], "value", "cores")
    pp = full.get("plugin_path")
    if pp:
        out["plugin_path"] = _pick(pp, "legs", "tickers", "p50_ms", "p99_ms", "max_ms", "ticks_over_10ms", "us_per_leg_tick",
                                   "launches_per_tick", "syncs_per_tick", "max_backlog_ms", "host_cores_granted",
                                   "legs_per_host_core", "legs_strict", "host_cores_for_value", "fits", "error")
        if isinstance(pp.get("us_per_leg_tick_by_load"), dict):
            out["plugin_path"]["us_per_leg_tick_by_load"] = pp["us_per_leg_tick_by_load"]
        if isinstance(pp.get("churn"), dict):
            out["plugin_path"]["churn"] = _pick(pp["churn"], "legs", "p50_ms", "p99_ms", "max_ms", "ticks_over_10ms", "error")
        eq = pp.get("fused_equals_one_by_one_4096_legs")
        if eq:
            out["plugin_path"]["fused_equals_one_by_one"] = eq.get("equal")
        if isinstance(pp.get("from_attach"), dict):
            out["plugin_path"]["from_attach"] = _pick(pp["from_attach"], "ticks", "p50_ms", "max_ms", "ticks_over_10ms")
        if isinstance(pp.get("walk_split_us_per_leg_tick"), dict):
            out["plugin_path"]["walk_split_us"] = _pick(pp["walk_split_us_per_leg_tick"], "plugin_facades", "harness_sources_and_sinks", "plugin_flush")
        out["plugin_path"]["rejected_counts"] = sorted({t["legs"] for t in pp.get("tried", []) if not t.get("fits")})
    for key in ("plugin_path_server",):
        if full.get(key):
            out[key] = _pick(full[key], "legs", "tickers", "p50_ms", "p99_ms", "max_ms", "ticks_over_10ms", "us_per_leg_tick",
                             "launches_per_tick", "pcie_bytes_per_leg_tick", "fits", "error")
    if "scaler_mpix_per_s" in full:
        sc = full["scaler_mpix_per_s"]
        out["scaler"] = {"mpix_per_s": sc.get("value"), "frac": sc.get("hbm_frac"), "frames_per_s": sc.get("frames_per_s")}
    if "video_pcie_inclusive" in full:
        out["video_pcie_inclusive"] = _pick(full["video_pcie_inclusive"], "frames_per_s", "mpix_per_s_in", "h2d_GBps", "d2h_GBps", "error")
    for key in ("session_pcie_inclusive", "session_trunk_g711"):
        if key in full:
            out[key] = _pick(full[key], "streams", "tick_ms_end_to_end", "fits", "error")
    if full.get("other_kernels"):
        # kernel -> [avg launch us, frac of the HBM peak]; the table's rows are in the detail file
        out["other_kernels"] = {f"{r.get('kernel', '?')}@{str(r.get('units_per_launch', '')).split(' ')[0]}":
                                [r.get("avg_launch_us"), r.get("frac")] for r in full["other_kernels"]}
    return out


def emit(full, detail_path):
    """Write the full record beside the line (and say where on stderr), then print the short line -- the only stdout output."""
    try:
        with open(detail_path, "w") as f:
            json.dump(full, f, indent=1)
        print(f"bench.py: detail written to {detail_path} ({os.path.getsize(detail_path)} bytes)", file=sys.stderr, flush=True)
    except OSError as e:
        print(f"bench.py: the detail file could not be written ({e}); the full record follows on stderr", file=sys.stderr)
        print("bench.py: detail " + json.dumps(full), file=sys.stderr, flush=True)
    out = short_line(full, os.path.basename(detail_path))
    s = json.dumps(out, separators=(",", ":"))
    for k in ("other_kernels", "session_trunk_g711", "session_pcie_inclusive", "video_pcie_inclusive", "plugin_path_server"):
        if len(s) < LINE_LIMIT:
            break
        out.pop(k, None)  # (never reached with today's keys: the line is ~3 KB; a guard, so that growth cannot cut the head off)
        s = json.dumps(out, separators=(",", ":"))
    print(s, flush=True)


def main():
    t_main = time.time()
    a = parse()
    if not os.path.exists(os.path.join(ROOT, "mediastreamer2_amd", "libmsmi355x.so")):
        import __graft_entry__ as entry  # build artefacts are git-ignored: a fresh checkout compiles them first
        entry.build()
    import torch
    ms = PLATFORM.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # the driver launches N ranks for --gpus N; a bare `--gpus N` without torchrun is a usage error
        print(f"bench.py: --gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE is {world})", file=sys.stderr)
        sys.exit(2)
    if ChainRig.MEMBERS % world:
        print(f"bench.py: {world} ranks do not divide a 32-party conference", file=sys.stderr)
        sys.exit(2)
    if not PLATFORM.available(torch):
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    # one rank per GPU; MSMI355X_BENCH_DEVICE pins every rank to one device (only for exercising the N>1 control
    # flow on a single-GPU box together with MSMI355X_BENCH_BACKEND=gloo)
    if os.environ.get("MSMI355X_BENCH_DEVICE"):
        local = int(os.environ["MSMI355X_BENCH_DEVICE"])
    PLATFORM.select(torch, local)
    dist, backend = (None, None)
    if world > 1:
        dist, backend = init_distributed(torch, rank, world, local)

    ctx = ms.Context(local)
    props = ctx.props()
    exchange = None
    if world > 1:
        try:  # one communicator for the whole run; any RCCL failure ends the run (no fallback transport)
            exchange = PLATFORM.exchange(ctx, local, dist, rank, world, backend)
        except Exception as e:
            print(f"bench.py: rank {rank}: the conference exchange (mi_exchange on RCCL) could not be set up: {str(e)[:300]}", file=sys.stderr)
            sys.exit(3)

    def reduce_scalar(v, op):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=PLATFORM.device)
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    # ---- steady state: the base rig's cancellers converge on their echo scenes once; every rig measured below is seeded
    # from them (Converged).  --from-reset measures cancellers that start from reset instead (the first second of a call)
    converged = None if a.from_reset else PLATFORM.converged(ms, torch, ctx, rank)
    log = (lambda p: print("bench.py: sweep", json.dumps(p), file=sys.stderr, flush=True)) if rank == 0 else None

    # ---- capacity: the largest leg count whose worst tick fits the 10 ms interval (every rank measures its own GPU)
    sweep = []
    if a.streams > 0:
        streams = a.streams // 32 * 32
    else:
        streams, sweep = find_capacity(ms, torch, ctx, lo=a.sweep_lo, hi=a.sweep_hi, log=log, converged=converged)
    streams = int(reduce_scalar(float(streams), "MIN"))
    if streams <= 0:
        print("bench.py: no stream count fits the 10 ms tick on this device", file=sys.stderr)
        sys.exit(1)

    def accept(streams):
        """The count the sweep proposed is held to the acceptance series, and moved until the verdict changes.  ALL the rules, in one place:
          1. every leg starting from RESET at the same moment (the first second of adaptation), --zero-ticks ticks: none may reach 10 ms;
          2. --worst-ticks (3000) CONSECUTIVE ticks in steady state -- with the exchange at N > 1 --, each timed alone: none may reach the
             10 ms interval (a late tick is a fault, src/base/msticker.c:46,441-443; no tick is discarded);
          3. --paced-ticks (3000) ticks at the DEPLOYED cadence, one per 10 ms of wall time (only when 2 passed): none may reach 10 ms;
          4. a series whose ONLY late ticks were stalls of the submitting host thread (measured beside every tick) is no verdict on the
             count: the count is tried again, once, and must then pass whole; both series stay in the record;
          5. a count that fails is stepped DOWN to the count whose median leaves room for the failed series' longest tick (a tick costs in
             proportion to the legs, a machine event does not), at least 2048, 2048, 4096, 4096, .. legs; once --accept-seconds are spent
             the room is for the largest event seen on this hardware (1.8 ms, at most 2.5);
          6. a count that passes with its longest tick below 9.75 ms is stepped UP (one step, or to the count whose longest tick would be
             ~9.6 ms) -- at most three times, once after --accept-seconds; a step up that fails leaves the count that passed (a large step
             is halved once);
          7. --streams N: measured as given, no search;  at most twelve attempts in all.
        Returns (streams, zero, head, series, fg0, worst, paced, tried); head is None when nothing passed."""
        zero = head = series = fg0 = None
        worst = float("inf")
        paced, paced_on = None, a.paced_ticks > 0 and PLATFORM.device != "cpu"
        tried = []  # the counts that did not pass, with what they measured: the step-downs are part of the result
        best = None  # (streams, zero, head, series, fg0, worst) of a count that passed while a larger one is being tried
        ups = 0
        retried = set()  # counts tried a second time because the first series' late ticks were host stalls
        t_accept0 = time.perf_counter()
        for attempt in range(12):
            if a.zero_ticks > 0 and converged is not None:
                zero = chain_capacity_point(ms, torch, ctx, streams, converged=None, worst_ticks=a.zero_ticks)
                zero["tick_ms_worst"] = reduce_scalar(zero["tick_ms_worst"], "MAX")
                if log:
                    log(dict(zero, test="from reset"))
            zero_ok = zero is None or zero["tick_ms_worst"] < 10.0
            head = None
            if zero_ok or a.streams > 0 or streams <= 8192:
                head = Headline(ms, torch, ctx, streams, world, rank, dist, local, exchange)
                head.prepare(a.warmup, converged)
                fg0 = head.rig.canceller_stats() if hasattr(head.rig, "canceller_stats") else None
                # the read-backs above left the GPU idle for tens of ms and its clocks down: in service a tick follows the
                # previous one within a millisecond or two.  One untimed scene period brings the device back to its running
                # state; from there on every tick counts (scripts/outlier_probe.py: after 0.5 s of idle the first tick takes
                # +1.7 ms, the second +0.6, then nothing)
                head.tick_series(rig_period(head))
                series = head.tick_series(a.worst_ticks)
                worst = reduce_scalar(float(series.max()), "MAX")
                if log:
                    log({"streams": head.rig.n, "test": f"{a.worst_ticks} consecutive ticks", **series_stats(series)})
                # (3) the same number of ticks at the DEPLOYED cadence -- one per 10 ms of wall time, as an MSTicker fires them --
                # with the product's keep-alive between them; only run when the back-to-back series passed
                paced = None
                if paced_on and (worst < 10.0 or a.streams > 0):
                    paced = head.paced_series(a.paced_ticks)
                    if log:
                        log({"streams": head.rig.n, "test": f"{a.paced_ticks} paced ticks (one per 10 ms)", **series_stats(paced)})
                    worst = max(worst, reduce_scalar(float(paced.max()), "MAX"))
                if a.streams > 0 or streams <= 8192:
                    break
                if worst < 10.0 and zero_ok:
                    # passed.  The sweep's proposal can be low (one machine event among a point's 64 ticks fails the point): while
                    # the longest of the 3000 ticks leaves room for another step (2048 legs are ~0.15 ms), the next count up is
                    # held to the same two tests -- at most three times; a count that fails leaves the last one that passed
                    if best is not None:
                        best[2].close()
                        best = None
                    # (once --accept-seconds are spent on step-downs -- a box that kept producing events -- one step up is tried, not three: a try is a minute)
                    if worst < 9.75 and ups < (1 if time.perf_counter() - t_accept0 > a.accept_seconds else 3) and streams + 2048 <= a.sweep_hi:
                        best = (streams, zero, head, series, fg0, worst, paced)
                        ups += 1
                        # one step -- or, where the longest tick leaves a lot of room (the sweep's proposal was held down by one event), the
                        # count whose longest tick would be ~9.6 ms, a tick costing in proportion to the legs
                        streams = min(a.sweep_hi // 2048 * 2048, streams + max(2048, int(streams * (9.6 / worst - 1.0)) // 2048 * 2048))
                        continue
                    break
                tried.append({"streams": head.rig.n, **series_stats(series), "from_reset_worst_ms": zero and zero["tick_ms_worst"],
                              "paced": series_stats(paced) if paced is not None else None})
                head.close()
                head = None
                # a series whose only late ticks were stalls of the submitting thread (measured beside every tick) is no verdict on
                # the count: the count is tried again, ONCE, and must then pass whole; both series stay in the line
                mine = [x for x in (series, paced) if x is not None and float(np.max(x)) >= 10.0]    # this rank's failed series
                excused = zero_ok and all(host_stalls_only(x) for x in mine)                           # (none failed here: another rank's did)
                if reduce_scalar(1.0 if excused else 0.0, "MIN") > 0 and streams not in retried:
                    retried.add(streams)
                    tried[-1]["late_ticks_were_host_stalls"] = "the count is tried again"
                    continue
                if best is not None:  # the step up did not pass: the count below it stands ...
                    gap = streams - best[0]
                    if gap > 2048 and ups < 3:  # ... unless the step was a large one: half of it is tried (the count that passed stays in hand)
                        ups += 1
                        streams = best[0] + max(2048, gap // 2 // 2048 * 2048)
                        continue
                    streams, zero, head, series, fg0, worst, paced = best
                    best = None
                    break
                # the next count to try: the one whose median leaves room for this series' longest tick (a tick costs in
                # proportion to the legs; what a machine event adds does not) -- rounded UP to the step, so the estimate can
                # only be optimistic and the series at that count decides; never less than one step down
                decisive = paced if (paced is not None and paced.max() >= series.max()) else series  # the series that failed the count
                p50 = reduce_scalar(float(np.median(decisive)), "MAX")
                # (a box that keeps producing events would have the default run step down for a quarter of an hour, a minute a try:
                # once --accept-seconds are spent the next count leaves room for the LARGEST event seen on this hardware, 1.8 ms)
                over = time.perf_counter() - t_accept0 > a.accept_seconds
                # (... and no more than 2.5 ms: what exceeds the power controller's events -- a stall of the box -- would make a tick
                # of ANY count late, and "room for it" would send the search to the bottom of the range)
                event = min(max(worst - p50, 1.8) if over else worst - p50, 2.5)
                room = int(streams * max(9.95 - event, 1.0) / p50) // 2048 * 2048 + (0 if over else 2048)
                streams = min(streams - 2048 * (1 + attempt // 2), room)  # at least 2048, 2048, 4096, 4096, ... down
            else:
                tried.append({"streams": streams, "from_reset_worst_ms": zero["tick_ms_worst"]})
                if best is not None:
                    streams, zero, head, series, fg0, worst, paced = best
                    best = None
                    break
                streams -= 2048 * (1 + attempt // 2)
            streams = max(streams, 8192)
        if best is not None:  # (the attempts ran out on the way up: the last count that passed stands)
            if head is not None and head is not best[2]:
                head.close()
            streams, zero, head, series, fg0, worst, paced = best
        return streams, zero, head, series, fg0, worst, paced, tried

    streams, zero, head, series, fg0, worst, paced, tried = accept(streams)
    if head is None:
        # every attempt failed and the last one's rig is gone: the line reports what was tried, value 0 (nothing below is
        # measured on a closed rig)
        if rank == 0:
            emit({"metric": "concurrent 48 kHz streams/node at <10 ms tick; Mpix/s YUV scale", "value": 0,
                  "unit": "concurrent 48 kHz streams (resample + AEC + AGC + 32-party mix every 10 ms tick, worst tick < 10 ms)",
                  "n_gpus": world, "steps": 0, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "north_star chain per call leg and 10 ms tick: " + CHAIN_DESC, "fits": False,
                             "consecutive_ticks_failed_at": tried}}, a.detail)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(1)
    rig = head.rig
    median_single = float(np.median(series))
    stats = series_stats(series)
    for k in ("p50_ms", "p99_ms", "p99_9_ms", "max_ms"):
        stats[k] = round(reduce_scalar(stats[k], "MAX"), 4)  # the slowest rank's
    fg1 = rig.canceller_stats() if fg0 is not None else None

    def sync_local():
        ctx.sync()
        PLATFORM.sync(torch)

    def barrier():
        if dist is not None:
            dist.barrier()

    # EXACTLY K = --steps steps bracketed by (synchronize + barrier) on both sides.  A step is one 10 ms tick of every leg (a scene
    # period is 16 ticks: 30 canceller frames per leg; whole periods replay the period's graph, the rest go tick by tick).
    P = rig.RING
    steps = max(1, a.steps)
    if a.min_timed_s > 0:  # (asked for explicitly: the line then carries steps_requested beside steps)
        ctx.timer_start()
        head.run(P)
        est = ctx.timer_stop() / P
        steps = max(steps, int(np.ceil(a.min_timed_s * 1e3 / max(est, 1e-3))))
    if dist is not None:
        steps = int(reduce_scalar(float(steps), "MAX"))
    sync_local()
    barrier()
    t0 = time.perf_counter()
    ev_ms = head.run(steps)
    sync_local()
    dt = time.perf_counter() - t0
    barrier()
    head.finish_period(steps)
    dt = reduce_scalar(dt, "MAX")
    ev_ms = reduce_scalar(ev_ms, "MAX")

    total_streams = int(reduce_scalar(float(rig.n), "SUM"))
    tick_ms = dt / steps * 1e3
    fits = worst < 10.0 and (zero is None or zero["tick_ms_worst"] < 10.0)
    ar_us = head.allreduce_alone_us() if world > 1 else None
    split_ok = head.check_split_mix(rank) if world > 1 else None
    overflows = rig.overflows()
    n_local, nconf_local, state_bytes = rig.n, rig.nconf, rig.state_bytes()
    if world > 1 and not split_ok:
        print("bench.py: the split conferences' all-reduced mix differs from the single-GPU mix", file=sys.stderr)
        sys.exit(4)

    parallelism = parallelism_short = "1 GPU"
    if world > 1:
        parallelism_short = (f"{world} ranks: static shards + {SPLIT_CONFERENCES} split conferences over "
                             f"{getattr(exchange, 'label_short', backend + ' (TEST BACKEND)')}")
        parallelism = (f"{world} ranks, one per GPU; legs and whole conferences sharded statically (no collective), "
                       f"{SPLIT_CONFERENCES} conferences split over all ranks: int32 partial sums -> "
                       f"{getattr(exchange, 'label', backend + ' all-reduce (TEST BACKEND, not RCCL)')} "
                       "-> finalize, every tick")
    line = {
        "metric": "concurrent 48 kHz streams/node at <10 ms tick; Mpix/s YUV scale",
        "value": total_streams if fits else 0,
        "unit": "concurrent 48 kHz streams (resample + AEC + AGC + 32-party mix every 10 ms tick, worst tick < 10 ms)",
        "n_gpus": world, "steps": steps, "steps_requested": a.steps, "warmup": a.warmup,
        "ms_per_step": round(tick_ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "north_star chain per call leg and 10 ms tick: " + CHAIN_DESC + "; configs[2]'s canceller "
                               "geometry (48 kHz, 128 ms tail, post-filter) fed by configs[1]'s resampler and mixed as configs[3]; "
                               "input: SURVEY 8(d)'s echo scene (microphone = 0.5 x far end, 20 ms late, through a 64-tap room + noise)",
                   "value_definition": "largest leg count (capacity sweep in steady state, step 2048, then stepped down -- or up, while the series leaves room -- until the verdict changes) at which ALL hold: no tick of "
                                       f"{a.worst_ticks} consecutive back-to-back ticks in steady state reaches the 10 ms MSTicker interval, none of "
                                       f"{a.paced_ticks if paced is not None else 0} ticks PACED one per 10 ms of wall time does either (the deployed cadence), "
                                       "and neither does any tick when every leg starts from reset at once; ms_per_step = average "
                                       "tick over the timed region at that count",
                   "streams_per_gpu": n_local, "conferences_per_gpu": nconf_local + (SPLIT_CONFERENCES if world > 1 else 0),
                   "tick_ms": 10, "worst_tick_ms": round(worst, 4), "single_tick_median_ms": round(median_single, 4),
                   "consecutive_ticks": stats, "consecutive_ticks_failed_at": tried,
                   "fits": bool(fits), "tick_budget_used": round(tick_ms / 10.0, 4),
                   "rate_equivalent_streams": int(total_streams * 10.0 / tick_ms),
                   "fifo_overflows": int(overflows), "aec_resident_state_bytes_per_gpu": state_bytes,
                   "leg_phases": "product: mi_aec_stagger_fifos gives every leg a re-framing lead (a hash of its slot), the launch serves "
                                 "the legs sorted by the frames they have; every tick carries 15/8 frames per leg",
                   "working_set_note": "every tick streams the cancellers' resident state (far larger than the 256 MiB "
                                       "Infinity Cache); the input ring is one scene period (16 ticks)",
                   "launch": "hipGraph of 16 ticks replayed" if world == 1 else "hipGraph per tick + all-reduce + finalize graph",
                   "parallelism": parallelism, "parallelism_short": parallelism_short, "device": props["name"], "cu_count": props["cu_count"]},
    }
    if converged is not None and fg1 is not None:
        nleg = fg1[3]
        line["config"]["steady_state"] = {
            "value": total_streams if worst < 10.0 else 0, "worst_tick_ms": round(worst, 4),
            "adapted_fraction": round(fg1[0], 4),
            "fg_updates_per_s": round((fg1[1] - fg0[1]) / nleg / max(1e-9, (fg1[2] - fg0[2]) / nleg * 256.0 / 48000.0), 4),
            "how": f"{SCENE_BASE} legs (one per distinct scene) converged from reset for {CONVERGE_TICKS} ticks on this GPU "
                   f"(adapted fraction there: {round(converged.adapted_fraction, 4)}), their state copied into every leg "
                   f"(mi_aec_copy_state), {SETTLE_TICKS} ticks to settle, then the sweep / the consecutive ticks; "
                   "fg_updates_per_s = foreground := background events per leg and second of audio, from the kernel's counters "
                   f"over the consecutive ticks ({nleg} legs sampled)"}
    if zero is not None:
        line["config"]["from_reset"] = {"value": total_streams if zero["tick_ms_worst"] < 10.0 else 0, "worst_tick_ms": round(zero["tick_ms_worst"], 4),
                                        "tick_ms_avg": zero["tick_ms_avg"], "ticks": a.zero_ticks,
                                        "how": "every leg's canceller starts from reset in the same tick; the ticks right after"}
    if sweep and rank == 0:
        line["config"]["capacity_sweep"] = [{k: p.get(k) for k in ("streams", "tick_ms_avg", "tick_ms_worst", "fits", "error", "first_series_held_a_stall") if k in p}
                                            for p in sweep]
    if world > 1:
        line["config"]["split_conferences"] = {"count": SPLIT_CONFERENCES, "members_per_rank": rig.mloc,
                                               "allreduce_bytes_per_tick": SPLIT_CONFERENCES * 480 * 4,
                                               "allreduce_alone_us": round(ar_us, 2) if ar_us else None,
                                               "mix_bit_exact_vs_single_gpu": bool(split_ok), "backend": backend}

    # ---- roofline of the dominant kernel, live: the canceller's launch INSIDE the running chain at the headline's leg
    # count and in its state, HIP events on the launch stream around that launch alone (eager ticks), and the whole tick
    if rank == 0:
        try:
            ms_tot, launches, frames = head.canceller_launches(a.roofline_ticks)
            alg = frames * AEC_FRAME_BYTES / launches
            pmc = pmc_traffic_at("aec_tick_kernel", n_local)
            r = roofline(ms_tot, launches, alg, pmc[0] if pmc else None)
            r["kernel"] = "aec_tick_kernel<256>"
            r["units_per_launch"] = (f"{n_local} leg-ticks = {frames / launches:.0f} stream-frames on average "
                                     "(256 samples; 48 kHz, 128 ms tail, canceller + post-filter + FIFOs in one launch)")
            r["units_short"] = f"{n_local} leg-ticks = {frames / launches:.0f} frames of 256 samples"
            r["timed_launches"] = launches
            r["traffic_source"] = pmc[1] if pmc else None
            # (the short line says where `traffic` comes from: never measured by THIS run -- counters need rocprofv3 passes of their own)
            r["traffic_src"] = None if not pmc else ("profiles@%d" % n_local if " at %d legs" % n_local in pmc[1] else "profiles, scaled to %d legs" % n_local)
        except Exception as e:
            r = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "error": str(e)[:200]}
        tick_alg = n_local * (AEC_FRAMES_PER_TICK * AEC_FRAME_BYTES + 1280 + 1920 + 1920)
        r["tick"] = {"algorithmic_bytes_per_tick": int(tick_alg), "achieved": round(tick_alg / (ev_ms * 1e-3 / steps) / 1e9, 2),
                     "frac": round(tick_alg / (ev_ms * 1e-3 / steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "whole tick of the chain over the timed region: 1.875 canceller frames + resampler 1280 B + volume 1920 B + "
                             "mixer 1920 B per leg, HIP events on the launch stream"}
        r["mfma"] = ("not used: the one candidate, the scaler's 3x3 BT.601 colour matrix, is 9 integer MACs per pixel inside a "
                     "byte-streaming kernel at ~80 % of the measured copy ceiling; v_mfma_f32_16x16x4 would use 3 of 16 columns")
        line["roofline"] = r
        if paced is not None:
            line["config"]["paced_ticks"] = dict(series_stats(paced), cadence_ms=10,
                                                 note="one tick per 10 ms of wall time, as an MSTicker fires them (the device idles for the rest of each "
                                                      "interval and its clocks sag in every gap); part of `value`'s criterion: no tick of this series may "
                                                      "reach 10 ms either")
    head.close()
    if converged is not None:
        converged.close()
        PLATFORM.release(torch)

    if rank == 0 and world == 1:
        if not a.no_extras:
            extras = []
            ksteps = 96  # whole 8-tick cycles for the canceller leg

            def make_resample_4096(ms_, torch_, ctx_):  # BASELINE configs[1]
                return make_resample_leg(ms_, torch_, ctx_, 4096)

            def make_resample_65536(ms_, torch_, ctx_):  # same kernel, a deployment-sized batch
                return make_resample_leg(ms_, torch_, ctx_, 65536)

            def make_aec_4096(ms_, torch_, ctx_):  # BASELINE configs[2]
                return make_aec_leg(ms_, torch_, ctx_, 4096)

            def make_mixer_1024(ms_, torch_, ctx_):  # BASELINE configs[3] at its full size: 1024 conferences x 32 members
                return make_mixer_leg(ms_, torch_, ctx_, nconf=1024)

            def make_scaler_i420(ms_, torch_, ctx_):  # the MSSizeConv case: I420 in, I420 out
                return make_scaler_leg(ms_, torch_, ctx_, fmt=ms_.MI_PIX_I420)

            def make_g711_encode(ms_, torch_, ctx_):
                return make_g711_leg(ms_, torch_, ctx_, encode=True)

            def make_aec_8k(ms_, torch_, ctx_):
                return make_aec_small_leg(ms_, torch_, ctx_, 8000, 64)

            def make_aec_16k(ms_, torch_, ctx_):
                return make_aec_small_leg(ms_, torch_, ctx_, 16000, 128)

            def make_aec_fifo_8k(ms_, torch_, ctx_):
                return make_aec_small_fifo_leg(ms_, torch_, ctx_, 8000, 64)

            def make_aec_fifo_16k(ms_, torch_, ctx_):
                return make_aec_small_fifo_leg(ms_, torch_, ctx_, 16000, 128)

            no_pmc = (make_resample_65536, make_mixer_1024, make_scaler_i420, make_aec_8k, make_aec_16k, make_aec_fifo_8k, make_aec_fifo_16k)
            for mk in (make_resample_4096, make_resample_65536, make_mixer_leg, make_mixer_1024, make_volume_leg, make_equalizer_leg,
                       make_aec_4096, make_aec_8k, make_aec_16k, make_aec_fifo_8k, make_aec_fifo_16k, make_scaler_leg, make_scaler_i420, make_pixconv_leg, make_g711_leg, make_g711_encode, make_plc_leg):
                try:
                    lg = mk(ms, torch, ctx)
                    g = lg.run(ksteps, 3, use_graph=not a.no_graph)
                    ctx.sync()
                    reps = [lg.timed(ksteps, g) for _ in range(7)]
                    ms_ = min(reps)  # per-kernel table: best replay; the spread goes into `replay_stats_us`
                    ctx.sync()
                    r = roofline(ms_, ksteps, lg.alg_bytes, None if mk in no_pmc else pmc_traffic(lg.name, 4096 if mk is make_aec_4096 else None))
                    r["traffic_source"] = None if r["traffic"] is None else "profiles/pmc_summary.json"
                    r["kernel"] = lg.name
                    r["units_per_launch"] = f"{lg.units} {lg.unit_name}"
                    per = np.array(reps) * 1e3 / ksteps  # the reference's profiler prints count/min/mean/max/sd per filter
                    r["replay_stats_us"] = {"count": len(reps), "launches_per_replay": ksteps, "min": round(float(per.min()), 3),
                                            "mean": round(float(per.mean()), 3), "max": round(float(per.max()), 3),
                                            "sd": round(float(per.std()), 3)}  # (src/base/msfactory.c ms_factory_log_statistics)
                    if hasattr(lg, "state_bytes"):
                        r["resident_state_bytes"] = int(lg.state_bytes)
                    if hasattr(lg, "valu_flop"):
                        tf = lg.valu_flop / (ms_ * 1e-3 / ksteps) / 1e12
                        r["valu"] = {"flop_per_launch": int(lg.valu_flop), "achieved_tflops": round(tf, 2),
                                     getattr(lg, "valu_peak_name", "peak_unfused_packed_fp32_tflops"): round(lg.valu_peak_tflops, 1),
                                     "frac": round(tf / lg.valu_peak_tflops, 3)}
                    if hasattr(lg, "mpix_in"):
                        r["mpix_per_s_in"] = round(lg.mpix_in / (ms_ * 1e-3 / ksteps), 1)
                        if mk is make_scaler_leg:  # the metric's second half: 1080p I420 -> 720p RGB24 (configs[4])
                            line["scaler_mpix_per_s"] = {"value": r["mpix_per_s_in"], "unit": "Mpix/s of 1080p input, YUV420 -> RGB24 + bilinear 720p",
                                                         "frames_per_s": round(lg.units / (ms_ * 1e-3 / ksteps), 1),
                                                         "hbm_frac": r["frac"], "streams_1080p30": int(lg.units / (ms_ * 1e-3 / ksteps) / 30)}
                    extras.append(r)
                    del lg, g
                    torch.cuda.empty_cache()
                except Exception as e:  # an optional leg must never take the headline down
                    extras.append({"kernel": mk.__name__, "error": str(e)[:200]})
            line["other_kernels"] = extras
            try:
                line["roofline"]["measured_copy_GBps"] = copy_ceiling(torch)
            except Exception:
                line["roofline"]["measured_copy_GBps"] = None
            try:  # the same count with every leg's re-framing in the SAME phase (no stagger): seven heavy ticks and a light one
                p = chain_capacity_point(ms, torch, ctx, n_local, stagger=False, worst_ticks=64)
                line["config"]["legs_in_phase"] = {
                    "streams": p["streams"], "tick_ms_avg": p["tick_ms_avg"], "tick_ms_worst": p["tick_ms_worst"], "fits": p["fits"],
                    "note": "not `value`: what the same leg count costs when no leg is given a lead (all legs have two frames in "
                            "seven ticks of eight, one in the eighth) -- the case the product's stagger removes; from reset, 64 ticks"}
            except Exception as e:
                line["config"]["legs_in_phase"] = {"error": str(e)[:200]}
            if not a.no_session:
                for key, kw in (("session_pcie_inclusive", {}), ("session_trunk_g711", {"trunk": True})):
                    try:
                        line[key] = session_probe(ms, ctx, n_local, **kw)
                    except Exception as e:
                        line[key] = {"error": str(e)[:200]}
        if not a.no_video_host:
            try:
                line["video_pcie_inclusive"] = video_host_probe(ms, ctx)
            except Exception as e:
                line["video_pcie_inclusive"] = {"error": str(e)[:300]}
        if not a.no_plugin_path:
            try:
                line["plugin_path"] = plugin_path_probe(a.plugin_legs, log=log)
                try:  # a conference SERVER's remote members (volrecv -> mixer -> G.711 encoder, no canceller: filters/server_leg.inl)
                    sv = plugin_path_probe(65536, ticks=400, log=log, shape="server", step_legs=16384, max_legs=131072, extras=False)
                    sv["what"] = ("a conference server's REMOTE members through the plugin, PCIe included: 8 kHz source (decoder .. dtmfgen) -> MSVolume (volrecv) -> "
                                  "in_resampler -> MSAudioMixer (conferences of 32) -> out_resampler -> MSUlawEnc -> sink (audioconference.c:121-179,209-257); metered, "
                                  "queued, mixed and ENCODED in one batch per ticker")
                    sv["pcie_bytes_per_leg_tick"] = 160 + 80 + 4   # PCM up, G.711 down, the block's length
                    line["plugin_path_server"] = sv
                except Exception as e:
                    line["plugin_path_server"] = {"error": str(e)[:300]}
                shapes = {}   # the other leg shapes the fused chain takes, one paced point each at a fixed count
                for name, sh in (("mic_equalizer", "eq"), ("server_g711_decoder_heads", "server dec"), ("server_g711_endpoints_in_a_16k_conference", "server dec wb"),
                                 ("audiostreams_8k_g711_full_duplex", "astream"), ("audiostreams_8k_g711_default_features", "astream default")):
                    try:
                        shapes[name] = plugin_shape_point(sh)
                        if sh.startswith("astream") and (shapes[name].get("late") or 0) > 0:   # what DOES fit on this host: the same shape at half the streams
                            shapes[name]["at_16384_streams"] = _pick(plugin_shape_point(sh, legs=16384), "legs", "p50_ms", "p99_ms", "max_ms", "late", "us_per_leg_tick")
                    except Exception as e:
                        shapes[name] = {"error": str(e)[:200]}
                line["plugin_path_shapes"] = shapes
                pp = line["plugin_path"]
                if pp.get("legs"):  # the ticker threads (one core each) it would take to bring `value` legs through the boundary at this load per core
                    pp["host_cores_for_value"] = int(-(-line["value"] // max(1, pp["legs"] // pp["tickers"])))
            except Exception as e:
                line["plugin_path"] = {"error": str(e)[:300]}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_chain(a.cpu_seconds, threads=1)
            try:
                ncores, quota = _host_cores()
                line["cpu_baseline_all_cores"] = cpu_baseline_chain(min(a.cpu_seconds, 8.0), threads=ncores)
                line["cpu_baseline_all_cores"]["cgroup_cpu_quota_cores"] = quota
            except Exception as e:
                line["cpu_baseline_all_cores"] = {"error": str(e)[:200]}
            if not a.no_extras:
                try:
                    ref_t = cpu_reference_times()
                    for r in line.get("other_kernels", []):
                        if r.get("kernel") in ref_t and "avg_launch_us" in r:
                            r["cpu_port_1core"] = ref_t[r["kernel"]]
                except Exception as e:
                    line["cpu_reference_error"] = str(e)[:200]
    if rank == 0:
        line["bench_wall_s"] = round(time.time() - t_main, 1)   # (the whole command, every probe included: detail file)
        emit(line, a.detail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
