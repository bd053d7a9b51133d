"""The chained hot path, device-resident end to end: MSResample 16k->48k -> (device FIFO: 480-sample ticks -> 256-sample
frames) -> MSSpeexEC -> (device FIFO: frames -> ticks) -> MSVolume (AGC) -> MSAudioMixer (conferences of 32), one tick
after another with no host round trip between the stages -- the north_star path of BASELINE.json.

Parity is stage-wise: every stage's GPU output is compared with the oracle stage fed with the SAME input the GPU stage
consumed (read back from the device), so each bar is the stage's own (<= 1 LSB resampler, <= 1e-4 RMS canceller,
bit-exact volume / mixer / FIFO) and nothing hides behind an accumulated tolerance."""
import numpy as np
import pytest

import mediastreamer2_amd as ms
from conftest import synth_pcm

pytestmark = pytest.mark.gpu


class NpFifo:
    """MSBufferizer semantics (msqueue.c:70-113) for a batch, in numpy."""

    def __init__(self, n):
        self.q = [np.zeros(0, np.int16) for _ in range(n)]

    def push(self, x, count=None):
        for s in range(len(self.q)):
            k = x.shape[1] if count is None else int(count[s])
            self.q[s] = np.concatenate([self.q[s], x[s, :k]])

    def pop(self, frame, gate=None):
        n = len(self.q)
        out, ok = np.zeros((n, frame), np.int16), np.zeros(n, np.uint8)
        for s in range(n):
            if (gate is None or gate[s]) and len(self.q[s]) >= frame:
                out[s], self.q[s] = self.q[s][:frame], self.q[s][frame:]
                ok[s] = 1
        return out, ok


def test_fifo_matches_bufferizer_model(ctx):
    torch = pytest.importorskip("torch")
    n, cap = 37, 1000
    f, m = ms.FifoBatch(ctx, n, cap), NpFifo(n)
    rng = np.random.default_rng(1)
    out = torch.zeros((n, 520), dtype=torch.int16, device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    lv = torch.zeros(n, dtype=torch.int32, device="cuda")
    for step in range(60):
        blk = rng.integers(-32768, 32768, (n, 488), dtype=np.int16)
        cnt = rng.choice([0, 160, 480, 37], n).astype(np.int32)
        cnt[[len(q) + c > cap for q, c in zip(m.q, cnt)]] = 0          # the model never overflows; nor does the test
        d_blk, d_cnt = torch.from_numpy(blk).cuda(), torch.from_numpy(cnt).cuda()
        d_gate = None
        torch.cuda.synchronize()
        f.push(d_blk, nsamples=480, count=d_cnt)
        m.push(blk, cnt)
        frame = int(rng.choice([256, 480, 128]))
        gate = (rng.random(n) > 0.2).astype(np.uint8)
        d_gate = torch.from_numpy(gate).cuda()
        torch.cuda.synchronize()
        f.pop(frame, out, ok=ok, gate=d_gate, zero_fill=True)
        want, wok = m.pop(frame, gate)
        ctx.sync()
        np.testing.assert_array_equal(ok.cpu().numpy(), wok)
        np.testing.assert_array_equal(out.cpu().numpy()[:, :frame], want)
        f.levels(lv)
        ctx.sync()
        np.testing.assert_array_equal(lv.cpu().numpy(), [len(q) for q in m.q])
    assert f.overflows() == 0
    # a block that does not fit is refused, counted, and leaves the ring intact
    big = torch.zeros((n, cap), dtype=torch.int16, device="cuda")
    f.push(big, nsamples=cap)
    assert f.overflows() == sum(1 for q in m.q if len(q) > 0)
    f.close()


def test_fifo_frame_ops_for_a_whole_tick_match_the_frame_by_frame_ones(ctx):
    """pop_frames / push_frames (the while loop of speexec.c:256 for a whole tick in one launch) == the model popped and
    pushed one frame at a time: as many whole frames as a stream holds (<= 2), the far end asked for the same number with
    silence for the frames it cannot supply (speexec.c:261-272), rings of a capacity that is no power of two wrapping
    many times."""
    torch = pytest.importorskip("torch")
    n, cap, F = 29, 1480, 256
    f_mic, f_ref, f_out = (ms.FifoBatch(ctx, n, cap) for _ in range(3))
    m_mic, m_ref, m_out = NpFifo(n), NpFifo(n), NpFifo(n)
    rng = np.random.default_rng(5)
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    micf, reff, cnt, got = z(n, 2 * F), z(n, 2 * F), z(n, dt=torch.uint8), z(n, dt=torch.uint8)
    tick, okt = z(n, 480), z(n, dt=torch.uint8)
    for step in range(200):
        blk = rng.integers(-32768, 32768, (n, 480), dtype=np.int16)
        rblk = rng.integers(-32768, 32768, (n, 480), dtype=np.int16)
        rcnt = rng.choice([480, 480, 480, 0, 200], n).astype(np.int32)  # the far end sometimes runs short
        d_blk, d_rblk, d_rcnt = torch.from_numpy(blk).cuda(), torch.from_numpy(rblk).cuda(), torch.from_numpy(rcnt).cuda()
        torch.cuda.synchronize()
        f_mic.push(d_blk)
        f_ref.push(d_rblk, nsamples=480, count=d_rcnt)
        m_mic.push(blk)
        m_ref.push(rblk, rcnt)
        f_mic.pop_frames(F, 2, micf, nframes_out=cnt)
        f_ref.pop_frames(F, 2, reff, nframes_out=got, wanted=cnt, zero_fill=True)
        f_out.push_frames(micf, F, 2, cnt)  # what a canceller would hand on: here the frames themselves
        f_out.pop(480, tick, ok=okt, zero_fill=True)
        ctx.sync()
        w_cnt = np.zeros(n, np.uint8)
        w_mic, w_ref = np.zeros((n, 2 * F), np.int16), np.zeros((n, 2 * F), np.int16)
        w_got = np.zeros(n, np.uint8)
        for r in range(2):
            a, ok = m_mic.pop(F)
            b, okr = m_ref.pop(F, gate=ok)
            for s in range(n):
                if ok[s]:
                    w_mic[s, r * F:(r + 1) * F] = a[s]
                    w_ref[s, r * F:(r + 1) * F] = b[s]  # zeros when the far end was short
                    w_cnt[s] += 1
                    w_got[s] += okr[s]
            m_out.push(a, ok.astype(np.int32) * F)
        w_tick, w_okt = m_out.pop(480)
        c = cnt.cpu().numpy()
        np.testing.assert_array_equal(c, w_cnt)
        np.testing.assert_array_equal(got.cpu().numpy(), w_got)
        gm, gr = micf.cpu().numpy(), reff.cpu().numpy()
        for s in range(n):
            np.testing.assert_array_equal(gm[s, :int(c[s]) * F], w_mic[s, :int(c[s]) * F])
            np.testing.assert_array_equal(gr[s, :int(c[s]) * F], w_ref[s, :int(c[s]) * F])
        np.testing.assert_array_equal(okt.cpu().numpy(), w_okt)
        np.testing.assert_array_equal(tick.cpu().numpy(), w_tick)
    assert f_mic.overflows() + f_ref.overflows() + f_out.overflows() == 0
    assert (np.array([len(q) for q in m_mic.q]) < F).all()
    for f in (f_mic, f_ref, f_out):
        f.close()


def test_canceller_with_the_fifos_folded_in_equals_the_separate_launches(ctx):
    """mi_aec_process_fifos (blocks queued, frames popped, cancelled, results queued: ONE launch) == mi_fifo_push x 2,
    mi_fifo_pop_frames x 2, mi_aec_process_frames, mi_fifo_push_frames: FIFO levels, what the output FIFO delivers and the
    canceller's state, bit for bit -- with a far end that sometimes skips a block (silence is injected for the frames it
    cannot supply) and one that starts with a delay line of silence (MS_ECHO_CANCELLER_SET_DELAY), 48 and 16 kHz -- and
    44.1 kHz, whose 441-sample ticks are no multiple of anything the lanes own (every sample finds its own source), with
    far-end blocks of odd lengths."""
    torch = pytest.importorskip("torch")
    for rate, F, tail in ((48000, 256, 128), (16000, 128, 128), (44100, 256, 100)):
        n, ns, nticks = 12, rate // 100, 60
        cap = 4 * F
        flen = tail * rate // 1000
        M = (flen + F - 1) // F
        rng = np.random.default_rng(rate)
        mic = np.stack([synth_pcm(200 + s, ns * nticks, rate=rate, sigma=2500.0) for s in range(n)])
        ref = np.stack([synth_pcm(300 + s, ns * nticks, rate=rate, sigma=3000.0) for s in range(n)])
        z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")

        def rig():
            a = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen)
            fm, fr, fo = (ms.FifoBatch(ctx, n, cap) for _ in range(3))
            delay = z(n, 2 * F)
            gate = torch.from_numpy((np.arange(n) % 3 == 0).astype(np.uint8)).cuda()   # every third leg: a far-end delay line
            torch.cuda.synchronize()
            fr.push(delay, nsamples=F + 32, gate=gate)
            return a, fm, fr, fo

        a1, fm1, fr1, fo1 = rig()
        a2, fm2, fr2, fo2 = rig()
        micf, reff, clean, cnt = z(n, 2 * F), z(n, 2 * F), z(n, 2 * F), z(n, dt=torch.uint8)
        cnt2 = z(n, dt=torch.uint8)
        t1, t2, ok1, ok2 = z(n, ns), z(n, ns), z(n, dt=torch.uint8), z(n, dt=torch.uint8)
        lv1, lv2 = z(n, dt=torch.int32), z(n, dt=torch.int32)
        for t in range(nticks):
            dm = torch.from_numpy(np.ascontiguousarray(mic[:, t * ns:(t + 1) * ns])).cuda()
            rblk = ref[:, t * ns:(t + 1) * ns].copy()
            skip = rng.random(n) < 0.15           # the far end of these legs delivers nothing this tick ...
            dr = torch.from_numpy(rblk).cuda()
            short = rng.integers(1, ns, n) if rate == 44100 else np.full(n, ns)      # ... or a short block of any length
            rc = torch.from_numpy(np.where(skip, 0, np.where(rng.random(n) < 0.2, short, ns)).astype(np.int32)).cuda()
            torch.cuda.synchronize()
            # separate launches
            fm1.push(dm)
            fr1.push(dr, nsamples=ns, count=rc)
            fm1.pop_frames(F, 2, micf, nframes_out=cnt)
            fr1.pop_frames(F, 2, reff, wanted=cnt, zero_fill=True)
            a1.process_frames(micf, reff, clean, cnt, max_frames=2)
            fo1.push_frames(clean, F, 2, cnt)
            fo1.pop(ns, t1, ok=ok1, zero_fill=True)
            # folded in
            a2.process_fifos(fm2, dm, fr2, dr, fo2, tick_len=ns, max_frames=2, count_out=cnt2, ref_len=rc)
            fo2.pop(ns, t2, ok=ok2, zero_fill=True)
            for f1, f2 in ((fm1, fm2), (fr1, fr2), (fo1, fo2)):
                f1.levels(lv1)
                f2.levels(lv2)
                ctx.sync()
                np.testing.assert_array_equal(lv1.cpu().numpy(), lv2.cpu().numpy(), err_msg=f"rate {rate} tick {t}")
            ctx.sync()
            np.testing.assert_array_equal(cnt.cpu().numpy(), cnt2.cpu().numpy())
            np.testing.assert_array_equal(ok1.cpu().numpy(), ok2.cpu().numpy())
            np.testing.assert_array_equal(t1.cpu().numpy(), t2.cpu().numpy(), err_msg=f"rate {rate} tick {t}")
        for s_ in range(n):
            for what, ln in (("W", M * 2 * F), ("foreground", M * 2 * F), ("X", (M + 1) * 2 * F), ("E", 2 * F), ("power_1", F + 1), ("scalars", 16)):
                x, y = a1.get(s_, what, ln), a2.get(s_, what, ln)
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"rate {rate} stream {s_}: {what}"
        assert fm2.overflows() + fr2.overflows() + fo2.overflows() == 0
        for o in (a1, a2, fm1, fr1, fo1, fm2, fr2, fo2):
            o.close()


def test_chained_tick_pipeline_stage_parity(ctx, oracle):
    torch = pytest.importorskip("torch")
    nconf, mm = 2, 32
    n, nticks, F = nconf * mm, 40, 256
    rate = 48000
    rs = ms.ResamplerBatch(ctx, n, 16000, rate)
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
    vol = ms.VolumeBatch(ctx, n, rate)
    p = vol.default_params()
    p.agc_enabled = 1
    vol.set_params([p] * n)
    mix = ms.MixerBatch(ctx, nconf, mm, 480)
    f_mic, f_ref, f_out = (ms.FifoBatch(ctx, n, 2048) for _ in range(3))
    # oracle side: one object per stream per stage
    o_rs = [oracle.Resampler(16000, rate) for _ in range(n)]
    o_ec = [oracle.Echo(F, 128 * rate // 1000, rate) for _ in range(n)]
    o_pp = [oracle.Preproc(F, rate, o_ec[s]) for s in range(n)]
    o_vol = [oracle.Volume(rate) for _ in range(n)]
    for v in o_vol:
        v.v.agc_enabled = 1
    m_mic, m_ref, m_out = NpFifo(n), NpFifo(n), NpFifo(n)

    mic16 = np.stack([synth_pcm(s, 160 * nticks, rate=16000, sigma=2500.0) for s in range(n)])
    ref48 = np.stack([synth_pcm(1000 + s, 480 * nticks, rate=rate, sigma=3000.0) for s in range(n)])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    micf = torch.zeros((n, F), dtype=torch.int16, device="cuda")
    reff = torch.zeros((n, F), dtype=torch.int16, device="cuda")
    clean = torch.zeros((n, F), dtype=torch.int16, device="cuda")
    okm = torch.zeros(n, dtype=torch.uint8, device="cuda")
    tick = torch.zeros((n, 480), dtype=torch.int16, device="cuda")
    okt = torch.zeros(n, dtype=torch.uint8, device="cuda")
    up = torch.zeros((n, 488), dtype=torch.int16, device="cuda")
    mixed = torch.zeros((nconf, mm, 480), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()  # torch fills these on ITS stream; the kernels below run on the context's stream
    aec_sq, aec_cnt, frames_done, ticks_out = 0.0, 0, 0, 0
    for t in range(nticks):
        # ---- resample, device resident
        mic_t = dev(mic16[:, t * 160:(t + 1) * 160])
        torch.cuda.synchronize()
        rs.process(mic_t, out=up)
        ctx.sync()
        g_up = up.cpu().numpy()[:, :480]
        want = np.stack([o_rs[s].process(mic16[s, t * 160:(t + 1) * 160]) for s in range(n)])
        assert np.abs(g_up.astype(int) - want).max() <= 1
        f_mic.push(up, nsamples=480)
        ref_t = dev(ref48[:, t * 480:(t + 1) * 480])
        torch.cuda.synchronize()
        f_ref.push(ref_t)
        m_mic.push(g_up)                      # the model continues from what the GPU stage produced
        m_ref.push(ref48[:, t * 480:(t + 1) * 480])
        # ---- 480-sample ticks -> 256-sample frames: one or two frames per tick
        for _ in range(2):
            f_mic.pop(F, micf, ok=okm, zero_fill=False)
            f_ref.pop(F, reff, gate=okm, zero_fill=True)
            aec.process(micf, reff, out=clean, run=okm)
            f_out.push(clean, gate=okm)
            w_mic, w_ok = m_mic.pop(F)
            w_ref, _ = m_ref.pop(F, gate=w_ok)
            ctx.sync()
            np.testing.assert_array_equal(okm.cpu().numpy(), w_ok)
            if not w_ok.any():
                continue
            np.testing.assert_array_equal(micf.cpu().numpy()[w_ok == 1], w_mic[w_ok == 1])
            np.testing.assert_array_equal(reff.cpu().numpy()[w_ok == 1], w_ref[w_ok == 1])
            g_clean = clean.cpu().numpy()
            w_clean = np.zeros((n, F), np.int16)
            for s in range(n):
                if w_ok[s]:
                    w_clean[s] = o_pp[s].run(o_ec[s].cancel(w_mic[s], w_ref[s]))
            d = (g_clean[w_ok == 1].astype(np.float64) - w_clean[w_ok == 1]) / 32768.0
            aec_sq += float((d * d).sum())
            aec_cnt += d.size
            frames_done += 1
            m_out.push(g_clean, count=w_ok.astype(np.int32) * F)
        # ---- frames -> ticks, AGC, conference mix
        f_out.pop(480, tick, ok=okt, zero_fill=True)
        w_tick, w_okt = m_out.pop(480)
        ctx.sync()
        np.testing.assert_array_equal(okt.cpu().numpy(), w_okt)
        np.testing.assert_array_equal(tick.cpu().numpy(), w_tick)
        vol.process(tick)
        ctx.sync()
        g_vol = tick.cpu().numpy()
        w_vol = np.stack([o_vol[s].chunk(w_tick[s]) for s in range(n)])
        np.testing.assert_array_equal(g_vol, w_vol)
        mix.process(tick.view(nconf, mm, 480), out=mixed)
        ctx.sync()
        g_mix = mixed.cpu().numpy()
        for c in range(nconf):
            w_mix, _ = oracle.mixer_tick(g_vol[c * mm:(c + 1) * mm])
            np.testing.assert_array_equal(g_mix[c], w_mix)
        ticks_out += int(w_okt.all())
    assert frames_done == (480 * nticks) // F               # 1.875 frames per tick
    assert ticks_out >= nticks - 2                          # the frame->tick FIFO delays the first ticks only
    assert np.sqrt(aec_sq / aec_cnt) <= 1e-4
    assert f_mic.overflows() == f_ref.overflows() == f_out.overflows() == 0


def test_the_headline_two_launch_tick_against_the_oracle_chain(ctx, oracle):
    """The tick exactly AS THE HEADLINE RUNS IT -- mi_aec_process_fifos_resampled (the leg's MSResample 16k->48k, both FIFO
    appends, every whole 256-sample frame through canceller + post-filter, the output FIFO append: one launch) and
    mi_mixer_process_volume_fifo (FIFO pop, AGC, 32-party mix: one launch) -- at the headline's own configuration: 128 ms
    tail (24 filter blocks), conferences of 32, the product's stagger, SURVEY 8(d)'s echo scene (bench.echo_scene), long
    enough (416 ticks) for every canceller to adapt and for both kinds of filter copy to occur.  Held DIRECTLY to the chain
    of oracle objects, stage by stage on what the two launches themselves queued (mi_fifo_snapshot reads the rings as they
    lie): <= 1 LSB for the up-sampled block, <= 1e-4 RMS for the cleaned frames (speexec.c:297-298), bit-exact meter +
    mix given those frames (msvolume.c:471-514, audiomixer.c:288-346)."""
    torch = pytest.importorskip("torch")
    import bench
    nconf, mm, F, rate, ns, nticks = 2, 32, 256, 48000, 480, 416
    n, flen = nconf * mm, 128 * rate // 1000
    mic16, ref48 = bench.echo_scene()
    P = bench.SCENE_TICKS
    rs = ms.ResamplerBatch(ctx, n, 16000, rate)
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen)
    vol = ms.VolumeBatch(ctx, n, rate)
    p = vol.default_params()
    p.agc_enabled = 1
    vol.set_params([p] * n)
    mix = ms.MixerBatch(ctx, nconf, mm, ns)
    cap = (2 * ns + 3 * F + F - 1) // F * F
    f_mic, f_ref, f_out = (ms.FifoBatch(ctx, n, cap) for _ in range(3))
    aec.stagger_fifos(f_mic, f_ref, ns)                       # the product's re-framing leads
    lead = [32 * ctx.L.mi_fifo_phase_of(s, 8) for s in range(n)]
    assert len(set(lead)) == 8
    o_rs = [oracle.Resampler(16000, rate) for _ in range(n)]
    o_ec = [oracle.Echo(F, flen, rate) for _ in range(n)]
    o_pp = [oracle.Preproc(F, rate, o_ec[s]) for s in range(n)]
    o_vol = [oracle.Volume(rate) for _ in range(n)]
    for v in o_vol:
        v.v.agc_enabled = 1
    q_mic = [np.zeros(lead[s], np.int16) for s in range(n)]
    q_ref = [np.zeros(lead[s], np.int16) for s in range(n)]
    q_out = [np.zeros(0, np.int16) for _ in range(n)]
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    cnt, mixed = z(n, dt=torch.uint8), z(nconf, mm, ns)
    d_mic = [torch.from_numpy(np.ascontiguousarray(mic16[:n, k * 160:(k + 1) * 160])).cuda() for k in range(P)]
    d_ref = [torch.from_numpy(np.ascontiguousarray(ref48[:n, k * ns:(k + 1) * ns])).cuda() for k in range(P)]
    torch.cuda.synchronize()

    def tail_of(snap, s, k):
        """the k samples most recently queued on stream s"""
        rings, head, level = snap
        end = int(head[s]) + int(level[s])
        idx = (np.arange(end - k, end)) % rings.shape[1]
        return rings[s, idx]

    sq, cnt_s, frames, up_worst, two_frame_ticks = 0.0, 0, 0, 0, 0
    for t in range(nticks):
        k = t % P
        aec.process_fifos_resampled(rs, d_mic[k], f_mic, f_ref, d_ref[k], f_out, max_frames=2, count_out=cnt)
        ctx.sync()
        s_mic, s_out = f_mic.snapshot(), f_out.snapshot()
        g_cnt = cnt.cpu().numpy()
        for s in range(n):
            # (the consumed frames still lie in the ring behind the read position: the tick's block is the last ns queued)
            rings, head, level = s_mic
            end = int(head[s]) + int(level[s])
            g_up = rings[s, np.arange(end - ns, end) % cap]
            want = o_rs[s].process(mic16[s, k * 160:(k + 1) * 160])[:ns]
            up_worst = max(up_worst, int(np.abs(g_up.astype(int) - want.astype(int)).max()))
            q_mic[s] = np.concatenate([q_mic[s], g_up])            # the oracle continues from what the launch queued
            q_ref[s] = np.concatenate([q_ref[s], ref48[s, k * ns:(k + 1) * ns]])
            nf = 0
            w_clean = []
            while len(q_mic[s]) >= F and nf < 2:                   # speexec.c:256
                m, q_mic[s] = q_mic[s][:F], q_mic[s][F:]
                r, q_ref[s] = q_ref[s][:F], q_ref[s][F:]
                w_clean.append(o_pp[s].run(o_ec[s].cancel(m, r)))
                nf += 1
            assert nf == int(g_cnt[s]), (t, s)
            two_frame_ticks += nf == 2
            if nf:
                g_clean = tail_of(s_out, s, nf * F)
                d = (g_clean.astype(np.float64) - np.concatenate(w_clean)) / 32768.0
                sq += float((d * d).sum())
                cnt_s += d.size
                frames += nf
                q_out[s] = np.concatenate([q_out[s], g_clean])
        mix.process_volume_fifo(vol, f_out, mixed)
        ctx.sync()
        g_mix = mixed.cpu().numpy()
        w_vol = np.zeros((n, ns), np.int16)
        for s in range(n):
            chunk = np.zeros(ns, np.int16)
            if len(q_out[s]) >= ns:                                # ms_bufferizer_read, all or nothing; a dry leg meters silence
                chunk, q_out[s] = q_out[s][:ns], q_out[s][ns:]
            w_vol[s] = o_vol[s].chunk(chunk)
        for c in range(nconf):
            w_mix, _ = oracle.mixer_tick(w_vol[c * mm:(c + 1) * mm])
            np.testing.assert_array_equal(g_mix[c], w_mix, err_msg=f"tick {t} conference {c}")
    assert up_worst <= 1
    assert np.sqrt(sq / cnt_s) <= 1e-4, np.sqrt(sq / cnt_s)
    assert frames == sum((lead[s] + nticks * ns) // F for s in range(n)) and two_frame_ticks > n * nticks * 0.8
    counters = np.array([aec.get(s, "counters", 4) for s in range(n)])
    adapted = np.array([aec.get(s, "scalars", 16)[8] for s in range(n)])
    assert (adapted == 1).all(), "every leg's canceller has adapted on the echo scene"
    assert (counters[:, 0] > 0).all() and counters[:, 1].sum() > 0 and counters[:, 2].sum() == 0, counters[:4]   # foreground updates on every leg, background resets seen, no state reset
    assert (counters[:, 3] == [(lead[s] + nticks * ns) // F for s in range(n)]).all()
    o_adapt = np.array([o_ec[s].get("scalars", 16)[8] for s in range(n)])
    assert (o_adapt == 1).all()
    assert f_mic.overflows() + f_ref.overflows() + f_out.overflows() == 0
    for o in (rs, aec, vol, mix, f_mic, f_ref, f_out):
        o.close()


@pytest.mark.parametrize("use_graphs,rate", [(True, 48000), (False, 48000), (False, 44100)])
def test_session_equals_the_chain_called_step_by_step(ctx, use_graphs, rate):
    """mi_session (three streams, up to three ticks in flight, hipGraph per slot) must produce exactly what the same
    C ABI objects produce when called one after the other on one stream -- it only adds plumbing.  44.1 kHz: ticks of 441
    samples against 256-sample frames (the rates msresample.c serves on a sound card's side)."""
    torch = pytest.importorskip("torch")
    nconf, mm, nticks, F = 3, 32, 25, 256
    n, ns = nconf * mm, rate // 100
    mic16 = np.stack([synth_pcm(s, 160 * nticks, rate=16000, sigma=2500.0) for s in range(n)])
    ref48 = np.stack([synth_pcm(500 + s, ns * nticks, rate=rate, sigma=3000.0) for s in range(n)])
    # ---- reference run: the individual objects, synchronous
    rs = ms.ResamplerBatch(ctx, n, 16000, rate)
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
    vol = ms.VolumeBatch(ctx, n, rate)
    p = vol.default_params()
    p.agc_enabled = 1
    vol.set_params([p] * n)
    mix = ms.MixerBatch(ctx, nconf, mm, ns)
    cap = (2 * ns + 3 * F - 1) // F * F
    f_mic, f_ref, f_out = (ms.FifoBatch(ctx, n, cap) for _ in range(3))
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    up, micf, reff, clean, tick, mixed = z(n, 488), z(n, F), z(n, F), z(n, F), z(n, ns), z(nconf, mm, ns)
    okm = z(n, dt=torch.uint8)
    want = []
    for t in range(nticks):
        d_mic = torch.from_numpy(np.ascontiguousarray(mic16[:, t * 160:(t + 1) * 160])).cuda()
        d_ref = torch.from_numpy(np.ascontiguousarray(ref48[:, t * ns:(t + 1) * ns])).cuda()
        torch.cuda.synchronize()
        rs.process(d_mic, out=up)
        f_mic.push(up, nsamples=ns)
        f_ref.push(d_ref)
        for _ in range(2):
            f_mic.pop(F, micf, ok=okm, zero_fill=False)
            f_ref.pop(F, reff, gate=okm, zero_fill=True)
            aec.process(micf, reff, out=clean, run=okm)
            f_out.push(clean, gate=okm)
        f_out.pop(ns, tick, zero_fill=True)
        vol.process(tick)
        mix.process(tick.view(nconf, mm, ns), out=mixed)
        ctx.sync()
        want.append(mixed.cpu().numpy().reshape(n, ns).copy())
    # ---- the session, pipelined: keep up to three ticks in flight
    se = ms.Session(ctx, n, members=mm, in_rate=16000, rate=rate, tail_ms=128, agc=True, use_graphs=use_graphs)
    got = []
    for t in range(nticks):
        if se.in_flight() == 3:
            got.append(se.collect().copy())
        h_mic, h_ref = se.acquire()
        h_mic[:] = mic16[:, t * 160:(t + 1) * 160]
        h_ref[:] = ref48[:, t * ns:(t + 1) * ns]
        se.submit()
    while se.in_flight():
        got.append(se.collect().copy())
    assert len(got) == nticks
    for t in range(nticks):
        np.testing.assert_array_equal(got[t], want[t], err_msg=f"tick {t}")
    assert any(g.any() for g in got)
    se.close()


def test_session_mute_and_levels(ctx):
    # Conference control plane: muting a member (MS_AUDIO_MIXER_SET_ACTIVE 0) removes it from everybody's mix from
    # the next submitted tick on; the level read-out follows MS_VOLUME_GET_LINEAR.
    nconf, mm = 2, 8
    n = nconf * mm
    se = ms.Session(ctx, n, members=mm, in_rate=48000, rate=48000, agc=False)
    rng = np.random.default_rng(3)
    L, A, O = ms.MI_MIX_LINKED, ms.MI_MIX_ACTIVE, ms.MI_MIX_OUTPUT

    def run(ticks):
        outs = []
        for _ in range(ticks):
            m, r = se.acquire()
            m[:] = tone
            r[:] = 0                      # silent far end: the canceller passes the microphone through
            se.submit()
            outs.append(se.collect().copy())
        return outs

    # one loud talker per conference (member 3), everybody else silent
    tone = np.zeros((n, 480), np.int16)
    for c in range(nconf):
        tone[c * mm + 3] = (8000 * np.sin(2 * np.pi * 440 * np.arange(480) / 48000)).astype(np.int16)
    outs = run(12)
    last = outs[-1].reshape(nconf, mm, 480)
    for c in range(nconf):
        assert np.abs(last[c, 0]).max() > 4000          # the others hear the talker
        assert np.abs(last[c, 3]).max() < 50            # the talker does not hear itself (sum - own)
    lv = se.levels().reshape(nconf, mm)
    assert (lv[:, 3] > 10 * lv[:, 0]).all()             # the meter singles the talker out
    # mute member 3 of conference 0 only
    flags = np.full(n, L | A | O, np.uint8)
    flags[3] = L | O
    se.set_controls(flags=flags)
    outs = run(6)
    last = outs[-1].reshape(nconf, mm, 480)
    assert np.abs(last[0, 0]).max() < 50                 # conference 0 went quiet
    assert np.abs(last[1, 0]).max() > 4000               # conference 1 did not
    se.close()


def test_session_members_join_and_leave_like_msaudioconference(ctx):
    """ms_audio_conference_remove_member / add_member (src/voip/audioconference.c:322-374) on a running session.
    A leg that leaves neither contributes nor hears: the remaining members' rows equal those of a twin session in which
    that leg merely went silent (their own per-leg state is untouched: the reference's filters survive the detach /
    re-attach of the conference graph), its own row reads zeros, the other conference never notices.
    A NEW leg joining the free slot starts from fresh filters: what member 0 hears of it equals what it hears in a
    brand-new session.  Then the active-speaker election (:436-452): the loudest plumbed, unmuted member above -30 dB."""
    mm, n, rate = 8, 16, 48000
    sig = [synth_pcm(50 + s, 480 * 40, rate=rate, sigma=1500.0 + 300 * s) for s in range(n)]

    def tick(se, t, silent=(), only=None):
        m, r = se.acquire()
        m[:] = 0
        r[:] = 0
        for s in range(n):
            if only is not None:
                if s == only[0]:
                    m[s] = only[1][t * 480:(t + 1) * 480]
            elif s not in silent:
                m[s] = sig[s][t * 480:(t + 1) * 480]
        se.submit()
        return se.collect().copy().reshape(2, mm, 480)

    mk = lambda: ms.Session(ctx, n, members=mm, in_rate=rate, rate=rate, agc=False)
    a, c = mk(), mk()
    for t in range(10):
        np.testing.assert_array_equal(tick(a, t), tick(c, t))
    assert a.member_count(0) == mm and a.member_count(1) == mm
    a.remove_member(3)
    assert a.member_count(0) == mm - 1 and a.member_count(1) == mm
    with pytest.raises(ms.MiError):
        a.remove_member(3)
    for t in range(10, 20):
        oa, oc = tick(a, t), tick(c, t, silent=(3,))
        np.testing.assert_array_equal(oa[1], oc[1])             # the other conference never notices
        assert not oa[0, 3].any()                               # the departed leg hears nothing
        if t >= 14:                                             # in the twin, leg 3's canceller tail has died by now
            for m in (0, 1, 2, 4, 5, 6, 7):
                np.testing.assert_array_equal(oa[0, m], oc[0, m], err_msg=f"tick {t} member {m}")
    # a NEW leg joins slot 3: fresh state for that leg, everybody else silent from here on
    a.add_member(3)
    assert a.member_count(0) == mm
    with pytest.raises(ms.MiError):
        a.add_member(3)
    fresh = mk()
    newsig = synth_pcm(999, 480 * 12, rate=rate, sigma=2500.0)
    for t in range(12):
        oa = tick(a, t, only=(3, newsig))
        of = tick(fresh, t, only=(3, newsig))
        if t >= 5:  # the others' cancellers went silent at the join: their tails are gone after a few frames
            np.testing.assert_array_equal(oa[0, 0], of[0, 0], err_msg=f"tick {t}: the joiner as member 0 hears it")
            assert np.abs(oa[0, 0]).max() > 1000
    # ---- active speaker: member 5 of conference 0 and member 2 of conference 1 shout; 5 is muted, then mm + 2 leaves
    e = mk()
    loud = np.zeros((n, 480), np.int16)
    loud[5] = (12000 * np.sin(2 * np.pi * 500 * np.arange(480) / rate)).astype(np.int16)
    loud[mm + 2] = (9000 * np.sin(2 * np.pi * 700 * np.arange(480) / rate)).astype(np.int16)
    loud[1] = (3000 * np.sin(2 * np.pi * 300 * np.arange(480) / rate)).astype(np.int16)   # audible, but not the loudest

    def shout(ticks, now):
        for _ in range(ticks):
            m, r = e.acquire()
            m[:] = loud
            r[:] = 0
            e.submit()
            e.collect()
        return e.active_speakers(now)

    win, db = shout(30, 300)
    assert list(win) == [5, mm + 2] and (db > -30).all()
    L, A, O = ms.MI_MIX_LINKED, ms.MI_MIX_ACTIVE, ms.MI_MIX_OUTPUT
    flags = np.full(n, L | A | O, np.uint8)
    flags[5] = L | O                                 # muted (MS_AUDIO_MIXER_SET_ACTIVE 0): skipped by the election (:445)
    e.set_controls(flags=flags)
    win, db = shout(5, 350)
    assert win[0] == 1 and win[1] == mm + 2          # the next loudest takes over
    e.remove_member(mm + 2)
    win, db = shout(5, 400)
    assert win[1] == -1                              # nobody else in conference 1 makes a sound
    # MS_VOLUME_GET_MAX is a one-second window fed by EVERY tick (msvolume.c:404), not by the polls: a burst that ended
    # 300 ms before the only poll still elects its speaker; a second later the window has started over and nobody is above -30 dB
    e2 = mk()

    def run(se, ticks, sig):
        for _ in range(ticks):
            m, r = se.acquire()
            m[:] = sig
            r[:] = 0
            se.submit()
            se.collect()

    run(e2, 12, loud)
    run(e2, 30, 0)
    win, db = e2.active_speakers(0)
    assert list(win) == [5, mm + 2] and (db > -30).all()
    run(e2, 110, 0)
    win, db = e2.active_speakers(0)
    assert list(win) == [-1, -1]
    e2.close()
    for se in (a, c, fresh, e):
        se.close()


def test_session_reset_streams_starts_a_leg_over(ctx):
    # A leg is replaced in its slot: after mi_session_reset_streams the slot behaves like a leg of a brand-new session
    # (resampler history, canceller, meter and FIFO phase all back to their initial state), the others are untouched.
    mm = 8
    x = synth_pcm(5, 160 * 30, rate=16000, sigma=2500.0)
    y = synth_pcm(6, 160 * 20, rate=16000, sigma=2500.0)

    def feed(se, sig, t, slot=5):
        m, r = se.acquire()
        m[:] = 0
        r[:] = 0
        m[slot] = sig[t * 160:(t + 1) * 160]
        se.submit()
        return se.collect().copy()

    a = ms.Session(ctx, mm, members=mm, agc=False)
    for t in range(13):                      # 13 ticks of the first leg: odd FIFO phase, adapted state, history
        feed(a, x, t)
    a.reset_streams(5, 1)
    got = [feed(a, y, t) for t in range(12)]
    b = ms.Session(ctx, mm, members=mm, agc=False)
    want = [feed(b, y, t) for t in range(12)]
    for t in range(12):
        np.testing.assert_array_equal(got[t], want[t], err_msg=f"tick {t}")
    assert any(g[0].any() for g in got)       # member 0 does hear the new leg
    a.close()
    b.close()


@pytest.mark.parametrize("law,plc,graphs", [(ms.MI_LAW_PCMA, False, False), (ms.MI_LAW_PCMU, False, False),
                                            (ms.MI_LAW_PCMA, True, False), (ms.MI_LAW_PCMU, True, True)])
def test_session_trunk_mode_equals_the_chain_called_step_by_step(ctx, oracle, law, plc, graphs):
    """The legs as a SIP trunk delivers them: G.711 at 8 kHz in and out, the far-end reference = what the leg was sent on
    the previous tick (delayed by ref_delay_ms), nothing but 80 + 80 bytes per leg and tick crossing PCIe.  Must equal
    MSAlawDec -> MSResample 8k->48k -> FIFO -> MSSpeexEC -> FIFO -> MSVolume -> MSAudioMixer -> MSResample 48k->8k ->
    MSAlawEnc built from the individual C ABI objects.  plc: MSGenericPLC behind the decoder, 10 % of the packets lost."""
    torch = pytest.importorskip("torch")
    nconf, mm, nticks, F, rate, delay_ms = 2, 16, 24, 256, 48000, 20
    n = nconf * mm
    pcm8 = np.stack([synth_pcm(40 + s, 80 * nticks, rate=8000, sigma=2500.0) for s in range(n)])
    codes = oracle.g711_encode(law, pcm8)
    # ---- the individual objects
    rs = ms.ResamplerBatch(ctx, n, 8000, rate)
    rs_out = ms.ResamplerBatch(ctx, n, rate, 8000)
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
    vol = ms.VolumeBatch(ctx, n, rate)
    p = vol.default_params()
    p.agc_enabled = 1
    vol.set_params([p] * n)
    mix = ms.MixerBatch(ctx, nconf, mm, 480)
    cap = (2 * 480 + 2 * F + 7) & ~7
    delay = delay_ms * rate // 1000
    f_mic, f_out = ms.FifoBatch(ctx, n, cap), ms.FifoBatch(ctx, n, cap)
    f_ref = ms.FifoBatch(ctx, n, (cap + delay + 7) & ~7)
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    pcm, up, micf, reff, clean, tick = z(n, 80), z(n, 488), z(n, F), z(n, F), z(n, F), z(n, 480)
    mixed, prev, down, enc = z(nconf, mm, 480), z(n, 480), z(n, 88), z(n, 80, dt=torch.uint8)
    okm = z(n, dt=torch.uint8)
    lost = np.random.default_rng(9).random((nticks, n)) < (0.1 if plc else 0.0)
    lost[:3] = False
    plcb = ms.PlcBatch(ctx, n, 8000, max_block=80) if plc else None
    lens80 = torch.full((n,), 80, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    f_ref.push(z(n, delay))
    want = []
    for t in range(nticks):
        d_codes = torch.from_numpy(np.ascontiguousarray(codes[:, t * 80:(t + 1) * 80])).cuda()
        ev = torch.from_numpy(np.where(lost[t], ms.MI_PLC_CONCEAL, ms.MI_PLC_RECEIVED).astype(np.uint8)).cuda()
        torch.cuda.synchronize()
        ms.g711_decode(ctx, law, d_codes, pcm)
        if plc:
            plcb.process(pcm, lens80, ev)
        rs.process(pcm, out=up)
        f_mic.push(up, nsamples=480)
        f_ref.push(prev)
        for _ in range(2):
            f_mic.pop(F, micf, ok=okm, zero_fill=False)
            f_ref.pop(F, reff, gate=okm, zero_fill=True)
            aec.process(micf, reff, out=clean, run=okm)
            f_out.push(clean, gate=okm)
        f_out.pop(480, tick, zero_fill=True)
        vol.process(tick)
        mix.process(tick.view(nconf, mm, 480), out=mixed)
        ctx.sync()
        prev = mixed.view(n, 480).clone()
        torch.cuda.synchronize()
        rs_out.process(prev, out=down)
        ms.g711_encode(ctx, law, down, enc, length=80)
        ctx.sync()
        want.append(enc.cpu().numpy().copy())
    # ---- the session
    kind = ms.MI_SESSION_PCMA if law == ms.MI_LAW_PCMA else ms.MI_SESSION_PCMU
    se = ms.Session(ctx, n, members=mm, in_rate=8000, rate=rate, tail_ms=128, agc=True, use_graphs=graphs,
                    mic_codec=kind, out_rate=8000, out_codec=kind, ref_loopback=True, ref_delay_ms=delay_ms, plc=plc)
    assert se.tick_bytes() == (80, 0, 80)
    got = []
    for t in range(nticks):
        if se.in_flight() == 3:
            got.append(se.collect().copy())
        h_mic, h_ref = se.acquire()
        assert h_ref is None and h_mic.dtype == np.uint8
        h_mic[:] = codes[:, t * 80:(t + 1) * 80]
        if plc:
            se.events()[lost[t]] = ms.MI_PLC_CONCEAL
        se.submit()
    while se.in_flight():
        got.append(se.collect().copy())
    assert len(got) == nticks
    for t in range(nticks):
        np.testing.assert_array_equal(got[t], want[t], err_msg=f"tick {t}")
    # the legs do hear each other: decoded output is not silence
    assert np.abs(oracle.g711_decode(law, got[-1])).max() > 500
    se.close()


def test_session_downsampled_pcm_output_and_reset(ctx):
    """16-bit output through the down-sampler (no codec): rows are packed to out_rate/100 samples; a reset leg starts over
    (its loop-back reference and delay line included)."""
    mm = 8
    x = synth_pcm(5, 160 * 24, rate=16000, sigma=2500.0)
    y = synth_pcm(6, 160 * 12, rate=16000, sigma=2500.0)

    def mk():
        return ms.Session(ctx, mm, members=mm, agc=False, use_graphs=False, out_rate=16000, ref_loopback=True, ref_delay_ms=10)

    def feed(se, sig, t, slot=2):
        m, r = se.acquire()
        assert r is None
        m[:] = 0
        m[slot] = sig[t * 160:(t + 1) * 160]
        se.submit()
        return se.collect().copy()

    a = mk()
    assert a.tick_bytes() == (320, 0, 320)
    for t in range(13):
        out = feed(a, x, t)
    assert out.shape == (mm, 160) and out[0].any()
    a.reset_streams(2, 1)
    b = mk()
    # the other members of `a` still carry what leg 2 said before the reset in their own loop-back state, so only the new
    # leg's own row is compared: it hears nothing but silence from the others in both sessions after a few ticks
    got = [feed(a, y, t) for t in range(12)]
    want = [feed(b, y, t) for t in range(12)]
    for t in range(4, 12):
        np.testing.assert_array_equal(got[t][0], want[t][0], err_msg=f"tick {t}")
    a.close()
    b.close()


def test_bench_rig_with_legs_out_of_phase(ctx):
    """bench.py's chain with the product's stagger (mi_aec_stagger_fifos: a lead of 32 x phase samples of silence in both
    queues, phase = a hash of the slot): in every tick about one leg in eight has ONE whole frame and the others two, nothing
    overflows (the output ring is two frames larger: a leg that misses its first pop stays one frame fuller), and after the
    start-up every leg delivers a tick per tick.  The phases come out of mi_fifo_phase_of, the same function on the host."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rig = bench.ChainRig(ms, torch, ctx, 256, stagger=True)
    try:
        lv = torch.zeros(rig.n, dtype=torch.int32, device="cuda")
        for t in range(24):
            rig.tick(t)
            rig.f_mic.levels(lv)
            ctx.sync()
            got = np.bincount(lv.cpu().numpy() // 32, minlength=8)
            want = np.bincount([(ctx.L.mi_fifo_phase_of(s, 8) - t - 1) % 8 for s in range(rig.n)], minlength=8)
            assert (lv.cpu().numpy() % 32 == 0).all() and (got == want).all(), (t, got, want)
            assert got.min() >= rig.n // 16 and got.max() <= rig.n // 4  # about one eighth of the legs in every phase
        assert rig.overflows() == 0
        rig.f_out.levels(lv)
        ctx.sync()
        out = lv.cpu().numpy()
        assert out.min() >= 0 and out.max() <= 1536 - 512
        # steady state: the output of a full cycle is not silence for any leg
        heard = np.zeros(rig.n, bool)
        for t in range(24, 32):
            rig.tick(t)
            ctx.sync()
            heard |= (rig.mixed.cpu().numpy() != 0).any(axis=1)   # every leg hears its conference
        assert heard.all()
    finally:
        rig.close()


def test_an_output_queue_that_starts_short_of_a_frame(ctx):
    """mi_fifo_reset_range_at: the canceller's launches append whole frames to their output FIFO at a tail they take to be
    frame-aligned.  A queue that starts with r samples short of a frame (what an MSVolume held when its conference graph was
    detached, handed to the plugin's fused leg: leg_chain.inl give_remainder) is emptied at head = capacity - r and given those
    r samples: its tail then lies at offset 0.  Every leg's popped stream = its r samples followed by exactly what a rig without
    them delivers, over 16+ wraps of the ring -- and the neighbours' rings are untouched (a frame written across the ring's end
    would land in the next leg's)."""
    torch = pytest.importorskip("torch")
    rate, F, ns, n, nticks = 48000, 256, 480, 6, 150
    cap, flen = 8 * F, 64 * rate // 1000
    mic = np.stack([synth_pcm(700 + s, ns * nticks, rate=rate, sigma=2500.0) for s in range(n)])
    ref = np.stack([synth_pcm(800 + s, ns * nticks, rate=rate, sigma=3000.0) for s in range(n)])
    rem = [0, 32, 224, 448, 8, 96]
    head = np.stack([synth_pcm(900 + s, ns, rate=rate, sigma=1000.0) for s in range(n)])
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")

    def run(with_rem):
        a = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen)
        fm, fr, fo = (ms.FifoBatch(ctx, n, cap) for _ in range(3))
        if with_rem:
            for s in range(n):
                if rem[s]:
                    fo.reset_range_at(s, 1, cap - rem[s])
            cnt = torch.from_numpy(np.array(rem, np.int32)).cuda()
            fo.push(torch.from_numpy(head).cuda(), nsamples=ns, count=cnt)
            _, h0, l0 = fo.snapshot()
            assert list(l0) == rem and list(h0) == [(cap - r) % cap for r in rem]
        out, ok, cnt8 = z(n, ns), z(n, dt=torch.uint8), z(n, dt=torch.uint8)
        got = [[] for _ in range(n)]
        for t in range(nticks):
            dm = torch.from_numpy(np.ascontiguousarray(mic[:, t * ns:(t + 1) * ns])).cuda()
            dr = torch.from_numpy(np.ascontiguousarray(ref[:, t * ns:(t + 1) * ns])).cuda()
            torch.cuda.synchronize()
            a.process_fifos(fm, dm, fr, dr, fo, tick_len=ns, max_frames=2, count_out=cnt8)
            fo.pop(ns, out, ok=ok, zero_fill=False)
            ctx.sync()
            o, k = out.cpu().numpy(), ok.cpu().numpy()
            for s in range(n):
                if k[s]:
                    got[s].append(o[s].copy())
        assert fm.overflows() + fr.overflows() + fo.overflows() == 0
        for o in (a, fm, fr, fo):
            o.close()
        return [np.concatenate(g) for g in got]

    plain, shifted = run(False), run(True)
    for s in range(n):
        r = rem[s]
        np.testing.assert_array_equal(shifted[s][:r], head[s][:r], err_msg=f"leg {s}: the samples it started with")
        m = min(len(shifted[s]) - r, len(plain[s]))
        assert m > 140 * ns
        np.testing.assert_array_equal(shifted[s][r:r + m], plain[s][:m], err_msg=f"leg {s} (r = {r})")


def test_queues_and_resampler_states_of_a_range_in_one_round_trip(ctx):
    """mi_fifo_export_range / import_range and mi_resampler_get_states / set_states: what a conference that is re-plumbed
    (audioconference.c:322-374) takes out of its batch and back in for all of its members at once.  export == what pops would have
    delivered, sample for sample, across a wrapped ring; import(tail_at_end) == mi_fifo_reset_range_at + push (head, level, content);
    the neighbours of the range are untouched; a resampler whose states went out and into OTHER slots continues bit for bit."""
    torch = pytest.importorskip("torch")
    n, cap = 10, 64 * 8
    f = ms.FifoBatch(ctx, n, cap)
    rng = np.random.default_rng(5)
    held = [np.zeros(0, np.int16) for _ in range(n)]
    out, ok = torch.zeros((n, 96), dtype=torch.int16, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    for t in range(40):   # pushes of ragged counts and pops: the heads wrap several times
        cnt = rng.integers(0, 121, n).astype(np.int32) // 8 * 8
        x = rng.integers(-30000, 30000, (n, 120)).astype(np.int16)
        f.push(torch.from_numpy(x).cuda(), nsamples=120, count=torch.from_numpy(cnt).cuda())
        for s in range(n):
            held[s] = np.concatenate([held[s], x[s, :cnt[s]]])
        f.pop(96, out, ok=ok, zero_fill=False)
        ctx.sync()
        o, k = out.cpu().numpy(), ok.cpu().numpy()
        for s in range(n):
            if k[s]:
                assert np.array_equal(o[s], held[s][:96])
                held[s] = held[s][96:]
    assert f.overflows() == 0
    rings0, head0, level0 = f.snapshot()
    got = f.export_range(3, 5)
    for k in range(5):
        assert np.array_equal(got[k], held[3 + k]), k
    # ... and back, the queues ending on the ring's end: the layout mi_fifo_reset_range_at + push leaves
    qs = [q[:len(q) // 8 * 8] for q in got]
    f.import_range(3, qs, tail_at_end=True)
    rings1, head1, level1 = f.snapshot()
    g = ms.FifoBatch(ctx, n, cap)
    for k, q in enumerate(qs):
        if len(q):
            g.reset_range_at(3 + k, 1, cap - len(q))
    xs = np.zeros((n, cap), np.int16)
    cnt = np.zeros(n, np.int32)
    for k, q in enumerate(qs):
        xs[3 + k, :len(q)], cnt[3 + k] = q, len(q)
    g.push(torch.from_numpy(xs).cuda(), nsamples=cap, count=torch.from_numpy(cnt).cuda())
    ctx.sync()
    rings2, head2, level2 = g.snapshot()
    for s in range(3, 8):
        assert (head1[s], level1[s]) == (head2[s], level2[s]) and np.array_equal(f.export_range(s, 1)[0], qs[s - 3])
    for s in (0, 1, 2, 8, 9):   # the neighbours: as they were
        assert (head1[s], level1[s]) == (head0[s], level0[s]) and np.array_equal(rings1[s], rings0[s])
    # the resampler's states: out of slots 2..5, into slots 6..9 of a second batch -- the streams continue bit for bit
    a, b = ms.ResamplerBatch(ctx, 8, 16000, 48000), ms.ResamplerBatch(ctx, 12, 16000, 48000)
    x = rng.normal(0, 4000, (8, 3 * 160)).round().clip(-32767, 32767).astype(np.int16)
    for t in range(2):
        a.process(torch.from_numpy(np.ascontiguousarray(x[:, t * 160:(t + 1) * 160])).cuda())
    ctx.sync()
    st = a.get_states(2, 4)
    assert st == b"".join(a.get_state(s) for s in range(2, 6))
    b.set_states(6, 4, st)
    xa = torch.from_numpy(np.ascontiguousarray(x[:, 320:480])).cuda()
    xb = torch.zeros((12, 160), dtype=torch.int16, device="cuda")
    xb[6:10] = xa[2:6]
    oa, _ = a.process(xa)
    ob, _ = b.process(xb)
    ctx.sync()
    assert torch.equal(oa[2:6, :480], ob[6:10, :480])
    for o in (f, g, a, b):
        o.close()


def test_the_fifo_entry_serves_the_legs_sorted_by_their_frames(ctx):
    """mi_aec_process_fifos keeps eight lists of legs (one per class b % 8 = the XCD a workgroup lands on) that every launch
    rebuilds for the next one: each leg enters itself into class (own class + own position) % 8, from the front if it will
    have two frames, from the back otherwise.  After any number of ticks the lists hold every leg exactly once, are sorted by
    what the legs really have in the next tick, stay even in size, and spread the light legs evenly over the classes -- here
    with phase = slot % 8, the arrangement that used to keep a tick's light legs on ONE XCD -- and the results do not depend
    on the order (same outputs as a batch whose legs are arranged the other way round)."""
    torch = pytest.importorskip("torch")
    rate, F, n, ns = 48000, 256, 203, 480
    flen = 32 * rate // 1000
    mic = np.stack([synth_pcm(400 + s, ns * 30, rate=rate, sigma=2500.0) for s in range(n)])
    ref = np.stack([synth_pcm(700 + s, ns * 30, rate=rate, sigma=3000.0) for s in range(n)])
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")

    def rig(rev):
        a = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen)
        fm, fr, fo = (ms.FifoBatch(ctx, n, 1536) for _ in range(3))
        return a, fm, fr, fo, (np.arange(n)[::-1].copy() if rev else np.arange(n))

    outs = []
    for rev in (False, True):
        a, fm, fr, fo, perm = rig(rev)
        # the same lead for the same SIGNAL in both arrangements; phase = (forward) slot % 8
        lead = np.array([32 * (int(perm[s]) % 8) for s in range(n)], np.int32)
        zeros, d_lead = z(n, 256), torch.from_numpy(lead).cuda()
        torch.cuda.synchronize()
        fm.push(zeros, nsamples=224, count=d_lead)
        fr.push(zeros, nsamples=224, count=d_lead)
        cnt, tick, lv = z(n, dt=torch.uint8), z(n, ns), z(n, dt=torch.int32)
        got = []
        for t in range(30):
            dm = torch.from_numpy(np.ascontiguousarray(mic[perm, t * ns:(t + 1) * ns])).cuda()
            dr = torch.from_numpy(np.ascontiguousarray(ref[perm, t * ns:(t + 1) * ns])).cuda()
            torch.cuda.synchronize()
            a.process_fifos(fm, dm, fr, dr, fo, tick_len=ns, max_frames=2, count_out=cnt)
            fo.pop(ns, tick, zero_fill=True)
            fm.levels(lv)
            ctx.sync()
            got.append(tick.cpu().numpy()[np.argsort(perm)].copy())
            order = a.get(0, "order", n + 8).astype(int)   # class after class, each closed by -1
            nxt = (lv.cpu().numpy() + ns) // F              # frames every leg will have in the next tick
            classes, cur = [], []
            for v in order:
                if v < 0:
                    classes.append(cur)
                    cur = []
                else:
                    cur.append(int(v))
            assert len(classes) == 8 and sorted(x for cl in classes for x in cl) == list(range(n)), t   # every leg exactly once
            for c, lst in enumerate(classes):
                assert abs(len(lst) - n / 8) <= 8, (t, c, len(lst))               # the classes stay even
                fr_next = nxt[lst]
                assert (np.diff((fr_next >= 2).astype(int)) <= 0).all(), (t, c, fr_next)   # two-frame legs first
            if t >= 2:  # a class's legs are dealt out over all classes every tick: the light legs end up spread evenly
                light = [int((nxt[lst] < 2).sum()) for lst in classes]
                assert max(light) - min(light) <= max(10, n // 16), (t, light)   # (with phase = slot % 8 and no mixing: n / 8 against 0)
        outs.append(np.stack(got))
        assert fm.overflows() + fr.overflows() + fo.overflows() == 0
        for o in (a, fm, fr, fo):
            o.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    assert outs[0].any()


@pytest.mark.parametrize("n", [1, 4, 9, 12, 20])
def test_small_batches_keep_their_leg_lists_over_thousands_of_ticks(ctx, n):
    """Batches whose eight leg lists are not all populated -- fewer than 8 legs, or classes the dealing empties for a tick
    (n = 9, 12, 20 with the product's stagger) -- must turn their lists over like any other: round 3's hand-over waited for
    every class to complete and an empty class never did (the lists froze on tick 0 and the placement counts grew past them
    after ~190 ticks).  2 200 ticks: every probe finds each leg exactly once in the lists, every leg was served once per
    tick (its frame counter is what its queue level implies), and the ticks' outputs equal those of the same legs inside a
    LARGER batch (whose lists are populated)."""
    torch = pytest.importorskip("torch")
    rate, F, ns, nticks = 48000, 256, 480, 2200
    flen = 16 * rate // 1000
    big = 64
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    gen = torch.Generator(device="cpu").manual_seed(1234 + n)
    block = (torch.randn((8, 2, big, ns), generator=gen) * 2500).round().clamp(-32767, 32767).to(torch.int16).cuda()   # 8 ticks, cycled
    rigs = []
    for m in (n, big):
        a = ms.AecBatch(ctx, m, rate, frame_size=F, filter_length=flen)
        fm, fr, fo = (ms.FifoBatch(ctx, m, 1792) for _ in range(3))
        a.stagger_fifos(fm, fr, ns)
        rigs.append((a, fm, fr, fo, z(m, ns), z(m, dt=torch.int32)))
    # leg s of the small batch must see the lead leg s of the big one has: the phases depend on the slot only
    torch.cuda.synchronize()
    for t in range(nticks):
        outs = []
        for (a, fm, fr, fo, tick, lv), m in zip(rigs, (n, big)):
            dm, dr = block[t % 8, 0, :m], block[t % 8, 1, :m]
            a.process_fifos(fm, dm, fr, dr, fo, tick_len=ns, max_frames=2)
            fo.pop(ns, tick, zero_fill=True)
            outs.append(tick)
        if t % 97 == 0 or t >= nticks - 3:
            ctx.sync()
            np.testing.assert_array_equal(outs[0].cpu().numpy(), outs[1].cpu().numpy()[:n], err_msg=f"tick {t}")
            a = rigs[0][0]
            order = a.get(0, "order", n + 8).astype(int)
            assert sorted(int(v) for v in order if v >= 0) == list(range(n)) and (order < 0).sum() == 8, (t, order)
    a, fm, fr, fo, tick, lv = rigs[0]
    fm.levels(lv)
    ctx.sync()
    lead = [32 * ctx.L.mi_fifo_phase_of(s, 8) for s in range(n)]
    for s in range(n):
        assert int(a.get(s, "counters", 4)[3]) == (lead[s] + nticks * ns - int(lv[s])) // F, s
    assert fm.overflows() + fr.overflows() + fo.overflows() == 0
    for rig_ in rigs:
        for o in rig_[:4]:
            o.close()


def test_canceller_state_copied_on_the_device_continues_bit_for_bit(ctx):
    """mi_aec_copy_state: a batch seeded from another one's converged legs continues exactly as they do."""
    rate, F, n = 16000, 128, 6
    flen = 64 * rate // 1000
    x = np.stack([synth_pcm(40 + s, F * 60, rate=rate, sigma=3000.0) for s in range(n)])
    m = np.stack([(0.4 * x[s] + synth_pcm(90 + s, F * 60, rate=rate, sigma=100.0)).astype(np.int16) for s in range(n)])
    a = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen)
    b = ms.AecBatch(ctx, 2 * n + 1, rate, frame_size=F, filter_length=flen)
    for f in range(40):
        a.process(np.ascontiguousarray(m[:, f * F:(f + 1) * F]), np.ascontiguousarray(x[:, f * F:(f + 1) * F]), flags=ms.MI_AEC_POSTFILTER)
    b.copy_state_from(a, 0, 0, n)
    b.copy_state_from(a, 0, n + 1, n)       # leg n of b stays as created
    with pytest.raises(ms.MiError):
        b.copy_state_from(a, 0, n + 2, n)   # does not fit
    for f in range(40, 60):
        sl = slice(f * F, (f + 1) * F)
        oa = a.process(np.ascontiguousarray(m[:, sl]), np.ascontiguousarray(x[:, sl]), flags=ms.MI_AEC_POSTFILTER)
        mb = np.concatenate([m[:, sl], m[:1, sl], m[:, sl]])
        xb = np.concatenate([x[:, sl], x[:1, sl], x[:, sl]])
        ob = b.process(np.ascontiguousarray(mb), np.ascontiguousarray(xb), flags=ms.MI_AEC_POSTFILTER)
        np.testing.assert_array_equal(ob[:n], oa)
        np.testing.assert_array_equal(ob[n + 1:], oa)
    assert a.get(0, "counters", 4)[3] == 60 and b.get(0, "counters", 4)[3] == 60 and b.get(n, "counters", 4)[3] == 20
    a.close()
    b.close()


def test_session_with_the_products_stagger_equals_the_chain_given_the_same_leads(ctx):
    """mi_session_config.stagger (the C default): every leg starts with mi_aec_stagger_fifos' lead; a leg that is reset gets
    its lead again.  Output == the chain of individual objects whose queues were given the same leads."""
    torch = pytest.importorskip("torch")
    nconf, mm, nticks, F, rate, ns = 2, 8, 20, 256, 48000, 480
    n = nconf * mm
    mic16 = np.stack([synth_pcm(s, 160 * nticks, rate=16000, sigma=2500.0) for s in range(n)])
    ref48 = np.stack([synth_pcm(500 + s, ns * nticks, rate=rate, sigma=3000.0) for s in range(n)])
    rs = ms.ResamplerBatch(ctx, n, 16000, rate)
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
    vol = ms.VolumeBatch(ctx, n, rate)
    p = vol.default_params()
    p.agc_enabled = 1
    vol.set_params([p] * n)
    mix = ms.MixerBatch(ctx, nconf, mm, ns)
    f_mic, f_ref, f_out = (ms.FifoBatch(ctx, n, 1536) for _ in range(3))
    assert aec.stagger_info(ns) == (32, 8) and ms.AecBatch(ctx, 1, 16000).stagger_info(160) == (32, 4)
    aec.stagger_fifos(f_mic, f_ref, ns)
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    up, tick, mixed = z(n, 488), z(n, ns), z(nconf, mm, ns)
    want = []
    for t in range(nticks):
        d_mic = torch.from_numpy(np.ascontiguousarray(mic16[:, t * 160:(t + 1) * 160])).cuda()
        d_ref = torch.from_numpy(np.ascontiguousarray(ref48[:, t * ns:(t + 1) * ns])).cuda()
        torch.cuda.synchronize()
        if t == 9:  # leg 3 is replaced in its slot
            ctx.sync()
            rs.reset(3, 1)
            aec.reset(3, 1)
            fresh = ms.VolumeBatch(ctx, 1, rate).get_state()[0]
            vol.set_state([fresh], first=3)
            vol.reset_max(3, 1)
            for f in (f_mic, f_ref, f_out):
                f.reset_range(3, 1)
            aec.stagger_fifos(f_mic, f_ref, ns, first=3, count=1)
        rs.process(d_mic, out=up)
        aec.process_fifos(f_mic, up, f_ref, d_ref, f_out, tick_len=ns, max_frames=2)
        vol.process_fifo(f_out, tick)
        mix.process(tick.view(nconf, mm, ns), out=mixed)
        ctx.sync()
        want.append(mixed.cpu().numpy().reshape(n, ns).copy())
    se = ms.Session(ctx, n, members=mm, in_rate=16000, rate=rate, tail_ms=128, agc=True, use_graphs=False, stagger=True)
    for t in range(nticks):
        if t == 9:
            se.reset_streams(3, 1)
        h_mic, h_ref = se.acquire()
        h_mic[:] = mic16[:, t * 160:(t + 1) * 160]
        h_ref[:] = ref48[:, t * ns:(t + 1) * ns]
        se.submit()
        np.testing.assert_array_equal(se.collect(), want[t], err_msg=f"tick {t}")
    se.close()


@pytest.mark.parametrize("in_rate,rate,F", [(16000, 48000, 256), (8000, 48000, 256), (8000, 16000, 128)])
def test_canceller_launch_with_the_resampler_folded_in_equals_the_two_launches(ctx, in_rate, rate, F):
    """mi_aec_process_fifos_resampled (MSResample + MSSpeexEC of a leg in ONE launch: the wavefront up-samples the block
    with the resampler's own tile FIR, history and table) == mi_resampler_process followed by mi_aec_process_fifos: what
    the output FIFO delivers, the FIFO levels, and the resampler's state afterwards (it carries on bit for bit)."""
    torch = pytest.importorskip("torch")
    n, nticks, nin, ns = 37, 40, in_rate // 100, rate // 100
    flen = 64 * rate // 1000
    mic = np.stack([synth_pcm(900 + s, nin * (nticks + 2), rate=in_rate, sigma=2500.0) for s in range(n)])
    ref = np.stack([synth_pcm(950 + s, ns * nticks, rate=rate, sigma=3000.0) for s in range(n)])
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")

    def rig():
        return (ms.ResamplerBatch(ctx, n, in_rate, rate), ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=flen),
                *(ms.FifoBatch(ctx, n, 6 * F) for _ in range(3)))

    (rs1, a1, fm1, fr1, fo1), (rs2, a2, fm2, fr2, fo2) = rig(), rig()
    for a, fm, fr in ((a1, fm1, fr1), (a2, fm2, fr2)):
        a.stagger_fifos(fm, fr, ns)
    up = z(n, (ns + 8 + 7) & ~7)
    t1, t2, lv1, lv2 = z(n, ns), z(n, ns), z(n, dt=torch.int32), z(n, dt=torch.int32)
    for t in range(nticks):
        dm = torch.from_numpy(np.ascontiguousarray(mic[:, t * nin:(t + 1) * nin])).cuda()
        dr = torch.from_numpy(np.ascontiguousarray(ref[:, t * ns:(t + 1) * ns])).cuda()
        torch.cuda.synchronize()
        rs1.process(dm, out=up)
        a1.process_fifos(fm1, up, fr1, dr, fo1, tick_len=ns, max_frames=2)
        a2.process_fifos_resampled(rs2, dm, fm2, fr2, dr, fo2, max_frames=2)
        fo1.pop(ns, t1, zero_fill=True)
        fo2.pop(ns, t2, zero_fill=True)
        fm1.levels(lv1)
        fm2.levels(lv2)
        ctx.sync()
        np.testing.assert_array_equal(t1.cpu().numpy(), t2.cpu().numpy(), err_msg=f"tick {t}")
        np.testing.assert_array_equal(lv1.cpu().numpy(), lv2.cpu().numpy())
    assert t1.cpu().numpy().any()
    # the folded resampler's state went along: both carry on alike
    for t in range(nticks, nticks + 2):
        dm = torch.from_numpy(np.ascontiguousarray(mic[:, t * nin:(t + 1) * nin])).cuda()
        torch.cuda.synchronize()
        o1, _ = rs1.process(dm)
        o2, _ = rs2.process(dm)
        ctx.sync()
        np.testing.assert_array_equal(o1.cpu().numpy()[:, :ns], o2.cpu().numpy()[:, :ns])
    assert fm2.overflows() + fr2.overflows() + fo2.overflows() == 0
    # a ratio the launch cannot up-sample itself is refused, not approximated
    rs3 = ms.ResamplerBatch(ctx, n, 16000, 44100)
    with pytest.raises(ms.MiError):
        a2.process_fifos_resampled(rs3, z(n, 160), fm2, fr2, z(n, 448), fo2)
    for o in (rs1, a1, fm1, fr1, fo1, rs2, a2, fm2, fr2, fo2, rs3):
        o.close()


def test_session_elects_what_the_oracle_conference_elects(ctx, oracle):
    """mi_session's own conference glue (mi_session_add_member / remove_member / set_controls / active_speakers) against the
    ORACLE's MSAudioConference (oracle/conference.c: pins, list order, the election of audioconference.c:419-464) fed by the chain
    of oracle objects on the same audio (Echo + Preproc -> Volume with AGC -> its 1 s OrtpExtremum): two conferences of 8, talkers
    whose loudness is scripted, the loudest muted and un-muted, one member leaving and a NEW one taking its slot.  Every poll
    (50 ms apart): the same winner -- polls in which the oracle's two loudest are within 0.5 dB of each other or of the -30 dB
    threshold left out -- and the winner's maximum within 0.2 dB."""
    mm, n, rate, ns, F, nticks = 8, 16, 48000, 480, 256, 330
    rng = np.random.default_rng(21)
    t = np.arange(nticks * ns)
    loud = np.full((n, 5), 150.0)                       # sigma per leg and 66-tick period
    loud[2] = [3000, 3000, 3000, 3000, 3000]
    loud[5] = [800, 800, 9000, 800, 800]
    loud[6] = [1500, 1500, 1500, 1500, 6000]
    loud[mm + 1] = [2500, 2500, 2500, 2500, 2500]
    loud[mm + 4] = [700, 700, 700, 5000, 5000]
    far = np.stack([(rng.normal(0, 1200, nticks * ns) + 800 * np.sin(2 * np.pi * (300 + 40 * s) * t / rate)) for s in range(n)])
    echo = 0.3 * np.concatenate([np.zeros((n, 2 * ns)), far[:, :-2 * ns]], axis=1)
    env = np.repeat(loud, 66 * ns, axis=1)[:, :nticks * ns]
    talk = env * (0.6 * rng.normal(0, 1, (n, nticks * ns)) + np.sin(2 * np.pi * 170 * t / rate))
    mic = (echo + talk).round().clip(-32767, 32767).astype(np.int16)
    far = far.round().clip(-32767, 32767).astype(np.int16)
    script = {70: ("mute", 2, True), 150: ("leave", mm + 1), 200: ("mute", 2, False), 240: ("join", mm + 1)}
    joiner = (2200 * (0.6 * rng.normal(0, 1, nticks * ns) + np.sin(2 * np.pi * 210 * t / rate))).round().clip(-32767, 32767).astype(np.int16)

    se = ms.Session(ctx, n, members=mm, in_rate=rate, rate=rate, tail_ms=64, agc=True)
    L, A, O = ms.MI_MIX_LINKED, ms.MI_MIX_ACTIVE, ms.MI_MIX_OUTPUT
    flags = np.full(n, L | A | O, np.uint8)
    flen = 64 * rate // 1000

    class OLeg:
        def __init__(self):
            self.ec = oracle.Echo(F, flen, rate)
            self.pp = oracle.Preproc(F, rate, self.ec)
            self.vol = oracle.Volume(rate)
            self.vol.v.agc_enabled = 1
            self.max = oracle.Extremum(1000)
            self.qm, self.qr, self.qv = (np.zeros(0, np.int16) for _ in range(3))

        def tick(self, now, m, r):
            self.qm, self.qr = np.concatenate([self.qm, m]), np.concatenate([self.qr, r])
            while len(self.qm) >= F:
                self.qv = np.concatenate([self.qv, self.pp.run(self.ec.cancel(self.qm[:F], self.qr[:F]))])
                self.qm, self.qr = self.qm[F:], self.qr[F:]
            if len(self.qv) >= ns:          # the session's volume + mix launch takes one chunk per tick (a dry leg is metered on silence)
                ch, self.qv = self.qv[:ns], self.qv[ns:]
            else:
                ch = np.zeros(ns, np.int16)
            self.vol.chunk(ch)
            self.max.record_max(now, self.vol.v.energy)

    legs = [OLeg() for _ in range(n)]
    books = [oracle.Conference(), oracle.Conference()]
    for c in range(2):
        assert [books[c].add_member() for _ in range(mm)] == list(range(mm))
    present = np.ones(n, bool)
    checked = skipped = 0
    for k in range(nticks):
        ev = script.get(k)
        if ev:
            s = ev[1]
            if ev[0] == "mute":
                flags[s] = (L | O) if ev[2] else (L | A | O)
                se.set_controls(flags=flags)
                books[s // mm].mute_member(s % mm, ev[2])
            elif ev[0] == "leave":
                se.remove_member(s)
                books[s // mm].remove_member(s % mm)
                present[s] = False
                flags[s] = 0
            else:
                se.add_member(s)
                assert books[s // mm].add_member() == s % mm            # the lowest free pin is the slot that was given up
                legs[s] = OLeg()                                         # a NEW endpoint: fresh filters
                present[s] = True
                flags[s] = L | A | O
                mic[s, k * ns:] = joiner[k * ns:]
        m, r = se.acquire()
        m[:] = mic[:, k * ns:(k + 1) * ns]
        r[:] = far[:, k * ns:(k + 1) * ns]
        m[~present] = 0
        se.submit()
        se.collect()
        for s in range(n):
            if present[s]:
                legs[s].tick(10 * k, mic[s, k * ns:(k + 1) * ns], far[s, k * ns:(k + 1) * ns])
        if k % 5 == 4:
            win, db = se.active_speakers(10 * k)
            for c in range(2):
                want = {p: oracle.linear_to_dbm0(legs[c * mm + p].max.current) for p in books[c].order}
                _, wpin, wdb = books[c].process_events(want)
                top = sorted(want.values(), reverse=True)
                if top[0] - top[1] < 0.5 or any(abs(v + 30.0) < 0.5 for v in top[:2]):
                    skipped += 1
                    continue
                checked += 1
                assert (win[c] - c * mm if win[c] >= 0 else -1) == wpin, (k, c, win[c], wpin, want)
                if wpin >= 0:
                    assert abs(db[c] - wdb) < 0.2, (k, c, db[c], wdb)
    assert checked > 0.8 * 2 * (nticks // 5), (checked, skipped)
    assert se.member_count(0) == books[0].size == mm and se.member_count(1) == books[1].size == mm
    se.close()
