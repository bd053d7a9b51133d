"""GPU parity: mi_equalizer_* vs the oracle's restatement of equalizer.c +
dsptools.c ms_fir_mem16.  Taps (host design) and int16 outputs (device FIR, same
float32 operation order, unfused) must be BIT-EXACT."""
import numpy as np
import pytest

import mediastreamer2_amd as ms
from conftest import synth_pcm

pytestmark = pytest.mark.gpu

GAINS = [(1000, 2.0, 500), (300, 0.3, 100), (6000, 4.0, 2000), (50, 1.5, 20), (3400, 0.1, 800)]


@pytest.mark.parametrize("rate,n", [(48000, 480), (16000, 160), (8000, 80), (44100, 441)])
def test_equalizer_taps_and_output_bit_exact(ctx, oracle, rate, n):
    ns = len(GAINS) + 2
    eq = ms.EqualizerBatch(ctx, ns, rate)
    orcs = [oracle.Equalizer(rate) for _ in range(ns)]
    for i, (f, g, w) in enumerate(GAINS):
        eq.set_gain(i, f, g, w)
        orcs[i].set_gain(f, g, w)
    # stream len(GAINS): two cumulative gains (A16); last stream: flat
    for (f, g, w) in GAINS[:2]:
        eq.set_gain(len(GAINS), f, g, w)
        orcs[len(GAINS)].set_gain(f, g, w)
    for i in range(ns):
        np.testing.assert_array_equal(eq.taps(i).view(np.uint32), orcs[i].taps().view(np.uint32),
                                      err_msg=f"taps stream {i}")
        spec = orcs[i].spectrum()
        dump = np.concatenate([[spec[0]], spec[1::2] * eq.fir_len])
        np.testing.assert_array_equal(eq.dump(i), dump.astype(np.float32)[:eq.fir_len // 2])
    for t in range(8):
        x = np.stack([synth_pcm(i, n, sigma=2500.0, rate=rate, t0=t * n) for i in range(ns)])
        got = eq.process(np.ascontiguousarray(x.copy()))
        for i in range(ns):
            np.testing.assert_array_equal(got[i], orcs[i].run(x[i]), err_msg=f"tick {t} stream {i}")
    eq.close()


def test_equalizer_stream_moves_between_batches_sample_for_sample(ctx, oracle):
    """mi_equalizer_get_history / set_history: a stream that leaves one batch for another in mid-stream (the plugin's mic_equalizer joining
    or leaving a fused leg) carries its FIR's memory -- ms_fir_mem16's `mem` lives as long as the filter (equalizer.c:256-268) -- and
    continues bit for bit as the oracle's single filter does; a slot whose memory was cleared starts like a new filter."""
    rate, n = 48000, 480
    a, b = ms.EqualizerBatch(ctx, 3, rate), ms.EqualizerBatch(ctx, 5, rate)
    orc, fresh = oracle.Equalizer(rate), oracle.Equalizer(rate)
    for (f, g, w) in GAINS[:3]:
        a.set_gain(1, f, g, w)
        b.set_gain(4, f, g, w)
        b.set_gain(2, f, g, w)
        orc.set_gain(f, g, w)
        fresh.set_gain(f, g, w)
    sig = synth_pcm(9, n * 12, sigma=3000.0, rate=rate)
    for t in range(6):
        x = np.zeros((3, n), np.int16)
        x[1] = sig[t * n:(t + 1) * n]
        np.testing.assert_array_equal(a.process(np.ascontiguousarray(x))[1], orc.run(sig[t * n:(t + 1) * n]))
    hist = a.history(1)
    assert hist.any()
    b.set_history(4, hist)
    noise = np.stack([synth_pcm(20 + i, n, sigma=3000.0, rate=rate) for i in range(5)])
    b.process(np.ascontiguousarray(noise[:, :n].copy()), )   # (slot 2 has run on something else: its memory is not clean)
    b.set_history(4, hist)
    b.set_history(2, None)
    for t in range(6, 12):
        x = noise.copy()
        x[4] = sig[t * n:(t + 1) * n]
        x[2] = sig[(t - 6) * n:(t - 5) * n]
        got = b.process(np.ascontiguousarray(x))
        np.testing.assert_array_equal(got[4], orc.run(sig[t * n:(t + 1) * n]), err_msg=f"moved stream, tick {t}")
        np.testing.assert_array_equal(got[2], fresh.run(sig[(t - 6) * n:(t - 5) * n]), err_msg=f"cleared slot, tick {t}")
    a.close()
    b.close()


def test_equalizer_flat_is_delay_and_inactive_passthrough(ctx, oracle):
    eq = ms.EqualizerBatch(ctx, 3, 48000)
    eq.set_active(1, 0)
    o = oracle.Equalizer(48000)
    x = np.stack([synth_pcm(3, 480) for _ in range(3)])
    got = eq.process(np.ascontiguousarray(x.copy()))
    np.testing.assert_array_equal(got[1], x[1])          # inactive: untouched
    np.testing.assert_array_equal(got[0], o.run(x[0]))
    np.testing.assert_array_equal(got[2], got[0])
    # re-activated stream starts from an untouched delay line (its filter never ran)
    eq.set_active(1, 1)
    got2 = eq.process(np.ascontiguousarray(x.copy()))
    np.testing.assert_array_equal(got2[1], got[0])
    eq.close()


def test_equalizer_saturation_and_ragged(ctx, oracle):
    eq = ms.EqualizerBatch(ctx, 2, 16000)
    o = [oracle.Equalizer(16000) for _ in range(2)]
    for e in (0, 1):
        eq.set_gain(e, 1000, 8.0, 4000)
        o[e].set_gain(1000, 8.0, 4000)
    for blk in (160, 7, 1, 320, 33):
        t = np.arange(blk)
        x = np.stack([(32000 * np.sin(2 * np.pi * 1000 * t / 16000)).astype(np.int16),
                      np.where(t % 16 < 8, 32767, -32768).astype(np.int16)])
        got = eq.process(np.ascontiguousarray(x.copy()))
        for e in (0, 1):
            np.testing.assert_array_equal(got[e], o[e].run(x[e]), err_msg=f"blk {blk} stream {e}")
    eq.close()


def test_equalizer_full_size_4096_streams(ctx, oracle):
    torch = pytest.importorskip("torch")
    n, ns = 4096, 480
    eq = ms.EqualizerBatch(ctx, n, 48000)
    o = oracle.Equalizer(48000)
    o.set_gain(2000, 3.0, 900)
    taps = o.taps()
    for s in range(n):
        eq.set_taps(s, taps)
    base = np.stack([synth_pcm(s % 16, ns * 2, sigma=3000.0) for s in range(n)])
    orc16 = []
    for s in range(16):
        oo = oracle.Equalizer(48000)
        oo.set_gain(2000, 3.0, 900)
        orc16.append(oo)
    for t in range(2):
        d = torch.from_numpy(np.ascontiguousarray(base[:, t * ns:(t + 1) * ns])).cuda()
        eq.process(d)
        ctx.sync()
        torch.cuda.synchronize()
        out = d.cpu().numpy()
        for s in range(16):
            ref = orc16[s].run(base[s, t * ns:(t + 1) * ns])
            np.testing.assert_array_equal(out[s], ref)
            np.testing.assert_array_equal(out[s + 4080], ref)
        grp = out.reshape(256, 16, ns)
        assert (grp == grp[:1]).all()
    eq.close()
