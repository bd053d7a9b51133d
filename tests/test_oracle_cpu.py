"""CPU tests of the oracle (the checker itself): analytic known answers, independent
numpy restatements, the committed golden fixtures, and -- when the reference tree is
present (this container, never the GPU box) -- its own WAV fixtures."""
import os
import struct

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ------------------------------------------------------------------ mixer
def np_mixer(x, has, gain, act, oen, conf_mode):
    """independent numpy statement of audiomixer.c:78-130,:301-344"""
    x = x.astype(np.int64)
    sat = lambda v: np.clip(v, -32767, 32767)
    c = np.where(has[:, None] != 0, x, 0)
    g = gain.astype(np.float32)[:, None]
    gained = sat(np.trunc((g * c.astype(np.float32)).astype(np.float64)).astype(np.int64))
    c = np.where((act[:, None] != 0) & (gain[:, None] != 1.0), gained, c)
    s = (c * (act[:, None] != 0)).sum(axis=0)
    if not conf_mode:
        return sat(s).astype(np.int16), s
    out = np.where(act[:, None] != 0, sat(s[None] - c), sat(s[None]))
    return out.astype(np.int16), s


def test_mixer_oracle_vs_numpy_and_golden(oracle):
    g = gold("mixer.npz")
    for c in range(g["x"].shape[0]):
        for mode, key in ((1, "out"), (0, "flat")):
            o, s = oracle.mixer_tick(g["x"][c], g["has"][c], g["gain"][c], g["act"][c], g["oen"][c], mode)
            ref, sref = np_mixer(g["x"][c], g["has"][c], g["gain"][c], g["act"][c], g["oen"][c], mode)
            np.testing.assert_array_equal(s, sref)
            if mode:
                m = g["oen"][c] != 0
                np.testing.assert_array_equal(o[m], ref[m])
                np.testing.assert_array_equal(o[m], g[key][c][m])
            else:
                np.testing.assert_array_equal(o, ref)
                np.testing.assert_array_equal(o, g[key][c])


def test_mixer_symmetric_saturation(oracle):
    x = np.array([[-32768] * 4, [-32768] * 4, [32767] * 4], np.int16)
    o, s = oracle.mixer_tick(x)
    assert o.min() == -32767 and o.max() <= 32767
    assert (s == -32768 - 32768 + 32767).all()


# ----------------------------------------------------------------- volume
def np_volume_plain(sig, n, static_gain):
    """independent float32 statement of update_energy + apply_gain (no AGC/gate)"""
    f32 = np.float32
    energy, gain, out, en_l = f32(0), f32(static_gain), [], []
    max_e = f32(32768 * f32(0.7))
    for t in range(len(sig) // n):
        x = sig[t * n:(t + 1) * n].astype(np.int32)
        acc = f32(0)
        for v in x:
            acc = f32(acc + f32(int(v) * int(v)))
        en = f32((np.sqrt(np.float64(f32(acc / f32(n)))) + 1) / np.float64(max_e))
        energy = f32(f32(en * f32(0.2)) + f32(energy * f32(f32(1.0) - f32(0.2))))
        en_l.append(energy)
        intgain = int(f32(gain * f32(4096)))
        if gain != 1:
            q = np.trunc(x.astype(np.int64) * intgain / 4096.0).astype(np.int64)
            out.append(np.clip(q, -32767, 32767).astype(np.int16))
        else:
            out.append(x.astype(np.int16))
    return np.concatenate(out), np.array(en_l, f32)


@pytest.mark.parametrize("g", [1.0, 0.5, 1.7])
def test_volume_oracle_vs_numpy(oracle, pcm, g):
    sig = pcm(5, 480 * 6, sigma=6000.0)
    v = oracle.Volume(48000)
    v.v.static_gain = v.v.gain = v.v.target_gain = g
    got = np.concatenate([v.chunk(sig[t * 480:(t + 1) * 480]) for t in range(6)])
    ref, en = np_volume_plain(sig, 480, g)
    np.testing.assert_array_equal(got, ref)
    assert np.float32(v.v.energy).view(np.uint32) == en[-1].view(np.uint32)


def test_volume_golden(oracle):
    g = gold("volume.npz")
    v = oracle.Volume(16000)
    v.v.agc_enabled = 1
    v.v.noise_gate_enabled = 1
    v.v.gain = v.v.target_gain = v.v.ng_floorgain
    out = np.concatenate([v.chunk(g["x"][t * 160:(t + 1) * 160]) for t in range(40)])
    np.testing.assert_array_equal(out, g["out"])
    assert np.float32(v.v.energy) == g["energy"][-1] and np.float32(v.v.gain) == g["gain"][-1]
    assert g["gain"].min() < 0.2 and g["gain"].max() > 0.5, "the fixture must exercise the gain ramps"


def test_volume_db_gain_is_power_ratio(oracle):
    """A10: MS_VOLUME_SET_DB_GAIN uses 10^(dB/10)"""
    v = oracle.Volume(8000)
    oracle.lib().orc_volume_set_db_gain(v.v, 3.0)
    assert abs(v.v.static_gain - 10 ** 0.3) < 1e-6


# -------------------------------------------------------------- resampler
def test_kaiser_table_is_analytic(oracle):
    from scipy.special import i0
    r = oracle.Resampler(16000, 48000)
    t = r.table().reshape(3, 48)
    # row 0 sits on the integer grid: h[j] = c*sinc(c*(j-23)) * kaiser8(|2(j-23)/48|)
    c = np.float32(0.917)
    j = np.arange(48) - 23
    x = np.abs(2.0 * j / 48)
    w = i0(8 * np.sqrt(np.clip(1 - x ** 2, 0, None))) / i0(8)
    h = c * np.sinc(c * j) * w
    assert np.abs(t[0] - h).max() < 2e-6


def test_resampler_framing_and_golden(oracle):
    for a, b, n in ((16000, 48000, 160), (48000, 16000, 480), (44100, 48000, 441)):
        g = gold(f"resample_{a}_{b}.npz")
        r = oracle.Resampler(a, b)
        np.testing.assert_array_equal(r.table(), g["table"])
        y = np.concatenate([r.process(g["x"][i * n:(i + 1) * n]) for i in range(10)])
        np.testing.assert_array_equal(y, g["y"])
        assert abs(len(y) - 10 * n * b // a) <= 1
    assert oracle.lib().orc_msresample_outcap(160, 16000, 48000) == 481  # msresample.c:151-152


@pytest.mark.parametrize("a,b", [(16000, 48000), (8000, 48000), (48000, 16000), (44100, 48000)])
def test_resampler_tone_response(oracle, a, b):
    """pass-band tone: unity gain, group delay filt_len/2 input samples; stop-band tone: >= 60 dB down."""
    r = oracle.Resampler(a, b)
    n = a // 100
    secs = 0.5
    t = np.arange(int(a * secs)) / a
    for f, want_pass in ((1000.0, True), (0.97 * min(a, b) / 2 + 0.06 * max(a, b) / 2 if a != b else 0, None)):
        if want_pass is None:
            continue
        x = (8000 * np.sin(2 * np.pi * f * t)).astype(np.int16)
        rr = oracle.Resampler(a, b)
        y = np.concatenate([rr.process(x[i:i + n]) for i in range(0, len(x) - n + 1, n)]).astype(np.float64)
        tt = np.arange(len(y)) / b - (r.filt_len / 2) / a
        ref = 8000 * np.sin(2 * np.pi * f * tt)
        sl = slice(4 * r.filt_len * b // a, None)
        k = np.dot(y[sl], ref[sl]) / np.dot(ref[sl], ref[sl])
        assert abs(k - 1.0) < 0.01, (a, b, k)
        assert np.sqrt(np.mean((y[sl] - k * ref[sl]) ** 2)) < 25.0
    if b > a and b % a == 0:  # image of a tone near the input Nyquist must be suppressed by the interpolation filter
        f = 0.45 * a
        x = (8000 * np.sin(2 * np.pi * f * t)).astype(np.int16)
        rr = oracle.Resampler(a, b)
        y = np.concatenate([rr.process(x[i:i + n]) for i in range(0, len(x) - n + 1, n)]).astype(np.float64)
        spec = np.abs(np.fft.rfft(y[2000:] * np.hanning(len(y) - 2000)))
        fr = np.fft.rfftfreq(len(y) - 2000, 1.0 / b)
        main = spec[np.abs(fr - f) < 50].max()
        image = spec[np.abs(fr - (a - f)) < 50].max()
        assert 20 * np.log10(main / image) > 50.0


def _read_wav(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    pos, rate, ch = 12, None, None
    while pos + 8 <= len(raw):
        cid, sz = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            _, ch, rate = struct.unpack("<HHI", raw[pos + 8:pos + 16])
        if cid == b"data":
            # sized from the FILE, not data.len
            return rate, ch, np.frombuffer(raw[pos + 8:pos + 8 + ((len(raw) - pos - 8) // 2) * 2], "<i2")
        pos += 8 + sz
    raise ValueError(path)


RESAMPLE_WAV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample_wav")


def reference_wav_pair():
    """Seconds 3..9 of the reference's tester/sounds/test_silence_voice_{16000,48000}.wav -- the SAME recording
    shipped at two rates (fixture cut by tests/golden/make_resample_excerpts.py)."""
    r16, _, x16 = _read_wav(os.path.join(RESAMPLE_WAV, "voice_16000_6s.wav"))
    r48, _, x48 = _read_wav(os.path.join(RESAMPLE_WAV, "voice_48000_6s.wav"))
    assert (r16, r48) == (16000, 48000)
    return x16[: 160 * 600], x48


def best_alignment_similarity(y, ref, max_shift=200):
    """normalised correlation at the best shift; the resampler delays by 24 input = 72 output samples"""
    y, ref = y.astype(np.float64), ref.astype(np.float64)
    best = 0.0
    for shift in range(0, max_shift):
        a, b = y[shift:], ref[:len(y) - shift]
        best = max(best, np.dot(a, b) / np.sqrt(np.dot(a, a) * np.dot(b, b) + 1e-9))
    return best


def test_resampler_against_reference_wav_pair(oracle):
    """Resampling the 16 k file must reproduce the 48 k one up to the similarity threshold the reference's own
    tester uses for a resampled path (aec3_tester.c:743-758: >= 0.98)."""
    x16, x48 = reference_wav_pair()
    rs = oracle.Resampler(16000, 48000)
    y = np.concatenate([rs.process(x16[i:i + 160]) for i in range(0, len(x16), 160)])
    assert best_alignment_similarity(y, x48[: len(y)]) >= 0.98


WAV_FAMILY_PAIRS = [(8000, 48000), (8000, 16000), (32000, 48000), (44100, 48000), (48000, 16000), (48000, 8000),
                    (48000, 32000), (16000, 8000), (48000, 44100)]


def reference_wav_family(rate, seconds=3):
    """The first `seconds` of the fixture excerpt at `rate`: tester/sounds/test_silence_voice_<rate>.wav from second 3 on
    (all five files are the same recording, time-aligned)."""
    name = f"voice_{rate}_6s.wav" if rate in (16000, 48000) else f"voice_{rate}_3s.wav"
    r, _, x = _read_wav(os.path.join(RESAMPLE_WAV, name))
    assert r == rate
    return x[: seconds * rate]


def resample_in_ticks(make, a, b, x):
    """x through a resampler object in 10 ms blocks (a ragged tail dropped); make(a, b) -> object with .process"""
    rs = make(a, b)
    n = a // 100
    return np.concatenate([rs.process(x[i:i + n]) for i in range(0, len(x) - n + 1, n)])


def family_target(a, b):
    """The file shipped at rate b; when b > a, band-limited to what a file at rate a can hold (0.475 a, linear phase,
    no delay) -- the 8 kHz file cannot reproduce the 4..8 kHz content of the 16 / 48 kHz ones, whatever the resampler."""
    ref = reference_wav_family(b)
    if b > a:
        from scipy.signal import firwin
        h = firwin(255, 0.475 * a, fs=b)
        ref = np.convolve(ref.astype(np.float64), h, mode="same")
    return ref


def check_wav_family(y, a, b):
    ref = family_target(a, b)
    n = min(len(y), len(ref))
    sim = best_alignment_similarity(y[:n], ref[:n], max_shift=400)
    assert sim >= 0.98, (a, b, sim)


@pytest.mark.parametrize("a,b", WAV_FAMILY_PAIRS)
def test_resampler_against_the_reference_wav_family(oracle, a, b):
    """Every ratio family the kernels special-case (x6, x2, 3/2, 2/3, /2, /3, /6, the interpolated 160/147 and 147/160)
    on the reference's own material: resampling the file shipped at rate a reproduces the one shipped at rate b with the
    similarity the reference's tester asks of a resampled path (aec3_tester.c:743-758: >= 0.98)."""
    check_wav_family(resample_in_ticks(oracle.Resampler, a, b, reference_wav_family(a)), a, b)

# -------------------------------------------------------------------- FFT
@pytest.mark.parametrize("n", [128, 256, 512])
def test_fft_vs_numpy(oracle, n):
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    f = oracle.ms_fft(x)
    F = np.fft.rfft(x.astype(np.float64)) / n
    packed = np.zeros(n)
    packed[0], packed[-1] = F[0].real, F[n // 2].real
    packed[1:-1:2], packed[2:-1:2] = F[1:n // 2].real, F[1:n // 2].imag
    assert np.abs(f - packed).max() < 1e-6
    assert np.abs(oracle.ms_ifft(f) - x).max() < 2e-6
    imp = np.zeros(n, np.float32)
    imp[0] = 1
    fi = oracle.ms_fft(imp)
    assert np.allclose(fi[0], 1 / n) and np.allclose(fi[1::2], 1 / n) and np.allclose(fi[2:-1:2], 0)


# -------------------------------------------------------------- equalizer
def test_equalizer_flat_is_windowed_delay(oracle):
    for rate, nfft in ((8000, 128), (16000, 256), (48000, 512)):
        e = oracle.Equalizer(rate)
        t = e.taps()
        assert len(t) == nfft
        assert np.argmax(np.abs(t)) == nfft // 2 and abs(t[nfft // 2] - 1.0) < 1e-6
        assert np.abs(np.delete(t, nfft // 2)).max() < 1e-6


def test_equalizer_gain_shapes_response_and_golden(oracle):
    g = gold("equalizer.npz")
    e = oracle.Equalizer(16000)
    e.set_gain(1000, 2.0, 500)
    e.set_gain(300, 0.3, 100)
    np.testing.assert_array_equal(e.taps(), g["taps"])
    y = np.concatenate([e.run(g["x"][i * 160:(i + 1) * 160]) for i in range(6)])
    np.testing.assert_array_equal(y, g["y"])
    H = np.abs(np.fft.rfft(g["taps"].astype(np.float64), 4096))
    fr = np.fft.rfftfreq(4096, 1 / 16000)
    assert 1.6 < H[np.argmin(np.abs(fr - 1000))] < 2.2
    assert H[np.argmin(np.abs(fr - 300))] < 0.7
    assert 0.9 < H[np.argmin(np.abs(fr - 5000))] < 1.1


def test_fir_matches_direct_convolution(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    h = rng.standard_normal(128).astype(np.float32)
    x = rng.standard_normal(400).astype(np.float32)
    mem = np.zeros(128, np.float32)
    y = np.zeros(400, np.float32)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    oracle.lib().orc_fir_mem16(p(x), p(h), p(y), 400, 128, p(mem))
    ref = np.convolve(x.astype(np.float64), h.astype(np.float64))[:400]
    assert np.abs(y - ref).max() < 1e-3


# ----------------------------------------------------------------- scaler
def test_scaler_identity_constant_golden_and_float_reference(oracle):
    g = gold("scaler.npz")
    np.testing.assert_array_equal(oracle.i420_scale_to_rgb24(g["src"], 64, 48, 40, 30), g["rgb"])
    np.testing.assert_array_equal(oracle.i420_scale(g["src"], 64, 48, 40, 30), g["i420"])
    np.testing.assert_array_equal(oracle.i420_scale(g["src"], 64, 48, 64, 48), g["src"])
    flat = np.full(oracle.i420_size(64, 48), 77, np.uint8)
    assert (oracle.i420_scale(flat, 64, 48, 24, 18) == 77).all()
    # independent float64 bilinear with pixel-centre mapping: within 1 LSB on a smooth picture
    sw, sh, dw, dh = 96, 64, 64, 48
    yy, xx = np.mgrid[0:sh, 0:sw]
    Y = (40 + 1.5 * xx + 0.9 * yy).clip(0, 255)
    src = np.concatenate([Y.astype(np.uint8).ravel(), np.full(sw * sh // 2, 128, np.uint8)])
    got = oracle.i420_scale(src, sw, sh, dw, dh)[:dw * dh].reshape(dh, dw).astype(np.float64)
    Yq = Y.astype(np.uint8).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * sw / dw - 0.5
    sy = (np.arange(dh) + 0.5) * sh / dh - 0.5
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0
    x1, y1 = np.minimum(x0 + 1, sw - 1), np.minimum(y0 + 1, sh - 1)
    top = Yq[y0][:, x0] * (1 - fx) + Yq[y0][:, x1] * fx
    bot = Yq[y1][:, x0] * (1 - fx) + Yq[y1][:, x1] * fx
    ref = top * (1 - fy[:, None]) + bot * fy[:, None]
    assert np.abs(got - ref).max() <= 1.0


def test_bt601_known_answers(oracle):
    for (y, u, v), rgb in {(16, 128, 128): (0, 0, 0), (235, 128, 128): (255, 255, 255), (81, 90, 240): (255, 0, 0),
                           (145, 54, 34): (0, 255, 0), (41, 240, 110): (0, 0, 255)}.items():
        f = np.concatenate([np.full(64, y, np.uint8), np.full(16, u, np.uint8), np.full(16, v, np.uint8)])
        got = oracle.i420_to_rgb24(f, 8, 8)
        ref = np.array([1.164 * (y - 16) + 1.596 * (v - 128), 1.164 * (y - 16) - 0.391 * (u - 128) - 0.813 * (v - 128),
                        1.164 * (y - 16) + 2.018 * (u - 128)]).clip(0, 255)   # src/yuv2rgb.fs
        assert np.abs(got[0, 0].astype(float) - ref).max() <= 1.0
        assert np.abs(got[0, 0].astype(int) - np.array(rgb)).max() <= 1


# -------------------------------------------------------------------- AEC
def test_aec_sizing_and_golden(oracle):
    L = oracle.lib()
    assert L.adjust_framesize_8000(64, 8000) == 64      # speexec.c:171-180
    assert L.adjust_framesize_8000(64, 16000) == 128
    assert L.adjust_framesize_8000(64, 48000) == 256    # 384 -> largest 2^k below (A23)
    assert L.adjust_framesize_8000(64, 44100) == 256
    g = gold("aec.npz")
    F = 128
    ec = oracle.Echo(F, 2048, 16000)
    pp = oracle.Preproc(F, 16000, ec)
    o1, o2 = [], []
    for f in range(30):
        sl = slice(f * F, (f + 1) * F)
        o = ec.cancel(g["mic"][sl], g["far"][sl])
        o1.append(o)
        o2.append(pp.run(o))
    np.testing.assert_array_equal(np.concatenate(o1), g["out"])
    np.testing.assert_array_equal(np.concatenate(o2), g["post"])
    np.testing.assert_array_equal(ec.get("W", 16 * 256), g["W"])


def test_aec_cancels_a_synthetic_echo(oracle):
    rate, F, flen = 16000, 128, 2048
    rng = np.random.default_rng(1)
    n = F * 400
    far = np.convolve(rng.normal(0, 3000, n), [0.5, 0.3, 0.2])[:n]
    ir = rng.normal(0, 1, 64) * np.exp(-np.arange(64) / 12.0)
    ir /= np.sqrt((ir ** 2).sum())
    mic = 0.5 * np.convolve(np.concatenate([np.zeros(320), far]), ir)[:n] + rng.normal(0, 30, n)
    to16 = lambda v: np.clip(np.round(v), -32767, 32767).astype(np.int16)
    mic, far = to16(mic), to16(far)
    ec = oracle.Echo(F, flen, rate)
    pp = oracle.Preproc(F, rate, ec)
    out = np.zeros(n, np.int16)
    post = np.zeros(n, np.int16)
    for f in range(400):
        sl = slice(f * F, (f + 1) * F)
        out[sl] = ec.cancel(mic[sl], far[sl])
        post[sl] = pp.run(out[sl])
    db = lambda v: 10 * np.log10(np.mean(v.astype(np.float64) ** 2) + 1e-9)
    tail = slice(-100 * F, None)
    assert db(mic[tail]) - db(out[tail]) > 25.0       # linear canceller ERLE
    assert db(mic[tail]) - db(post[tail]) > 40.0      # + residual echo suppression
    assert ec.get("scalars", 16)[8] == 1.0            # adapted


# ---------------------------------------------------------------- pixconv (packed -> I420)
def test_pixconv_known_answers(oracle):
    """BT.601 limited-range anchors of the libyuv rows (white/black/primaries), the JPEG set for MS_RGB24,
    and the 4:2:2 chroma average with its +1 rounding."""
    w, h = 4, 2

    def solid(fmt, r, g, b):
        px = {oracle.PIX_BGR24: [b, g, r], oracle.PIX_RGB24_RAW: [r, g, b], oracle.PIX_BGRA32: [b, g, r, 255]}[fmt]
        out = oracle.pixconv_to_i420(fmt, np.array(px * (w * h), np.uint8), w, h)
        return int(out[0]), int(out[w * h]), int(out[w * h + (w // 2) * (h // 2)])

    assert solid(oracle.PIX_RGB24_RAW, 255, 255, 255) == (235, 128, 128)
    assert solid(oracle.PIX_RGB24_RAW, 0, 0, 0) == (16, 128, 128)
    assert solid(oracle.PIX_BGRA32, 255, 0, 0) == (82, 90, 240)     # red   (66*255+0x1080)>>8 = 82, hand-computed
    assert solid(oracle.PIX_BGRA32, 0, 255, 0) == (144, 54, 34)     # green (129*255+0x1080)>>8 = 144
    assert solid(oracle.PIX_BGRA32, 0, 0, 255) == (41, 240, 110)    # blue
    assert solid(oracle.PIX_BGR24, 255, 255, 255) == (255, 128, 128)  # full range (J420)
    assert solid(oracle.PIX_BGR24, 0, 0, 0) == (0, 128, 128)
    # YUY2: luma is copied, chroma is the rounded mean of the two rows
    row0 = [10, 100, 20, 200, 30, 101, 40, 201]
    row1 = [50, 103, 60, 203, 70, 102, 80, 204]
    out = oracle.pixconv_to_i420(oracle.PIX_YUY2, np.array(row0 + row1, np.uint8), 4, 2)
    assert list(out[:8]) == [10, 20, 30, 40, 50, 60, 70, 80]
    assert list(out[8:10]) == [(100 + 103 + 1) >> 1, (101 + 102 + 1) >> 1]
    assert list(out[10:12]) == [(200 + 203 + 1) >> 1, (201 + 204 + 1) >> 1]
    uy = oracle.pixconv_to_i420(oracle.PIX_UYVY, np.array([100, 10, 200, 20, 101, 30, 201, 40] + [103, 50, 203, 60, 102, 70, 204, 80], np.uint8), 4, 2)
    np.testing.assert_array_equal(uy, out)


def test_pixconv_matches_numpy_restatement(oracle):
    """Independent vectorised restatement of the RGB rows (nested rounded averages, Q8 coefficients)."""
    rng = np.random.default_rng(11)
    w, h = 32, 17  # odd height: the last row pairs with itself
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = oracle.pixconv_to_i420(oracle.PIX_RGB24_RAW, src.ravel(), w, h)
    r, g, b = [src[..., k].astype(np.int64) for k in range(3)]
    y = (66 * r + 129 * g + 25 * b + 0x1080) >> 8
    np.testing.assert_array_equal(got[: w * h].reshape(h, w), y)
    hp = h + 1
    pad = np.concatenate([src, src[-1:]], 0).astype(np.int64)  # row h-1 again
    avg = lambda a, c: (a + c + 1) >> 1
    a2 = avg(avg(pad[0::2, 0::2], pad[1::2, 0::2]), avg(pad[0::2, 1::2], pad[1::2, 1::2]))
    ar, ag, ab = a2[..., 0], a2[..., 1], a2[..., 2]
    u = (112 * ab - 74 * ag - 38 * ar + 0x8080) >> 8
    v = (112 * ar - 94 * ag - 18 * ab + 0x8080) >> 8
    cw, chh = w // 2, hp // 2
    np.testing.assert_array_equal(got[w * hp: w * hp + cw * chh].reshape(chh, cw), u)
    np.testing.assert_array_equal(got[w * hp + cw * chh:].reshape(chh, cw), v)


def test_oracle_extremum_windows(oracle):
    """OrtpExtremum as MSVolume uses it (msvolume.c:115-116,405-406; oracle/conference.c): the extremum of the window a value was
    recorded in; a record more than `period` ms after the window's first one closes it and opens the next with itself"""
    e = oracle.Extremum(1000)
    assert e.current == 0.0
    seen = []
    for t, v in [(0, .1), (500, .3), (900, .2), (1000, .25), (1001, .05), (1500, .04), (2001, .02), (2002, .01), (2003, .5)]:
        e.record_max(t, v)
        seen.append(round(e.current, 3))
    assert seen == [.1, .3, .3, .3, .05, .05, .05, .01, .5]     # (1000 is not MORE than the period; 2001 is not either, counted from 1001; 2002 is)
    assert abs(e.e.last_stable - .05) < 1e-6
    m = oracle.Extremum(30000)                                  # the minimum, 30 s windows (msvolume.c:116)
    for t, v in [(10, .2), (20, .1), (30000, .3), (30011, .4), (30012, .35)]:
        m.record_min(t, v)
    assert abs(m.current - .35) < 1e-6 and abs(m.e.last_stable - .1) < 1e-6
    e.reset()
    assert e.current == 0.0                                     # volume_preprocess (:467-468): nothing recorded yet reads as 0 -> -120 dB
    assert oracle.linear_to_dbm0(0.0) == -120 and abs(oracle.linear_to_dbm0(0.001) + 30) < 1e-5


def test_oracle_conference_bookkeeping_and_election(oracle):
    """src/voip/audioconference.c in mixer mode as oracle/conference.c restates it: the lowest free pin (:198-207), sizes (:390-392),
    muting (:376-388), a participant's volume (:394-418), the election (:419-464)"""
    c = oracle.Conference()
    assert [c.add_member() for _ in range(4)] == [0, 1, 2, 3] and c.size == 4 and c.active_speaker == -1
    c.remove_member(1)
    assert c.size == 3 and c.add_member() == 1 and c.order == [0, 2, 3, 1]         # the freed pin is taken again; the list appends
    # strictly above -30 dB, strictly above the best so far: of two equals the first in the LIST wins (pin 2 joined before pin 1)
    assert c.process_events({0: -40.0, 1: -20.0, 2: -20.0, 3: -50.0}) == (True, 2, -20.0) and c.active_speaker == 2
    assert c.process_events({0: -40.0, 1: -20.0, 2: -20.0, 3: -50.0}) == (False, 2, -20.0)
    assert c.process_events({0: -30.0, 1: -30.0, 2: -31.0, 3: -120.0}) == (False, -1, -120.0) and c.active_speaker == 2   # -30 itself is not above: nobody; the speaker stays
    c.mute_member(2, True)
    assert c.process_events({0: -40.0, 1: -25.0, 2: -5.0, 3: -50.0}) == (True, 1, -25.0)      # a muted member is passed over (:445)
    assert c.participant_volume(2, -5.0) == -120 and c.participant_volume(1, -25.7) == -25 and c.participant_volume(7, 0.0) == -32768
    c.mute_member(2, False)
    assert c.process_events({0: -40.0, 1: -25.0, 2: -5.0, 3: -50.0}) == (True, 2, -5.0)
    c.remove_member(2)
    assert c.active_speaker == 2 and c.size == 3                                   # the pointer to the one who left stays until somebody wins (:460-464)
    assert c.process_events({0: -10.0, 1: -25.0, 3: -50.0}) == (True, 0, -10.0)
    full = oracle.Conference()
    assert [full.add_member() for _ in range(50)] == list(range(50)) and full.add_member() == -1     # MIXER_MAX_CHANNELS (audiomixer.c:29); the reference aborts
