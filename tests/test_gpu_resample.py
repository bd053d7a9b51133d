"""GPU parity: mi_resampler_* (HIP) vs the CPU oracle's restatement of the
speex resampler as msresample.c:150-177 drives it.  Float path: the kernels use
fused multiply-add, the oracle separate mul/add, so the int16 outputs may differ
by 1 LSB where a value sits on a rounding boundary.  Tolerances (north_star):
RMS error <= 1e-4 of full scale; we also require max |diff| <= 1 LSB."""
import numpy as np
import pytest

import mediastreamer2_amd as ms
from conftest import synth_pcm

pytestmark = pytest.mark.gpu

FULL_SCALE = 32768.0
RMS_TOL = 1e-4  # relative to full scale, BASELINE.json north_star


def _run_both(ctx, oracle, in_rate, out_rate, nstreams, in_len, nticks, sigma=3000.0):
    rs = ms.ResamplerBatch(ctx, nstreams, in_rate, out_rate)
    orcs = [oracle.Resampler(in_rate, out_rate) for _ in range(nstreams)]
    x_all = np.stack([synth_pcm(s, in_len * nticks, sigma=sigma, rate=in_rate) for s in range(nstreams)])
    worst, sq, cnt = 0, 0.0, 0
    for t in range(nticks):
        x = x_all[:, t * in_len:(t + 1) * in_len]
        out, olen = rs.process(np.ascontiguousarray(x))
        for s in range(nstreams):
            ref = orcs[s].process(x[s])
            assert olen[s] == len(ref), (s, t, olen[s], len(ref))
            d = out[s, :olen[s]].astype(np.int32) - ref.astype(np.int32)
            worst = max(worst, int(np.abs(d).max()))
            sq += float((d.astype(np.float64) ** 2).sum())
            cnt += len(d)
    rs.close()
    return worst, np.sqrt(sq / cnt) / FULL_SCALE


@pytest.mark.parametrize("in_rate,out_rate,in_len", [
    (16000, 48000, 160),   # BASELINE config 2 (fast up-sampling kernel, den=3)
    (8000, 48000, 80),     # BASELINE config 1 (den=6)
    (8000, 16000, 80),     # den=2
    (48000, 16000, 480),   # down-sampling, 144 taps, generic kernel
    (44100, 48000, 441),   # interpolated (non-direct) table
    (16000, 8000, 160),
    (48000, 44100, 480),
    (48000, 8000, 480),    # integer down-sampling fast kernel, 6 input phases (288 taps)
    (32000, 8000, 320),    # 4 phases
    (48000, 24000, 480),   # 2 phases
    (32000, 48000, 320),   # rational 3/2 kernel
    (48000, 32000, 480),   # rational 2/3 kernel (72 taps)
    (16000, 24000, 160),
    (12000, 8000, 120),
])
def test_resampler_matches_oracle(ctx, oracle, in_rate, out_rate, in_len):
    worst, rms = _run_both(ctx, oracle, in_rate, out_rate, nstreams=9, in_len=in_len, nticks=12)
    assert worst <= 1, f"max |gpu-oracle| = {worst} LSB"
    assert rms <= RMS_TOL, f"rms = {rms}"


def test_resampler_table_equals_oracle_table(ctx, oracle):
    for a, b in ((16000, 48000), (48000, 16000), (44100, 48000), (8000, 48000)):
        rs = ms.ResamplerBatch(ctx, 1, a, b)
        o = oracle.Resampler(a, b)
        assert rs.info() == {"filt_len": o.filt_len, "den_rate": o.den_rate, "num_rate": o.num_rate,
                             "direct": int(o.direct)}
        np.testing.assert_array_equal(rs.table(), o.table())
        rs.close()


def test_resampler_edge_inputs(ctx, oracle):
    """silence, full-scale square (saturation in WORD2INT), -32768, ragged block sizes."""
    n = 6
    rs = ms.ResamplerBatch(ctx, n, 16000, 48000)
    orcs = [oracle.Resampler(16000, 48000) for _ in range(n)]
    for blk in (160, 8, 3, 1, 157, 160, 320):
        x = np.zeros((n, blk), np.int16)
        x[1] = 32767
        x[2] = -32768
        x[3] = np.where(np.arange(blk) % 2 == 0, 32767, -32768)
        x[4] = synth_pcm(4, blk, sigma=20000.0, rate=16000)
        x[5] = np.where((np.arange(blk) // 5) % 2 == 0, 32767, -32767)
        out, olen = rs.process(x)
        for s in range(n):
            ref = orcs[s].process(x[s])
            assert olen[s] == len(ref)
            d = np.abs(out[s, :olen[s]].astype(np.int32) - ref.astype(np.int32))
            assert d.max(initial=0) <= 1, (blk, s, d.max())
    rs.close()


def test_resampler_reset_and_independence(ctx, oracle):
    """resetting one stream zeroes its history only; streams do not leak into each other."""
    rs = ms.ResamplerBatch(ctx, 4, 16000, 48000)
    x = np.stack([synth_pcm(s, 160, rate=16000) for s in range(4)])
    first, _ = rs.process(x)
    second, _ = rs.process(x)
    assert not np.array_equal(first, second)  # history matters
    rs.reset(2, 1)
    third, _ = rs.process(x)
    np.testing.assert_array_equal(third[2], first[2])
    o = oracle.Resampler(16000, 48000)
    for _ in range(3):
        ref = o.process(x[0])
    assert np.abs(third[0, :480].astype(int) - ref.astype(int)).max() <= 1
    rs.close()


def test_resampler_full_size_config2_properties(ctx, oracle):
    """BASELINE config 2 at full size (4096 streams), device-resident path:
    identical streams give identical outputs, and sampled streams match the oracle."""
    torch = pytest.importorskip("torch")
    n, in_len = 4096, 160
    rs = ms.ResamplerBatch(ctx, n, 16000, 48000)
    base = np.stack([synth_pcm(s % 64, in_len * 3, rate=16000) for s in range(n)])
    x = torch.from_numpy(base).cuda()
    outs = []
    for t in range(3):
        xt = x[:, t * in_len:(t + 1) * in_len].contiguous()
        out, _ = rs.process(xt)
        ctx.sync()
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy()[:, :480])
    for t in range(3):
        o = outs[t].reshape(64, 64, 480)  # stream s and s+64k carry the same signal
        assert (o == o[:1]).all()
    for s in (0, 1, 63, 4095):
        orc = oracle.Resampler(16000, 48000)
        for t in range(3):
            ref = orc.process(base[s, t * in_len:(t + 1) * in_len])
            assert np.abs(outs[t][s].astype(int) - ref.astype(int)).max() <= 1
    rs.close()


def test_resampler_on_the_reference_wav_pair(ctx, oracle):
    """The reference ships one recording at 16 kHz and at 48 kHz (tester/sounds/test_silence_voice_*.wav, fixture
    excerpt in tests/golden/resample_wav/): the 16 k file resampled on the GPU must reproduce the 48 k file with
    the similarity the reference's tester demands of a resampled path (aec3_tester.c:743-758: >= 0.98), and
    equal the oracle within 1 LSB."""
    from test_oracle_cpu import reference_wav_pair, best_alignment_similarity
    x16, x48 = reference_wav_pair()
    rs = ms.ResamplerBatch(ctx, 3, 16000, 48000)
    orc = oracle.Resampler(16000, 48000)
    got, want = [], []
    for i in range(0, len(x16), 160):
        blk = np.stack([x16[i:i + 160], np.zeros(160, np.int16), x16[i:i + 160]])
        out, olen = rs.process(blk)
        assert list(olen) == [480, 480, 480]
        np.testing.assert_array_equal(out[0], out[2])
        got.append(out[0][:480].copy())
        want.append(orc.process(x16[i:i + 160]))
    got, want = np.concatenate(got), np.concatenate(want)
    assert np.abs(got.astype(np.int32) - want).max() <= 1
    assert best_alignment_similarity(got, x48[: len(got)]) >= 0.98


def test_downsampler_fast_and_generic_paths_interleave(ctx, oracle):
    # The integer down-sampling kernel needs every stream on the output-period grid; a ragged block hands the batch to
    # the generic kernel for good (the library's (last_sample, frac) leaves zero), a reset brings it back.  Either way
    # the stream of samples must keep matching the oracle.
    n = 5
    rs = ms.ResamplerBatch(ctx, n, 48000, 16000)
    orcs = [oracle.Resampler(48000, 16000) for _ in range(n)]
    x = np.stack([synth_pcm(40 + s, 480 * 12, rate=48000) for s in range(n)])
    pos = 0
    for blk in (480, 480, 479, 481, 480, 240, 480):
        out, olen = rs.process(np.ascontiguousarray(x[:, pos:pos + blk]))
        for s in range(n):
            ref = orcs[s].process(x[s, pos:pos + blk])
            assert olen[s] == len(ref)
            assert np.abs(out[s, :olen[s]].astype(int) - ref).max() <= 1
        pos += blk
    rs.reset()
    orcs = [oracle.Resampler(48000, 16000) for _ in range(n)]
    for t in range(3):
        out, olen = rs.process(np.ascontiguousarray(x[:, t * 480:(t + 1) * 480]))
        for s in range(n):
            ref = orcs[s].process(x[s, t * 480:(t + 1) * 480])
            assert olen[s] == len(ref) == 160
            assert np.abs(out[s, :160].astype(int) - ref).max() <= 1
    rs.close()


from test_oracle_cpu import WAV_FAMILY_PAIRS  # noqa: E402


@pytest.mark.parametrize("a,b", WAV_FAMILY_PAIRS)
def test_resampler_on_the_reference_wav_family(ctx, oracle, a, b):
    """The reference ships ONE recording at 8 / 16 / 32 / 44.1 / 48 kHz (tester/sounds/test_silence_voice_*.wav, excerpts in
    tests/golden/resample_wav/).  For every ratio family the kernels special-case, the file at rate a resampled on the GPU
    must reproduce the file at rate b with the similarity the reference's tester demands of a resampled path (>= 0.98),
    and equal the oracle within 1 LSB."""
    from test_oracle_cpu import check_wav_family, reference_wav_family
    x = reference_wav_family(a)
    n = a // 100
    rs = ms.ResamplerBatch(ctx, 2, a, b)
    orc = oracle.Resampler(a, b)
    got, want = [], []
    for i in range(0, len(x) - n + 1, n):
        blk = np.stack([x[i:i + n], x[i:i + n]])
        out, olen = rs.process(blk)
        assert olen[0] == olen[1]
        np.testing.assert_array_equal(out[0], out[1])
        got.append(out[0][:olen[0]].copy())
        want.append(orc.process(x[i:i + n]))
    got, want = np.concatenate(got), np.concatenate(want)
    assert got.size == want.size
    assert np.abs(got.astype(np.int32) - want).max() <= 1
    check_wav_family(got, a, b)
    rs.close()


@pytest.mark.parametrize("in_rate,out_rate,in_len", [(16000, 48000, 160), (44100, 48000, 441), (48000, 16000, 480)])
def test_a_streams_state_moves_to_another_resampler(ctx, in_rate, out_rate, in_len):
    """mi_resampler_get_state / set_state: what a speex handle carries from call to call (position + history).  A stream resampled
    in one object, and the same stream whose state is moved to a slot of ANOTHER object half-way (what the plugin's fused call leg
    does when its conference is re-plumbed: msresample.c keeps the handle across a detach), deliver the same samples."""
    n, nticks = 3, 12
    x = np.stack([synth_pcm(40 + s, in_len * nticks, sigma=3000.0, rate=in_rate) for s in range(n)])
    a = ms.ResamplerBatch(ctx, n, in_rate, out_rate)
    want = []
    for t in range(nticks):
        out, olen = a.process(np.ascontiguousarray(x[:, t * in_len:(t + 1) * in_len]))
        want.append([out[s, :olen[s]].copy() for s in range(n)])
    a.close()
    b, c = ms.ResamplerBatch(ctx, n, in_rate, out_rate), ms.ResamplerBatch(ctx, 5, in_rate, out_rate)
    got = []
    for t in range(nticks // 2):
        out, olen = b.process(np.ascontiguousarray(x[:, t * in_len:(t + 1) * in_len]))
        got.append([out[s, :olen[s]].copy() for s in range(n)])
    where = [4, 0, 2]                                         # stream s of b carries on in slot where[s] of c
    for s in range(n):
        st = b.get_state(s)
        assert len(st) == ctx.L.mi_resampler_state_bytes(b.h)
        c.set_state(where[s], st)
    for t in range(nticks // 2, nticks):
        blk = np.zeros((5, in_len), np.int16)
        for s in range(n):
            blk[where[s]] = x[s, t * in_len:(t + 1) * in_len]
        out, olen = c.process(blk)
        got.append([out[where[s], :olen[where[s]]].copy() for s in range(n)])
    for t in range(nticks):
        for s in range(n):
            np.testing.assert_array_equal(got[t][s], want[t][s], err_msg=f"tick {t} stream {s}")
    b.close()
    c.close()
