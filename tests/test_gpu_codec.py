"""G.711 / L16 / channel adapter / audio flow controller through the C ABI vs the oracle.

All integer work: bit-exact.  The oracle's four G.711 conversions are themselves pinned against the reference's
own g711.c (tests/test_oracle_cpu.py::test_g711_matches_the_reference_build), and the GPU kernels are compared with
that build directly here when oracle/_ref travelled with the snapshot."""
import numpy as np
import pytest
import torch

import mediastreamer2_amd as ms

pytestmark = pytest.mark.gpu


def dev(x):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    torch.cuda.synchronize()
    return t


def host(ctx, t):
    ctx.sync()
    return t.cpu().numpy()


@pytest.mark.parametrize("law", [ms.MI_LAW_PCMA, ms.MI_LAW_PCMU])
def test_g711_decode_every_code_word(ctx, oracle, law):
    codes = np.tile(np.arange(256, dtype=np.uint8), 4).reshape(4, 256)
    pcm = torch.zeros((4, 256), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    got = host(ctx, ms.g711_decode(ctx, law, dev(codes), pcm))
    np.testing.assert_array_equal(got, oracle.g711_decode(law, codes))
    R = oracle.g711_ref()
    if R is not None:  # the reference's own g711.c
        fn = R.Snack_Alaw2Lin if law == ms.MI_LAW_PCMA else R.Snack_Mulaw2Lin
        np.testing.assert_array_equal(got[0], np.array([fn(int(c)) for c in range(256)], np.int16))


@pytest.mark.parametrize("law", [ms.MI_LAW_PCMA, ms.MI_LAW_PCMU])
def test_g711_encode_every_pcm_value(ctx, oracle, law):
    pcm = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16).reshape(64, 1024)
    codes = torch.zeros((64, 1024), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    got = host(ctx, ms.g711_encode(ctx, law, dev(pcm), codes))
    np.testing.assert_array_equal(got, oracle.g711_encode(law, pcm))
    R = oracle.g711_ref()
    if R is not None:
        fn = R.Snack_Lin2Alaw if law == ms.MI_LAW_PCMA else R.Snack_Lin2Mulaw
        want = np.array([fn(int(v)) for v in pcm.ravel()], np.uint8).reshape(pcm.shape)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("law", [ms.MI_LAW_PCMA, ms.MI_LAW_PCMU])
@pytest.mark.parametrize("rows,n,stride", [(1, 160, 160), (37, 80, 96), (5, 163, 176), (3, 7, 16), (2, 1120, 1120), (9, 33, 33)])
def test_g711_ragged_rows_and_strides(ctx, oracle, law, rows, n, stride):
    rng = np.random.default_rng(rows * 1000 + n)
    codes = rng.integers(0, 256, (rows, stride), dtype=np.uint8)
    lens = rng.integers(0, n + 1, rows).astype(np.int32)
    lens[0] = n
    sentinel = 12345
    pcm = torch.full((rows, stride), sentinel, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    got = host(ctx, ms.g711_decode(ctx, law, dev(codes), pcm, length=n, lens=dev(lens)))
    for r in range(rows):
        np.testing.assert_array_equal(got[r, : lens[r]], oracle.g711_decode(law, codes[r, : lens[r]]))
        assert (got[r, lens[r]:] == sentinel).all()  # nothing written past a row's count
    # and back: encode(decode(c)) over the same ragged layout
    back = torch.full((rows, stride), 0xEE, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    enc = host(ctx, ms.g711_encode(ctx, law, dev(got), back, length=n, lens=dev(lens)))
    for r in range(rows):
        np.testing.assert_array_equal(enc[r, : lens[r]], oracle.g711_encode(law, got[r, : lens[r]]))
        assert (enc[r, lens[r]:] == 0xEE).all()


@pytest.mark.parametrize("law", [ms.MI_LAW_PCMA, ms.MI_LAW_PCMU])
def test_g711_round_trip_is_idempotent_at_full_size(ctx, law):
    """Size-independent property: decode o encode o decode == decode (every code word is a fixed point of the pair,
    up to the two zeros of mu-law), on 65 536 streams x 80 samples."""
    g = torch.Generator(device="cpu").manual_seed(law)
    codes = torch.randint(0, 256, (65536, 80), dtype=torch.uint8, generator=g).cuda()
    pcm = torch.empty((65536, 80), dtype=torch.int16, device="cuda")
    codes2 = torch.empty_like(codes)
    pcm2 = torch.empty_like(pcm)
    torch.cuda.synchronize()
    ms.g711_decode(ctx, law, codes, pcm)
    ms.g711_encode(ctx, law, pcm, codes2)
    ms.g711_decode(ctx, law, codes2, pcm2)
    ctx.sync()
    assert torch.equal(pcm, pcm2)
    if law == ms.MI_LAW_PCMA:
        assert torch.equal(codes, codes2)
    else:  # 0x7F (negative zero) decodes to 0 and re-encodes as 0xFF
        assert torch.equal(torch.where(codes == 0x7F, torch.full_like(codes, 0xFF), codes), codes2)


@pytest.mark.parametrize("n", [1, 7, 8, 480, 4099])
def test_l16_swap(ctx, oracle, n):
    x = np.random.default_rng(n).integers(-32768, 32768, n).astype(np.int16)
    out = torch.zeros(n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    got = host(ctx, ms.l16_swap(ctx, dev(x), out))
    np.testing.assert_array_equal(got, oracle.l16_swap(x))
    assert got.view(np.uint8)[0] == x.view(np.uint8)[1]
    xin = dev(x)
    np.testing.assert_array_equal(host(ctx, ms.l16_swap(ctx, xin, xin)), oracle.l16_swap(x))  # in place


@pytest.mark.parametrize("frames", [1, 8, 160, 483])
def test_channel_adapter_modes(ctx, oracle, frames):
    rng = np.random.default_rng(frames)
    a = rng.integers(-32768, 32768, frames).astype(np.int16)
    b = rng.integers(-32768, 32768, frames).astype(np.int16)
    st = rng.integers(-32768, 32768, 2 * frames).astype(np.int16)
    out2 = torch.zeros(2 * frames, dtype=torch.int16, device="cuda")
    out1 = torch.zeros(frames, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(ctx, ms.chan_adapt(ctx, ms.MI_CHAN_MONO_TO_STEREO, dev(a), out2)), oracle.chan_adapt(0, a))
    np.testing.assert_array_equal(host(ctx, ms.chan_adapt(ctx, ms.MI_CHAN_STEREO_TO_MONO, dev(st), out1)), oracle.chan_adapt(1, st))
    np.testing.assert_array_equal(host(ctx, ms.chan_adapt(ctx, ms.MI_CHAN_TWO_MONO_TO_STEREO, dev(a), out2, b=dev(b))), oracle.chan_adapt(2, a, b))
    np.testing.assert_array_equal(host(ctx, ms.chan_adapt(ctx, ms.MI_CHAN_TWO_MONO_TO_STEREO, dev(a), out2)), oracle.chan_adapt(2, a, None))


def speechy(rng, n, loud=True):
    t = np.arange(n)
    x = 6000 * np.sin(2 * np.pi * t * 220 / 16000) * (0.5 + 0.5 * np.sin(2 * np.pi * t / 1600)) + rng.normal(0, 300 if loud else 20, n)
    if not loud:
        x *= 0.01
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("strategy", [ms.MI_FLOWCTL_SOFT, ms.MI_FLOWCTL_BASIC])
def test_flow_controller_follows_the_oracle_block_by_block(ctx, oracle, strategy):
    """16 streams, 60 blocks of 160 samples at 16 kHz: different drop requests (none, small, one larger than a block's
    eighth, silent frames), a second request while the first is still running (ignored, flowcontrol.c:213), and a
    re-arm after completion."""
    S, n, ticks = 16, 160, 60
    rng = np.random.default_rng(7 + strategy)
    fc = ms.FlowControlBatch(ctx, S, 256)
    fc.set_config(strategy, 0.02)
    refs = [oracle.FlowCtl(strategy, 0.02) for _ in range(S)]
    drop = np.array([0, 16, 40, 160, 320, 7, 1, 100, 480, 33, 64, 0, 250, 12, 900, 5], np.uint32)
    total = np.array([0, 1600, 1600, 3200, 3200, 800, 160, 480, 4800, 1000, 640, 0, 2000, 160, 3000, 8000], np.uint32)
    x = torch.zeros((S, 256), dtype=torch.int16, device="cuda")
    out = torch.zeros((S, 256), dtype=torch.int16, device="cuda")
    olen = torch.zeros(S, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dropped_any = 0
    for t in range(ticks):
        if t in (2, 5, 40):  # t=5: most are still running and must ignore it
            req_d, req_t = (drop, total) if t != 5 else (np.full(S, 48, np.uint32), np.full(S, 320, np.uint32))
            fc.request_drop(req_d, req_t)
            for s in range(S):
                if (req_d[s] or req_t[s]) and not (refs[s].c.total_samples > 0 and refs[s].c.target_samples > 0):
                    refs[s].set_target(int(req_d[s]), int(req_t[s]))
        blocks = np.stack([speechy(rng, n, loud=not (s % 4 == 3 and t % 3 == 0)) for s in range(S)])
        x[:, :n].copy_(torch.from_numpy(blocks))
        torch.cuda.synchronize()
        fc.process(x, out, olen, length=n)
        ctx.sync()
        got, gl = out.cpu().numpy(), olen.cpu().numpy()
        for s in range(S):
            want = refs[s].process(blocks[s])
            assert gl[s] == want.size, (t, s)
            np.testing.assert_array_equal(got[s, : gl[s]], want, err_msg=f"tick {t} stream {s}")
            st = fc.state(s)
            c = refs[s].c
            assert (st["target"], st["total"], st["pos"], st["dropped"]) == (c.target_samples, c.total_samples, c.current_pos, c.current_dropped)
            dropped_any += n - gl[s]
    assert dropped_any > 1000  # the scenario really exercised the droppers


def test_flow_controller_ragged_blocks_and_in_place(ctx, oracle):
    """Per-stream block lengths (0 = no block this round) and d_out == d_in."""
    S, cap = 9, 1920
    rng = np.random.default_rng(3)
    fc = ms.FlowControlBatch(ctx, S, cap)
    refs = [oracle.FlowCtl() for _ in range(S)]
    fc.request_drop(np.full(S, 300, np.uint32), np.full(S, 20000, np.uint32))
    for r in refs:
        r.set_target(300, 20000)
    x = torch.zeros((S, cap), dtype=torch.int16, device="cuda")
    olen = torch.zeros(S, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for t in range(12):
        lens = rng.integers(0, 5, S) * 480
        lens[t % S] = 0
        blocks = [speechy(rng, int(L)) for L in lens]
        for s in range(S):
            if lens[s]:
                x[s, : lens[s]].copy_(torch.from_numpy(blocks[s]))
        dl = torch.from_numpy(lens.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        fc.process(x, x, olen, length=cap, lens=dl)
        ctx.sync()
        got, gl = x.cpu().numpy(), olen.cpu().numpy()
        for s in range(S):
            if lens[s] == 0:
                assert gl[s] == 0
                continue
            want = refs[s].process(blocks[s])
            assert gl[s] == want.size
            np.testing.assert_array_equal(got[s, : gl[s]], want)
    fc.reset()
    assert fc.state(0) == dict(target=0, total=0, pos=0, dropped=0)


def test_flow_controller_randomised_scenarios(ctx, oracle):
    """Seeded random scenarios: block sizes 9..640, targets from one sample to several blocks, totals around the
    boundaries of the rules (todrop * 8 < nsamples, nsamples <= target, current_pos >= total), both strategies, thresholds
    that make some frames 'silent'.  Every block must come out as the oracle's."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        S = int(rng.integers(1, 9))
        n = int(rng.integers(9, 641))
        strategy = int(rng.integers(0, 2))
        thr = float(rng.choice([0.02, 0.2, 0.0005]))
        fc = ms.FlowControlBatch(ctx, S, 640)
        fc.set_config(strategy, thr)
        refs = [oracle.FlowCtl(strategy, thr) for _ in range(S)]
        x = torch.zeros((S, 640), dtype=torch.int16, device="cuda")
        out = torch.zeros((S, 640), dtype=torch.int16, device="cuda")
        olen = torch.zeros(S, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for t in range(14):
            if t in (1, 6, 9):
                drop = rng.integers(0, 3 * n, S).astype(np.uint32) * (rng.random(S) < 0.8)
                total = rng.integers(1, 12 * n, S).astype(np.uint32)
                fc.request_drop(drop, total)
                for s in range(S):
                    if (drop[s] or total[s]) and not (refs[s].c.total_samples > 0 and refs[s].c.target_samples > 0):
                        refs[s].set_target(int(drop[s]), int(total[s]))
            amp = rng.choice([30.0, 3000.0, 12000.0], S)
            blocks = np.stack([np.clip(np.round(rng.normal(0, amp[s], n) + 2000 * np.sin(np.arange(n) * 0.05 * (s + 1))),
                                       -32768, 32767).astype(np.int16) for s in range(S)])
            x[:, :n].copy_(torch.from_numpy(blocks))
            torch.cuda.synchronize()
            fc.process(x, out, olen, length=n)
            ctx.sync()
            got, gl = out.cpu().numpy(), olen.cpu().numpy()
            for s in range(S):
                want = refs[s].process(blocks[s])
                assert gl[s] == want.size, (case, t, s, n, strategy)
                np.testing.assert_array_equal(got[s, : gl[s]], want, err_msg=f"case {case} tick {t} stream {s}")
        fc.close()


def test_g711_randomised_layouts(ctx, oracle):
    """Seeded random (rows, length, stride, per-row counts, base offsets): vector and scalar paths, tails, both laws and
    directions."""
    rng = np.random.default_rng(711)
    for case in range(40):
        rows, n = int(rng.integers(1, 40)), int(rng.integers(1, 700))
        stride = n + int(rng.integers(0, 40))
        off = int(rng.choice([0, 0, 1, 8, 16]))
        law = int(rng.integers(0, 2))
        codes = rng.integers(0, 256, (rows, stride), dtype=np.uint8)
        lens = rng.integers(0, n + 1, rows).astype(np.int32)
        big_c = torch.zeros(rows * stride + 64, dtype=torch.uint8, device="cuda")
        big_p = torch.full((rows * stride + 64,), 777, dtype=torch.int16, device="cuda")
        d_codes = big_c[off: off + rows * stride].view(rows, stride)
        d_pcm = big_p[off: off + rows * stride].view(rows, stride)
        d_codes.copy_(torch.from_numpy(codes))
        d_lens = torch.from_numpy(lens).cuda()
        torch.cuda.synchronize()
        ms.g711_decode(ctx, law, d_codes, d_pcm, length=n, lens=d_lens)
        ctx.sync()
        got = d_pcm.cpu().numpy()
        for r in range(rows):
            np.testing.assert_array_equal(got[r, : lens[r]], oracle.g711_decode(law, codes[r, : lens[r]]), err_msg=f"case {case} row {r}")
            assert (got[r, lens[r]:] == 777).all()
        back = torch.full((rows * stride + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        d_back = back[off: off + rows * stride].view(rows, stride)
        torch.cuda.synchronize()
        ms.g711_encode(ctx, law, d_pcm, d_back, length=n, lens=d_lens)
        ctx.sync()
        enc = d_back.cpu().numpy()
        for r in range(rows):
            np.testing.assert_array_equal(enc[r, : lens[r]], oracle.g711_encode(law, got[r, : lens[r]]), err_msg=f"case {case} row {r}")
            assert (enc[r, lens[r]:] == 0xAB).all()
