"""The oracle's codec / adapter / flow-control restatements (oracle/g711.c), on the CPU.

G.711 is the one part of the path whose reference source compiles here from the file where it lies
(src/audiofilters/g711.c includes only its own header): oracle/build_ref.sh builds it into oracle/_ref and the
restatement is compared with it over its whole domain.  The ITU-T G.711 known answers below hold either way."""
import numpy as np
import pytest


def test_g711_matches_the_reference_build(oracle):
    R = oracle.g711_ref()
    if R is None:
        pytest.skip("oracle/_ref/libg711_ref.so not built (no /root/reference on this machine)")
    pcm = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    np.testing.assert_array_equal(oracle.g711_encode(oracle.LAW_PCMA, pcm), np.array([R.Snack_Lin2Alaw(int(v)) for v in pcm], np.uint8))
    np.testing.assert_array_equal(oracle.g711_encode(oracle.LAW_PCMU, pcm), np.array([R.Snack_Lin2Mulaw(int(v)) for v in pcm], np.uint8))
    codes = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(oracle.g711_decode(oracle.LAW_PCMA, codes), np.array([R.Snack_Alaw2Lin(int(c)) for c in codes], np.int16))
    np.testing.assert_array_equal(oracle.g711_decode(oracle.LAW_PCMU, codes), np.array([R.Snack_Mulaw2Lin(int(c)) for c in codes], np.int16))


def test_g711_known_answers(oracle):
    """ITU-T G.711 tables 1a/2a (first / last decision levels) scaled to 16 bits."""
    A, U = oracle.LAW_PCMA, oracle.LAW_PCMU
    dec = lambda law, c: int(oracle.g711_decode(law, np.array([c], np.uint8))[0])
    enc = lambda law, v: int(oracle.g711_encode(law, np.array([v], np.int16))[0])
    # A-law: even bits inverted, 0xD5 / 0x55 are +/- the smallest interval (centre 1 of 4096 -> 8 of 32768)
    assert (dec(A, 0xD5), dec(A, 0x55)) == (8, -8)
    assert (dec(A, 0xAA), dec(A, 0x2A)) == (32256, -32256)  # largest interval: (0x1F8 << 6)
    assert (enc(A, 0), enc(A, -1), enc(A, 32767), enc(A, -32768)) == (0xD5, 0x55, 0xAA, 0x2A)
    # mu-law: complemented code words, 0xFF / 0x7F are the two zeros, 0x80 / 0x00 the extremes (8031 * 4)
    assert (dec(U, 0xFF), dec(U, 0x7F), dec(U, 0x80), dec(U, 0x00)) == (0, 0, 32124, -32124)
    assert (enc(U, 0), enc(U, -1), enc(U, 32767), enc(U, -32768)) == (0xFF, 0x7E, 0x80, 0x00)  # -1 >> 2 is -1: magnitude 1
    # both laws are monotonic in the magnitude and odd-symmetric in the decoded value
    for law in (A, U):
        codes = np.arange(256, dtype=np.uint8)
        lin = oracle.g711_decode(law, codes).astype(np.int32)
        np.testing.assert_array_equal(lin[:128], -lin[128:])
        np.testing.assert_array_equal(oracle.g711_decode(law, oracle.g711_encode(law, lin.astype(np.int16))), lin)  # fixed points
        recoded = oracle.g711_encode(law, lin.astype(np.int16))
        np.testing.assert_array_equal(recoded, codes if law == A else np.where(codes == 0x7F, 0xFF, codes))  # mu-law has two zeros
        pcm = np.arange(-32768, 32768, 7, dtype=np.int32).astype(np.int16)
        back = oracle.g711_decode(law, oracle.g711_encode(law, pcm)).astype(np.int32)
        assert (np.diff(back) >= 0).all()
        # companding error stays within half a step of the segment: < 1/16 of the magnitude + the smallest step
        assert (np.abs(back - pcm) <= np.abs(pcm.astype(np.int32)) / 16 + 36).all()


def test_l16_and_channel_adapter(oracle):
    x = np.array([0x0102, -2, 0x7FFF, -32768], np.int16)
    np.testing.assert_array_equal(oracle.l16_swap(x).view(np.uint8), x.view(np.uint8).reshape(-1, 2)[:, ::-1].ravel())
    np.testing.assert_array_equal(oracle.l16_swap(oracle.l16_swap(x)), x)
    a, b = np.array([1, 2, 3], np.int16), np.array([-1, -2, -3], np.int16)
    np.testing.assert_array_equal(oracle.chan_adapt(0, a), [1, 1, 2, 2, 3, 3])
    np.testing.assert_array_equal(oracle.chan_adapt(1, np.array([1, 9, 2, 9, 3, 9], np.int16)), a)  # the left sample is kept
    np.testing.assert_array_equal(oracle.chan_adapt(2, a, b), [1, -1, 2, -2, 3, -3])
    np.testing.assert_array_equal(oracle.chan_adapt(2, a, None), [1, 0, 2, 0, 3, 0])


def loud(rng, n):
    return np.clip(np.round(8000 * np.sin(np.arange(n) * 0.09) + rng.normal(0, 400, n)), -32768, 32767).astype(np.int16)


def test_flow_controller_soft_strategy(oracle):
    """flowcontrol.c:107-152: over `total` samples exactly `target` are removed, spread in proportion, each removal
    at the smoothest spot; afterwards blocks pass untouched."""
    rng = np.random.default_rng(1)
    fc = oracle.FlowCtl()
    first = loud(rng, 160)
    np.testing.assert_array_equal(fc.process(first), first)  # not armed: untouched
    fc.set_target(40, 1600)
    removed = 0
    for k in range(10):
        blk = loud(rng, 160)
        out = fc.process(blk)
        removed += 160 - out.size
        assert 160 - out.size == 4  # 40 * 160k / 1600 - dropped so far
        # what is left is a subsequence of the block
        it = iter(blk.tolist())
        assert all(any(v == w for w in it) for v in out.tolist())
    assert removed == 40 and fc.c.target_samples == 0
    again = loud(rng, 160)
    np.testing.assert_array_equal(fc.process(again), again)


def test_flow_controller_deletes_the_last_smoothest_middle_sample(oracle):
    fc = oracle.FlowCtl()
    fc.set_target(1, 16)
    blk = np.array([0, 100, 300, 301, 302, 500, 900, 901, 902, 1500, 0, 9, 50, 99, 200, 7], np.int16)
    out = fc.process(blk)
    # two triples tie at |d|+|d| = 2: (300,301,302) and (900,901,902); the <= comparison keeps the LAST: 901 goes
    np.testing.assert_array_equal(out, np.delete(blk, 7))


def test_flow_controller_silent_and_oversized_requests(oracle):
    rng = np.random.default_rng(2)
    fc = oracle.FlowCtl()
    fc.set_target(320, 1600)
    quiet = rng.integers(-30, 31, 160).astype(np.int16)  # power << 0.02: dropped whole (:127-133)
    assert fc.process(quiet).size == 0 and fc.c.current_dropped == 160
    fc = oracle.FlowCtl()
    fc.set_target(100, 480)  # 33 samples of the first 160: more than an eighth -> the whole frame (:137-142)
    assert fc.process(loud(rng, 160)).size == 0 and fc.c.current_dropped == 160
    basic = oracle.FlowCtl(strategy=0)
    basic.set_target(320, 1600)
    sizes = [basic.process(loud(rng, 160)).size for _ in range(4)]
    assert sizes == [0, 0, 160, 160]  # basic: whole blocks until the target is reached (:115-121)


def _voiced(seed, n, rate):
    t = np.arange(n)
    x = 5000 * np.sin(2 * np.pi * (120 + seed) * t / rate) + 2000 * np.sin(2 * np.pi * 3 * (120 + seed) * t / rate + 1)
    return np.round(x).astype(np.int16)


@pytest.mark.parametrize("rate", [8000, 16000, 48000])
def test_plc_clean_stream_is_a_pure_delay(oracle, rate):
    """Without losses MSGenericPLC only delays the stream by TRANSITION_DELAY = 5 ms (genericplc.c:212-231)."""
    n, T = rate // 100, rate * 5 // 1000
    x = _voiced(1, n * 20, rate)
    p = oracle.Plc(rate)
    assert p.info()["nb"] == rate // 20
    y = np.concatenate([p.received(x[k * n:(k + 1) * n]) for k in range(20)])
    np.testing.assert_array_equal(y[T:], x[:-T])
    assert (y[:T] == 0).all()


def test_plc_concealment_shape(oracle):
    """First loss: the generated signal continues the delayed stream without a jump, keeps roughly the level, fades
    out between 100 and 150 ms and is silent afterwards (genericplc.c:187-203); the first good block cross-fades."""
    rate, n = 8000, 80
    x = _voiced(2, n * 60, rate)
    p = oracle.Plc(rate)
    heard = [p.received(x[k * n:(k + 1) * n]) for k in range(12)]
    lost = [p.conceal(n) for _ in range(20)]
    y = np.concatenate(heard + lost).astype(np.int32)
    jump = np.abs(np.diff(y))[12 * n - 3: 12 * n + 3].max()
    assert jump < 2 * np.abs(np.diff(np.concatenate(heard).astype(np.int32))).max()
    lvl = [np.abs(b).max() for b in lost]
    assert lvl[0] > 2000 and lvl[9] > 1000           # 0 .. 100 ms: full level x 0.85 per regeneration
    assert lvl[13] < max(lvl[:10]) / 2 and lvl[14] < max(lvl[:10]) / 10  # fading out from 100 ms on ...
    assert all(v == 0 for v in lvl[15:])                                  # ... silence from 150 ms on
    assert p.info()["used"] == 20 * n
    back = p.received(x[32 * n:33 * n])
    assert p.info()["used"] == 0 and back.shape == (n,)


def test_plc_filter_concealer_timing(oracle):
    """generic_plc_process + MSConcealerContext (mscommon.c:328-366): concealment starts on the first tick whose packet
    did not come, one block per tick, and stops when audio is back; blocks arriving in a burst are all forwarded."""
    rate, n = 8000, 80
    x = _voiced(3, n * 40, rate)
    f = oracle.GenericPlcFilter(rate)
    out_sizes = []
    k = 0
    for t in range(30):
        now = 1000 + 10 * t
        if 10 <= t < 14:
            blocks = []                      # four packets lost
        elif t == 20:
            blocks = []                      # late ...
        elif t == 21:
            blocks = [x[k * n:(k + 1) * n], x[(k + 1) * n:(k + 2) * n]]  # ... then both at once
            k += 2
        else:
            blocks = [x[k * n:(k + 1) * n]]
            k += 1
        out = f.tick(now, blocks)
        out_sizes.append([b.size for b in out])
    assert out_sizes[9] == [n] and out_sizes[10:14] == [[n]] * 4 and out_sizes[14] == [n]
    assert out_sizes[20] == [n]        # concealed
    assert out_sizes[21] == [n, n]     # the burst: both forwarded, no concealment on top (sample_time ran ahead)
    assert f.con.total_number_for_plc == 5
