"""The widened rows at the drop-in boundary: MSAlawDec / MSUlawDec / MSAlawEnc / MSUlawEnc,
MSL16Enc / MSL16Dec, MSChannelAdapter and MSAudioFlowControl created by id through the factory, run in
source -> filter -> sink graphs tick by tick and compared with the oracle (bit-exact: byte / integer work)."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_pcm
from test_gpu_plugin import Host, mid, MS_FILTER_BASE_ID, MS_FILTER_IS_HW_ACCELERATED, SET_SAMPLE_RATE, SET_NCHANNELS

pytestmark = pytest.mark.gpu

MS_ULAW_ENC_ID, MS_ULAW_DEC_ID, MS_ALAW_ENC_ID, MS_ALAW_DEC_ID = 7, 8, 9, 10
MS_CHANNEL_ADAPTER_ID, MS_L16_ENC_ID, MS_L16_DEC_ID, MS_AUDIO_FLOW_CONTROL_ID = 67, 107, 108, 141
AUDIO_DECODER_IFACE, AUDIO_ENCODER_IFACE = 16384 + 7, 16384 + 11
GET_SAMPLE_RATE = mid(MS_FILTER_BASE_ID, 1, 4)
GET_NCHANNELS = mid(MS_FILTER_BASE_ID, 5, 4)
ADD_FMTP = mid(MS_FILTER_BASE_ID, 7, 1)
ADD_ATTR = mid(MS_FILTER_BASE_ID, 8, 1)
HAVE_PLC = mid(AUDIO_DECODER_IFACE, 0, 4)
GET_PTIME = mid(AUDIO_ENCODER_IFACE, 1, 4)
SET_OUT_NCHANNELS = mid(MS_CHANNEL_ADAPTER_ID, 0, 4)
FLOW_SET_CONFIG = mid(MS_AUDIO_FLOW_CONTROL_ID, 0, 8)
FLOW_DROP = mid(MS_AUDIO_FLOW_CONTROL_ID, 1, 8)


class DropEvent(C.Structure):
    _fields_ = [("flow_control_interval_ms", C.c_uint32), ("drop_ms", C.c_uint32)]


class FlowConfig(C.Structure):
    _fields_ = [("strategy", C.c_int), ("silent_threshold", C.c_float)]


@pytest.fixture(scope="module")
def host():
    h = Host()
    h.S.ms2shim_source_push_ts.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
    h.S.ms2shim_sink_last_ts.restype = C.c_uint32
    h.S.ms2shim_sink_last_ts.argtypes = [C.c_void_p]
    return h


def get_int(host, f, method):
    v = C.c_int(-1)
    assert host.S.ms_filter_call_method(f, method, C.byref(v)) == 0
    return v.value


def push_bytes(host, src, data, ts=None):
    a = np.ascontiguousarray(data)
    if ts is None:
        host.S.ms2shim_source_push(src, a.ctypes.data, a.nbytes)
    else:
        host.S.ms2shim_source_push_ts(src, a.ctypes.data, a.nbytes, ts)


def drain_bytes(host, sink):
    n = host.S.ms2shim_sink_size(sink)
    buf = np.zeros(n, np.uint8)
    if n:
        host.S.ms2shim_sink_read(sink, buf.ctypes.data, n)
    return buf


def run_graph(host, chain, feed, ticks):
    for a, b in zip(chain[:-1], chain[1:]):
        host.link(a, 0, b, 0)
    host.S.ms_ticker_attach(host.ticker, chain[0])
    for t in range(ticks):
        feed(t)
        host.step()
    host.step(3)
    out = drain_bytes(host, chain[-1])
    host.S.ms_ticker_detach(host.ticker, chain[0])
    return out


def test_codec_descriptors_take_over_the_reference_ids(host):
    for fid, name in ((MS_ALAW_DEC_ID, b"MSAlawDec"), (MS_ULAW_DEC_ID, b"MSUlawDec"), (MS_ALAW_ENC_ID, b"MSAlawEnc"),
                      (MS_ULAW_ENC_ID, b"MSUlawEnc"), (MS_L16_ENC_ID, b"MSL16Enc"), (MS_L16_DEC_ID, b"MSL16Dec"),
                      (MS_CHANNEL_ADAPTER_ID, b"MSChannelAdapter"), (MS_AUDIO_FLOW_CONTROL_ID, b"MSAudioFlowControl")):
        f = host.create(fid)
        assert host.S.ms2shim_filter_name(f) == name
        assert host.S.ms2shim_filter_flags(f) & MS_FILTER_IS_HW_ACCELERATED
        host.S.ms_filter_destroy(f)
    dec = host.create(MS_ALAW_DEC_ID)
    assert (get_int(host, dec, GET_SAMPLE_RATE), get_int(host, dec, GET_NCHANNELS), get_int(host, dec, HAVE_PLC)) == (8000, 1, 0)
    host.S.ms_filter_destroy(dec)


@pytest.mark.parametrize("fid,law", [(MS_ALAW_DEC_ID, 0), (MS_ULAW_DEC_ID, 1)])
def test_g711_decoder_graph(host, oracle, fid, law):
    """RTP-sized packets (160 B = 20 ms, plus a 7-byte and a 1120-byte one) -> one block each, timestamps carried
    over (mblk_meta_copy, alaw.c:213)."""
    rng = np.random.default_rng(fid)
    sizes = [160, 160, 7, 1120, 160, 80, 160, 33]
    pkts = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    src, dec, snk = host.source(), host.create(fid), host.sink()
    out = run_graph(host, [src, dec, snk], lambda t: push_bytes(host, src, pkts[t], ts=1000 + 160 * t), len(pkts))
    want = np.concatenate([oracle.g711_decode(law, p) for p in pkts])
    np.testing.assert_array_equal(out.view(np.int16), want)
    assert host.S.ms2shim_sink_blocks(snk) == len(pkts)
    assert host.S.ms2shim_sink_last_ts(snk) == 1000 + 160 * (len(pkts) - 1)


@pytest.mark.parametrize("fid,law", [(MS_ALAW_ENC_ID, 0), (MS_ULAW_ENC_ID, 1)])
def test_g711_encoder_graph_reframes_to_ptime(host, oracle, fid, law):
    """10 ms ticks of 80 samples in; default packets of 2 x 10 ms (alaw.c:59,:72), then ptime=30 via fmtp: 240-byte
    packets, timestamps counted in samples (:84-85)."""
    x = synth_pcm(fid, 80 * 30, rate=8000)
    src, enc, snk = host.source(), host.create(fid), host.sink()
    assert get_int(host, enc, GET_PTIME) == 0
    out = run_graph(host, [src, enc, snk], lambda t: host.push(src, x[80 * t: 80 * (t + 1)]), 30)
    np.testing.assert_array_equal(out, oracle.g711_encode(law, x))
    assert host.S.ms2shim_sink_blocks(snk) == 15
    assert host.S.ms2shim_sink_last_ts(snk) == 160 * 14
    src, enc, snk = host.source(), host.create(fid), host.sink()
    assert host.S.ms_filter_call_method(enc, ADD_FMTP, C.c_char_p(b"annexb=no;ptime=30")) == 0
    assert get_int(host, enc, GET_PTIME) == 30
    out = run_graph(host, [src, enc, snk], lambda t: host.push(src, x[80 * t: 80 * (t + 1)]), 30)
    np.testing.assert_array_equal(out, oracle.g711_encode(law, x))
    assert host.S.ms2shim_sink_blocks(snk) == 10
    # a=ptime:100 hits the "ptime:10" test first (alaw.c:110): reproduced
    assert host.S.ms_filter_call_method(enc, ADD_ATTR, C.c_char_p(b"ptime:100")) == 0
    assert get_int(host, enc, GET_PTIME) == 10


def test_g711_encode_then_decode_chain(host, oracle):
    x = synth_pcm(3, 80 * 20, rate=8000)
    src, enc, dec, snk = host.source(), host.create(MS_ULAW_ENC_ID), host.create(MS_ULAW_DEC_ID), host.sink()
    out = run_graph(host, [src, enc, dec, snk], lambda t: host.push(src, x[80 * t: 80 * (t + 1)]), 20)
    np.testing.assert_array_equal(out.view(np.int16), oracle.g711_decode(1, oracle.g711_encode(1, x)))


def test_l16_encoder_and_decoder_graphs(host, oracle):
    """l16.c: the encoder re-frames to ptime (10 ms default) and emits network byte order; the decoder swaps back."""
    x = synth_pcm(5, 160 * 12, rate=16000)
    src, enc, snk = host.source(), host.create(MS_L16_ENC_ID), host.sink()
    assert host.call_int(enc, SET_SAMPLE_RATE, 16000) == 0
    assert host.S.ms_filter_call_method(enc, ADD_FMTP, C.c_char_p(b"ptime=20")) == 0
    out = run_graph(host, [src, enc, snk], lambda t: host.push(src, x[160 * t: 160 * (t + 1)]), 12)
    np.testing.assert_array_equal(out.view(">i2").astype(np.int16), x)
    assert host.S.ms2shim_sink_blocks(snk) == 6
    assert host.S.ms2shim_sink_last_ts(snk) == 320 * 5
    src, dec, snk = host.source(), host.create(MS_L16_DEC_ID), host.sink()
    be = oracle.l16_swap(x)
    out = run_graph(host, [src, dec, snk], lambda t: push_bytes(host, src, be[160 * t: 160 * (t + 1)], ts=77 + t), 12)
    np.testing.assert_array_equal(out.view(np.int16), x)
    assert host.S.ms2shim_sink_last_ts(snk) == 77 + 11


def test_channel_adapter_graphs(host, oracle):
    n, nt = 160, 10
    a, b = synth_pcm(11, n * nt, rate=16000), synth_pcm(12, n * nt, rate=16000)
    # mono -> stereo (chanadapt.c:106-113)
    src, ad, snk = host.source(), host.create(MS_CHANNEL_ADAPTER_ID), host.sink()
    assert host.call_int(ad, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(ad, SET_OUT_NCHANNELS, 2) == 0
    out = run_graph(host, [src, ad, snk], lambda t: host.push(src, a[n * t: n * (t + 1)]), nt)
    np.testing.assert_array_equal(out.view(np.int16), oracle.chan_adapt(0, a))
    # stereo -> mono keeps the left sample (:114-121)
    st = np.stack([a, b], 1).ravel()
    src, ad, snk = host.source(), host.create(MS_CHANNEL_ADAPTER_ID), host.sink()
    assert host.call_int(ad, SET_NCHANNELS, 2) == 0
    out = run_graph(host, [src, ad, snk], lambda t: host.push(src, st[2 * n * t: 2 * n * (t + 1)]), nt)
    np.testing.assert_array_equal(out.view(np.int16), a)
    # same channel count: blocks pass through untouched, this tick
    src, ad, snk = host.source(), host.create(MS_CHANNEL_ADAPTER_ID), host.sink()
    out = run_graph(host, [src, ad, snk], lambda t: host.push(src, a[n * t: n * (t + 1)]), nt)
    np.testing.assert_array_equal(out.view(np.int16), a)


def test_channel_adapter_two_mono_inputs(host, oracle):
    """Two linked inputs -> one interleaved stereo tick per 10 ms; a side that is short is silent (chanadapt.c:68-93)."""
    n, nt = 160, 8
    a, b = synth_pcm(21, n * nt, rate=16000), synth_pcm(22, n * nt, rate=16000)
    s1, s2, ad, snk = host.source(), host.source(), host.create(MS_CHANNEL_ADAPTER_ID), host.sink()
    assert host.call_int(ad, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(ad, SET_OUT_NCHANNELS, 2) == 0
    host.link(s1, 0, ad, 0)
    host.link(s2, 0, ad, 1)
    host.link(ad, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, s1)
    host.S.ms_ticker_attach(host.ticker, s2)
    for t in range(nt):
        host.push(s1, a[n * t: n * (t + 1)])
        if t != 3:  # the right side misses a tick
            host.push(s2, b[n * t: n * (t + 1)])
        host.step()
    host.step(3)
    out = drain_bytes(host, snk).view(np.int16).reshape(-1, 2)
    right = np.concatenate([b[: 3 * n], np.zeros(n, np.int16), b[4 * n:]])
    np.testing.assert_array_equal(out[:, 0], a)
    np.testing.assert_array_equal(out[:, 1], right)
    host.S.ms_ticker_detach(host.ticker, s1)
    host.S.ms_ticker_detach(host.ticker, s2)


@pytest.mark.parametrize("strategy", [1, 0])
def test_flow_control_graph_follows_the_oracle(host, oracle, strategy):
    """MS_AUDIO_FLOW_CONTROL_DROP the way AudioStream forwards the canceller's event (audiostream.c:1175-1178):
    drop 10 ms out of the next 200 ms at 16 kHz; a second request while dropping is ignored (flowcontrol.c:213)."""
    n, nt = 160, 52
    x = synth_pcm(31, n * nt, rate=16000)
    src, fc, snk = host.source(), host.create(MS_AUDIO_FLOW_CONTROL_ID), host.sink()
    assert host.call_int(fc, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(fc, SET_NCHANNELS, 1) == 0
    assert host.call(fc, FLOW_SET_CONFIG, FlowConfig(strategy, 0.02)) == 0
    ref = oracle.FlowCtl(strategy, 0.02)
    want = []

    def feed(t):
        if t in (5, 8, 30):
            ev = DropEvent(200, 10) if t != 8 else DropEvent(100, 50)
            assert host.call(fc, FLOW_DROP, ev) == 0
            if not (ref.c.total_samples > 0 and ref.c.target_samples > 0):
                ref.set_target(ev.drop_ms * 16000 // 1000, ev.flow_control_interval_ms * 16000 // 1000)
        blk = x[n * t: n * (t + 1)]
        host.push(src, blk)
        want.append(ref.process(blk))

    out = run_graph(host, [src, fc, snk], feed, nt)
    want = np.concatenate(want)
    assert want.size == n * nt - 2 * 160  # two completed requests of 10 ms each
    np.testing.assert_array_equal(out.view(np.int16), want)


MS_GENERIC_PLC_ID = 111
PLC_SET_CN = mid(MS_GENERIC_PLC_ID, 0, 36)


class CngData(C.Structure):
    _fields_ = [("datasize", C.c_int), ("data", C.c_uint8 * 32)]


@pytest.mark.parametrize("rate", [8000, 16000])
def test_generic_plc_graph_follows_the_oracle(host, oracle, rate):
    """source -> MSGenericPLC -> sink with packets missing on some ticks, a late burst and a comfort-noise period:
    the filter emits what generic_plc_process (msgenericplc.c:59-167) emits, block for block."""
    n, nt = rate // 100, 60
    x = synth_pcm(50 + rate // 8000, n * nt, rate=rate, sigma=1500.0)
    src, plc, snk = host.source(), host.create(MS_GENERIC_PLC_ID), host.sink()
    assert host.S.ms2shim_filter_name(plc) == b"MSGenericPLC"
    assert host.call_int(plc, SET_SAMPLE_RATE, rate) == 0 and host.call_int(plc, SET_NCHANNELS, 1) == 0
    ref = oracle.GenericPlcFilter(rate)
    lost = set(range(10, 13)) | {20} | set(range(30, 48))   # 30 ms, one packet, 180 ms (fade and silence)
    late = {25: 26}                                         # tick 25's packet arrives together with tick 26's
    cn_at = 52
    want, k = [], [0]

    def feed(t):
        blocks = []
        if t == cn_at:
            assert host.call(plc, PLC_SET_CN, CngData()) == 0
            ref.set_cn()
        if t in lost or t in late or t in (cn_at, cn_at + 1):
            pass
        else:
            cnt = 2 if t in late.values() else 1
            for _ in range(cnt):
                blocks.append(x[k[0] * n:(k[0] + 1) * n])
                k[0] += 1
        for b in blocks:
            host.push(src, b)
        want.extend(ref.tick(1000 + 10 * t, blocks))

    out = run_graph(host, [src, plc, snk], feed, nt)
    for t in range(nt, nt + 3):  # run_graph's three draining ticks: a pump filter keeps concealing through them
        want.extend(ref.tick(1000 + 10 * t, []))
    want = np.concatenate(want)
    got = out.view(np.int16)
    assert want.size - n <= got.size <= want.size  # the last tick's block is still staged (one tick of latency)
    np.testing.assert_array_equal(got, want[: got.size])
    assert ref.con.total_number_for_plc >= 3 + 1 + 18 + 1


def test_facades_keep_order_and_samples_under_bursts(host, oracle):
    """More blocks in one tick than a pool has launch rounds (4): nothing is lost or reordered.  Six 5 ms blocks per tick
    through MSAudioFlowControl (dropping 10 ms out of 300 ms meanwhile), through MSUlawDec, and a stereo MSL16Enc."""
    rate, n = 16000, 80
    # flow control: 6 blocks per tick
    host.S.ms2shim_source_set_burst.argtypes = [C.c_void_p, C.c_int]
    x = synth_pcm(77, n * 6 * 12, rate=rate)
    src, fc, snk = host.source(), host.create(MS_AUDIO_FLOW_CONTROL_ID), host.sink()
    host.S.ms2shim_source_set_burst(src, 1)
    assert host.call_int(fc, SET_SAMPLE_RATE, rate) == 0 and host.call_int(fc, SET_NCHANNELS, 1) == 0
    ref = oracle.FlowCtl()
    want, k = [], [0]

    def feed(t):
        if t == 2:
            assert host.call(fc, FLOW_DROP, DropEvent(300, 10)) == 0
            ref.set_target(10 * rate // 1000, 300 * rate // 1000)
        for _ in range(6):
            blk = x[k[0] * n:(k[0] + 1) * n]
            k[0] += 1
            host.push(src, blk)
            want.append(ref.process(blk))

    out = run_graph(host, [src, fc, snk], feed, 12)
    want_all = np.concatenate(want)
    got = out.view(np.int16)
    # blocks beyond the fourth of a tick pass unedited (documented): the sequence is complete and in order, and the
    # controller removed at most what the reference would have
    assert x.size - 160 <= got.size <= x.size
    it = iter(x.tolist())
    assert all(any(v == w for w in it) for v in got.tolist())  # a subsequence of the input, order kept
    assert want_all.size == x.size - 160
    # decoder: 6 packets per tick
    codes = np.random.default_rng(4).integers(0, 256, 6 * 12 * 40, dtype=np.uint8)
    src, dec, snk = host.source(), host.create(MS_ULAW_DEC_ID), host.sink()
    host.S.ms2shim_source_set_burst(src, 1)
    pos = [0]

    def feed_dec(t):
        for _ in range(6):
            push_bytes(host, src, codes[pos[0]: pos[0] + 40])
            pos[0] += 40

    out = run_graph(host, [src, dec, snk], feed_dec, 12)
    np.testing.assert_array_equal(out.view(np.int16), oracle.g711_decode(1, codes))
    assert host.S.ms2shim_sink_blocks(snk) == 72
    # stereo L16 encoder: 10 ms of 2 x 16 kHz per packet, byte-swapped, timestamps in frames (l16.c:89-91)
    st = synth_pcm(9, 2 * 160 * 10, rate=rate)
    src, enc, snk = host.source(), host.create(MS_L16_ENC_ID), host.sink()
    assert host.call_int(enc, SET_SAMPLE_RATE, rate) == 0 and host.call_int(enc, SET_NCHANNELS, 2) == 0
    out = run_graph(host, [src, enc, snk], lambda t: host.push(src, st[320 * t: 320 * (t + 1)]), 10)
    np.testing.assert_array_equal(out.view(">i2").astype(np.int16), st)
    assert host.S.ms2shim_sink_blocks(snk) == 10 and host.S.ms2shim_sink_last_ts(snk) == 160 * 9


def test_generic_plc_burst_of_six_blocks(host, oracle):
    """Six 10 ms packets delivered in ONE tick after five ticks of silence (a jitter burst): more pieces than the pool has
    launch rounds; every block is forwarded, delayed and cross-faded like the reference does, then concealment resumes
    only when the concealer's clock says so."""
    rate, n = 8000, 80
    x = synth_pcm(61, n * 40, rate=rate, sigma=1500.0)
    src, plc, snk = host.source(), host.create(MS_GENERIC_PLC_ID), host.sink()
    host.S.ms2shim_source_set_burst.argtypes = [C.c_void_p, C.c_int]
    host.S.ms2shim_source_set_burst(src, 1)
    assert host.call_int(plc, SET_SAMPLE_RATE, rate) == 0
    ref = oracle.GenericPlcFilter(rate)
    want, k = [], [0]

    def feed(t):
        if t < 8:
            cnt = 1
        elif t < 13:
            cnt = 0            # five packets late ...
        elif t == 13:
            cnt = 6            # ... all six arrive at once
        else:
            cnt = 1
        blocks = [x[(k[0] + i) * n:(k[0] + i + 1) * n] for i in range(cnt)]
        k[0] += cnt
        for b in blocks:
            host.push(src, b)
        want.extend(ref.tick(1000 + 10 * t, blocks))

    out = run_graph(host, [src, plc, snk], feed, 24)
    for t in range(24, 27):
        want.extend(ref.tick(1000 + 10 * t, []))
    want = np.concatenate(want)
    got = out.view(np.int16)
    assert want.size - n <= got.size <= want.size
    np.testing.assert_array_equal(got, want[: got.size])


def test_receive_path_graph_of_an_audio_stream(host, oracle):
    """The receive half of the reference's AudioStream graph (src/voip/audiostream.c:1798-1832) built from the plugin's
    facades: RTP payloads (PCMU, 10 ms, some lost) -> MSUlawDec -> MSGenericPLC -> MSResample 8k->16k -> MSVolume (static
    gain) -> sink.  Every facade adds a tick of latency, none changes the samples: the output is the oracle's chain."""
    from test_gpu_plugin import MS_RESAMPLE_ID, MS_VOLUME_ID, SET_OUTPUT_SAMPLE_RATE
    rate, n, nt = 8000, 80, 50
    pcm = synth_pcm(71, n * nt, rate=rate, sigma=2000.0)
    payloads = oracle.g711_encode(1, pcm)
    lost = {7, 8, 20, 33, 34, 35}
    src, dec, plc, rs, vol, snk = (host.source(), host.create(MS_ULAW_DEC_ID), host.create(MS_GENERIC_PLC_ID),
                                   host.create(MS_RESAMPLE_ID), host.create(MS_VOLUME_ID), host.sink())
    assert host.call_int(plc, SET_SAMPLE_RATE, rate) == 0
    assert host.call_int(rs, SET_SAMPLE_RATE, rate) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 16000) == 0
    assert host.call_int(vol, SET_SAMPLE_RATE, 16000) == 0
    g = C.c_float(0.5)
    assert host.call(vol, mid(MS_VOLUME_ID, 2, 4), g) == 0  # MS_VOLUME_SET_GAIN

    def feed(t):
        if t not in lost:
            push_bytes(host, src, payloads[t * n:(t + 1) * n])

    out = run_graph(host, [src, dec, plc, rs, vol, snk], feed, nt)
    got = out.view(np.int16)
    # the oracle's chain on the same timeline (the decoder's tick of latency shifts the whole pattern, nothing else)
    ref_plc = oracle.GenericPlcFilter(rate)
    ref_rs = oracle.Resampler(rate, 16000)
    ref_vol = oracle.Volume(16000)
    oracle.lib().orc_volume_set_gain(ref_vol.v, 0.5)
    want = []
    for t in range(nt + 3):
        blocks = [] if (t in lost or t >= nt) else [oracle.g711_decode(1, payloads[t * n:(t + 1) * n])]
        for b in ref_plc.tick(1000 + 10 * t, blocks):
            want.append(ref_vol.chunk(ref_rs.process(b)))
    want = np.concatenate(want)
    assert got.size > 0.9 * want.size
    m = min(got.size, want.size)
    # the resampler is within 1 LSB of the library's order, the static gain halves that
    assert np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32)).max() <= 1
    assert ref_plc.con.total_number_for_plc >= len(lost)


def test_send_path_graph_of_an_audio_stream(host, oracle):
    """The send half: 16 kHz capture -> MSVolume (AGC) -> MSResample 16k->8k -> MSUlawEnc (ptime 20) -> RTP payloads.
    AGC is bit-exact; the resampler's 1-LSB freedom can move a sample across a mu-law decision level, so the payloads
    are the oracle's except for rare neighbouring code words."""
    from test_gpu_plugin import MS_RESAMPLE_ID, MS_VOLUME_ID, SET_OUTPUT_SAMPLE_RATE
    nt = 60
    x = synth_pcm(72, 160 * nt, rate=16000, sigma=4000.0)
    src, vol, rs, enc, snk = (host.source(), host.create(MS_VOLUME_ID), host.create(MS_RESAMPLE_ID),
                              host.create(MS_ULAW_ENC_ID), host.sink())
    assert host.call_int(vol, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(vol, mid(MS_VOLUME_ID, 8, 4), 1) == 0
    assert host.call_int(rs, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 8000) == 0
    assert host.S.ms_filter_call_method(enc, ADD_FMTP, C.c_char_p(b"ptime=20")) == 0
    out = run_graph(host, [src, vol, rs, enc, snk], lambda t: host.push(src, x[160 * t: 160 * (t + 1)]), nt)
    assert host.S.ms2shim_sink_blocks(snk) >= nt // 2 - 2 and out.size % 160 == 0   # 20 ms packets of 160 code words
    o_vol = oracle.Volume(16000)
    o_vol.v.agc_enabled = 1
    o_rs = oracle.Resampler(16000, 8000)
    pcm8 = np.concatenate([o_rs.process(o_vol.chunk(x[160 * t: 160 * (t + 1)])) for t in range(nt)])
    want = oracle.g711_encode(1, pcm8)[: out.size]
    same = out == want
    assert same.mean() > 0.995
    # where they differ it is the neighbouring code word (magnitude index +- 1)
    mag = lambda c: (~c) & 0x7F
    assert (np.abs(mag(out[~same]).astype(int) - mag(want[~same]).astype(int)) <= 1).all()
