"""Drop-in boundary on the GPU: the plugin library libmsmi355xfilters.so is loaded through a
factory the way src/base/msfactory.c:531-586 loads plugins, its descriptors take over the
reference's MS_*_ID (registration prepends, lookup is first-match), and graphs of
source -> filter -> sink run tick by tick like the reference's testers build them
(tester/mediastreamer2_basic_audio_tester.c, tester/mediastreamer2_aec3_tester.c).
Outputs are compared with the CPU oracle; the only intended difference is the one-tick
pipeline delay of the batched filters."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import synth_pcm

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mediastreamer2_amd")

MS_FILTER_BASE_ID, MS_SPEEX_EC_ID, MS_RESAMPLE_ID, MS_VOLUME_ID, MS_EQUALIZER_ID, MS_AUDIO_MIXER_ID = 2, 28, 41, 43, 61, 68
EC_IFACE = 16384 + 4
MS_FILTER_IS_PUMP, MS_FILTER_IS_HW_ACCELERATED = 1, 2


def mid(fid, idx, argsize):
    return ((fid & 0xFFFF) << 16) | (idx << 8) | (argsize & 0xFF)


SET_SAMPLE_RATE = mid(MS_FILTER_BASE_ID, 0, 4)
SET_OUTPUT_SAMPLE_RATE = mid(MS_FILTER_BASE_ID, 13, 4)
SET_NCHANNELS = mid(MS_FILTER_BASE_ID, 6, 4)


class MixerCtl(C.Structure):
    _fields_ = [("pin", C.c_int), ("param", C.c_float)]


class EqGain(C.Structure):
    _fields_ = [("frequency", C.c_float), ("gain", C.c_float), ("width", C.c_float)]


class Host:
    """ctypes view of the shim runtime (tests/host/ms2shim.c)."""

    def __init__(self):
        import torch  # noqa: F401  (one HIP runtime per process, see mediastreamer2_amd/_lib.py)
        self.S = C.CDLL(os.path.join(ROOT, "tests", "host", "libms2shim.so"), mode=C.RTLD_GLOBAL)
        S = self.S
        vp = C.c_void_p
        S.ms_factory_new.restype = vp
        S.ms_factory_create_filter.restype = vp
        S.ms_factory_create_filter.argtypes = [vp, C.c_int]
        S.ms_factory_load_plugin.argtypes = [vp, C.c_char_p]
        S.ms_factory_lookup_filter_by_id.restype = vp
        S.ms_factory_lookup_filter_by_id.argtypes = [vp, C.c_int]
        S.ms2shim_register_test_filters.argtypes = [vp]
        S.ms2shim_new_source.restype = vp
        S.ms2shim_new_source.argtypes = [vp]
        S.ms2shim_new_sink.restype = vp
        S.ms2shim_new_sink.argtypes = [vp]
        S.ms2shim_source_push.argtypes = [vp, vp, C.c_size_t]
        S.ms2shim_sink_read.restype = C.c_size_t
        S.ms2shim_sink_read.argtypes = [vp, vp, C.c_size_t]
        S.ms2shim_sink_size.restype = C.c_size_t
        S.ms2shim_sink_size.argtypes = [vp]
        S.ms2shim_sink_blocks.argtypes = [vp]
        S.ms2shim_filter_name.restype = C.c_char_p
        S.ms2shim_filter_name.argtypes = [vp]
        S.ms2shim_filter_flags.restype = C.c_uint
        S.ms2shim_filter_flags.argtypes = [vp]
        S.ms_filter_link.argtypes = [vp, C.c_int, vp, C.c_int]
        S.ms_filter_call_method.argtypes = [vp, C.c_uint, vp]
        S.ms_filter_destroy.argtypes = [vp]
        S.ms_ticker_new.restype = vp
        S.ms_ticker_attach.argtypes = [vp, vp]
        S.ms_ticker_detach.argtypes = [vp, vp]
        S.ms_ticker_step.argtypes = [vp]
        S.ms_ticker_destroy.argtypes = [vp]
        self.fac = S.ms_factory_new()
        S.ms2shim_register_test_filters(self.fac)
        assert S.ms_factory_load_plugin(self.fac, os.path.join(PKG, "libmsmi355xfilters.so").encode()) == 0
        self.ticker = S.ms_ticker_new()

    def create(self, fid):
        f = self.S.ms_factory_create_filter(self.fac, fid)
        assert f
        return f

    def source(self):
        return self.S.ms2shim_new_source(self.fac)

    def sink(self):
        return self.S.ms2shim_new_sink(self.fac)

    def link(self, a, pa, b, pb):
        assert self.S.ms_filter_link(a, pa, b, pb) == 0

    def call_int(self, f, method, val):
        v = C.c_int(val)
        return self.S.ms_filter_call_method(f, method, C.byref(v))

    def call(self, f, method, obj):
        return self.S.ms_filter_call_method(f, method, C.byref(obj))

    def push(self, src, samples):
        a = np.ascontiguousarray(samples, np.int16)
        self.S.ms2shim_source_push(src, a.ctypes.data, a.nbytes)

    def drain(self, sink):
        n = self.S.ms2shim_sink_size(sink)
        buf = np.zeros(n // 2, np.int16)
        if n:
            self.S.ms2shim_sink_read(sink, buf.ctypes.data, n)
        return buf

    def step(self, n=1):
        for _ in range(n):
            self.S.ms_ticker_step(self.ticker)


@pytest.fixture(scope="module")
def host():
    return Host()


def test_plugin_overrides_reference_ids(host):
    """msfactory.c:281 + :440-450: the plugin's descriptors win the lookup by id."""
    for fid, name, pump in ((MS_RESAMPLE_ID, b"MSResample", 0), (MS_VOLUME_ID, b"MSVolume", 0),
                            (MS_EQUALIZER_ID, b"MSEqualizer", 0), (MS_AUDIO_MIXER_ID, b"MSAudioMixer", 1),
                            (MS_SPEEX_EC_ID, b"MSSpeexEC", 0)):
        f = host.create(fid)
        assert host.S.ms2shim_filter_name(f) == name
        flags = host.S.ms2shim_filter_flags(f)
        assert flags & MS_FILTER_IS_HW_ACCELERATED
        assert bool(flags & MS_FILTER_IS_PUMP) == bool(pump)
        host.S.ms_filter_destroy(f)


def test_resample_graph_matches_oracle(host, oracle):
    """source(16 kHz, 160 samples/tick) -> MSResample -> sink, configured through the reference's
    methods (msresample.c:229-233)."""
    src, rs, snk = host.source(), host.create(MS_RESAMPLE_ID), host.sink()
    assert host.call_int(rs, SET_SAMPLE_RATE, 16000) == 0
    assert host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000) == 0
    host.link(src, 0, rs, 0)
    host.link(rs, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    nt = 25
    x = synth_pcm(1, 160 * nt, rate=16000)
    for t in range(nt):
        host.push(src, x[t * 160:(t + 1) * 160])
    host.step(nt + 2)
    got = host.drain(snk)
    o = oracle.Resampler(16000, 48000)
    ref = np.concatenate([o.process(x[t * 160:(t + 1) * 160]) for t in range(nt)])
    assert len(got) == len(ref) == 480 * nt
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1
    assert host.S.ms2shim_sink_blocks(snk) == nt
    host.S.ms_ticker_detach(host.ticker, src)


SET_OUTPUT_NCHANNELS = mid(MS_FILTER_BASE_ID, 28, 4)


def test_resample_graph_stereo_and_channel_adapt(host, oracle):
    """Interleaved stereo in (speex_resampler_process_interleaved_int, msresample.c:160-161): each channel is its own
    stream; and the channel adaptation of :87-100 -- mono resampled then copied to both output channels."""
    nt, n = 12, 80
    L, R = synth_pcm(7, n * nt, rate=8000), synth_pcm(8, n * nt, rate=8000)
    # (a) stereo -> stereo
    src, rs, snk = host.source(), host.create(MS_RESAMPLE_ID), host.sink()
    assert host.call_int(rs, SET_SAMPLE_RATE, 8000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000) == 0
    assert host.call_int(rs, SET_NCHANNELS, 2) == 0 and host.call_int(rs, SET_OUTPUT_NCHANNELS, 2) == 0
    host.link(src, 0, rs, 0)
    host.link(rs, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    inter = np.stack([L, R], 1).ravel()
    for t in range(nt):
        host.push(src, inter[t * 2 * n:(t + 1) * 2 * n])
    host.step(nt + 2)
    got = host.drain(snk).reshape(-1, 2)
    oL, oR = oracle.Resampler(8000, 48000), oracle.Resampler(8000, 48000)
    refL = np.concatenate([oL.process(L[t * n:(t + 1) * n]) for t in range(nt)])
    refR = np.concatenate([oR.process(R[t * n:(t + 1) * n]) for t in range(nt)])
    assert got.shape[0] == len(refL) == 480 * nt
    assert np.abs(got[:, 0].astype(int) - refL).max() <= 1 and np.abs(got[:, 1].astype(int) - refR).max() <= 1
    host.S.ms_ticker_detach(host.ticker, src)
    # (b) mono -> stereo: first channel duplicated
    src, rs, snk = host.source(), host.create(MS_RESAMPLE_ID), host.sink()
    assert host.call_int(rs, SET_SAMPLE_RATE, 8000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000) == 0
    assert host.call_int(rs, SET_OUTPUT_NCHANNELS, 2) == 0
    host.link(src, 0, rs, 0)
    host.link(rs, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    for t in range(nt):
        host.push(src, L[t * n:(t + 1) * n])
    host.step(nt + 2)
    got = host.drain(snk).reshape(-1, 2)
    assert got.shape[0] == 480 * nt
    np.testing.assert_array_equal(got[:, 0], got[:, 1])
    assert np.abs(got[:, 0].astype(int) - refL).max() <= 1
    host.S.ms_ticker_detach(host.ticker, src)


def test_volume_agc_graph_bit_exact(host, oracle):
    src, vol, snk = host.source(), host.create(MS_VOLUME_ID), host.sink()
    assert host.call_int(vol, SET_SAMPLE_RATE, 48000) == 0
    assert host.call_int(vol, mid(MS_VOLUME_ID, 8, 4), 1) == 0  # MS_VOLUME_ENABLE_AGC
    g = C.c_float(0.8)
    assert host.call(vol, mid(MS_VOLUME_ID, 2, 4), g) == 0       # MS_VOLUME_SET_GAIN
    host.link(src, 0, vol, 0)
    host.link(vol, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    nt = 20
    x = synth_pcm(2, 480 * nt, sigma=5000.0)
    for t in range(nt):
        host.push(src, x[t * 480:(t + 1) * 480])
    host.step(nt + 2)
    got = host.drain(snk)
    o = oracle.Volume(48000)
    o.v.agc_enabled = 1
    oracle.lib().orc_volume_set_gain(o.v, 0.8)
    ref = np.concatenate([o.chunk(x[t * 480:(t + 1) * 480]) for t in range(nt)])
    np.testing.assert_array_equal(got, ref)
    lin = C.c_float()
    assert host.call(vol, mid(MS_VOLUME_ID, 1, 4), lin) == 0      # MS_VOLUME_GET_LINEAR
    assert np.float32(lin.value) == np.float32(o.v.energy)
    host.S.ms_ticker_detach(host.ticker, src)


def test_equalizer_graph_bit_exact(host, oracle):
    src, eq, snk = host.source(), host.create(MS_EQUALIZER_ID), host.sink()
    assert host.call_int(eq, SET_SAMPLE_RATE, 16000) == 0           # rate first, gains after
    assert host.call(eq, mid(MS_EQUALIZER_ID, 0, 12), EqGain(1000.0, 2.0, 500.0)) == 0
    n = C.c_int()
    assert host.call(eq, mid(MS_EQUALIZER_ID, 4, 4), n) == 0 and n.value == 128
    host.link(src, 0, eq, 0)
    host.link(eq, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    nt = 12
    x = synth_pcm(3, 160 * nt, sigma=2500.0, rate=16000)
    for t in range(nt):
        host.push(src, x[t * 160:(t + 1) * 160])
    host.step(nt + 2)
    got = host.drain(snk)
    o = oracle.Equalizer(16000)
    o.set_gain(1000, 2.0, 500)
    ref = np.concatenate([o.run(x[t * 160:(t + 1) * 160]) for t in range(nt)])
    np.testing.assert_array_equal(got, ref)
    host.S.ms_ticker_detach(host.ticker, src)


def test_conference_mixer_graph_bit_exact(host, oracle):
    """4 members on pins 0..3 of one MSAudioMixer in conference mode (audioconference.c:67-92)."""
    mx = host.create(MS_AUDIO_MIXER_ID)
    assert host.call_int(mx, SET_SAMPLE_RATE, 16000) == 0
    assert host.call_int(mx, SET_NCHANNELS, 1) == 0
    assert host.call_int(mx, mid(MS_AUDIO_MIXER_ID, 2, 4), 1) == 0       # ENABLE_CONFERENCE_MODE
    ctl = MixerCtl(2, 0.5)
    assert host.call(mx, mid(MS_AUDIO_MIXER_ID, 0, 8), ctl) == 0          # SET_INPUT_GAIN pin 2
    bad = MixerCtl(77, 1.0)
    assert host.call(mx, mid(MS_AUDIO_MIXER_ID, 0, 8), bad) == -1         # invalid pin -> -1 (audiomixer.c:375-378)
    nm = 4
    srcs, snks = [host.source() for _ in range(nm)], [host.sink() for _ in range(nm)]
    for i in range(nm):
        host.link(srcs[i], 0, mx, i)
        host.link(mx, i, snks[i], 0)
    host.S.ms_ticker_attach(host.ticker, mx)
    nt, ns = 15, 160
    x = np.stack([synth_pcm(10 + i, ns * nt, sigma=9000.0, rate=16000) for i in range(nm)])
    for t in range(nt):
        for i in range(nm):
            host.push(srcs[i], x[i, t * ns:(t + 1) * ns])
    host.step(nt + 2)
    gain = np.array([1, 1, 0.5, 1], np.float32)
    ref = np.concatenate([oracle.mixer_tick(x[:, t * ns:(t + 1) * ns], gain=gain)[0] for t in range(nt)], axis=1)
    for i in range(nm):
        got = host.drain(snks[i])
        # ALWAYS_STREAMOUT: the pump also mixes the (silent) ticks after the sources ran dry
        np.testing.assert_array_equal(got[:ns * nt], ref[i])
        assert (got[ns * nt:] == 0).all()
    host.S.ms_ticker_detach(host.ticker, mx)


@pytest.mark.parametrize("rate,F", [(16000, 128), (8000, 64)])
def test_speex_ec_graph_matches_oracle(host, oracle, rate, F):
    """ref + mic sources -> MSSpeexEC -> sinks, tail 128 ms: the framing of speexec.c:223-305 (128-sample frames
    out of 160-sample ticks at 16 kHz, 64 out of 80 at 8 kHz -- the filter's default rate --, zero injection while
    the reference is short) plus the canceller and post-filter, compared with the same framing driven through
    the oracle."""
    ec = host.create(MS_SPEEX_EC_ID)
    assert host.call_int(ec, SET_SAMPLE_RATE, rate) == 0
    assert host.call_int(ec, mid(EC_IFACE, 2, 4), 128) == 0   # SET_TAIL_LENGTH
    assert host.call_int(ec, mid(EC_IFACE, 0, 4), 0) == 0     # SET_DELAY
    d = C.c_int(-1)
    assert host.call(ec, mid(EC_IFACE, 7, 4), d) == 0 and d.value == 0
    s_ref, s_mic, k_ref, k_mic = host.source(), host.source(), host.sink(), host.sink()
    host.link(s_ref, 0, ec, 0)
    host.link(s_mic, 0, ec, 1)
    host.link(ec, 0, k_ref, 0)
    host.link(ec, 1, k_mic, 0)
    host.S.ms_ticker_attach(host.ticker, ec)
    nt, ns = 40, rate // 100
    rng = np.random.default_rng(5)
    far = np.clip(np.round(rng.normal(0, 3000, ns * nt)), -32767, 32767).astype(np.int16)
    ir = rng.normal(0, 1, 48) * np.exp(-np.arange(48) / 10.0)
    mic = np.clip(np.round(0.4 * np.convolve(far.astype(float), ir)[:ns * nt] + rng.normal(0, 50, ns * nt)),
                  -32767, 32767).astype(np.int16)
    for t in range(nt):
        host.push(s_ref, far[t * ns:(t + 1) * ns])
        host.push(s_mic, mic[t * ns:(t + 1) * ns])
    host.step(nt + 3)
    got = host.drain(k_mic)
    ref_out = host.drain(k_ref)
    # oracle-side restatement of the same framing
    e = oracle.Echo(F, 128 * rate // 1000, rate)
    pp = oracle.Preproc(F, rate, e)
    echo_fifo, dref_fifo = np.zeros(0, np.int16), np.zeros(0, np.int16)
    started, outs = False, []
    for t in range(nt):
        if started:
            dref_fifo = np.concatenate([dref_fifo, far[t * ns:(t + 1) * ns]])
        echo_fifo = np.concatenate([echo_fifo, mic[t * ns:(t + 1) * ns]])
        while len(echo_fifo) >= F:
            fr, echo_fifo = echo_fifo[:F], echo_fifo[F:]
            started = True
            if len(dref_fifo) < F:
                dref_fifo = np.concatenate([dref_fifo, np.zeros(F, np.int16)])
            r, dref_fifo = dref_fifo[:F], dref_fifo[F:]
            outs.append(pp.run(e.cancel(fr, r)))
    ref = np.concatenate(outs)
    assert len(got) == len(ref)
    d = got.astype(np.float64) - ref.astype(np.float64)
    assert np.sqrt(np.mean(d ** 2)) / 32768.0 <= 1e-4
    assert len(ref_out) == len(ref)   # one reference frame goes to the speaker per processed frame
    host.S.ms_ticker_detach(host.ticker, ec)


def test_two_ticker_threads_run_concurrently(host, oracle):
    """SURVEY 8(b) threading: different tickers run different filter instances on different threads; the plugin
    keeps one set of pools per ticker and serialises the device context.  Two ticker threads step their own
    resample + volume graphs at the same time; every stream must still match the oracle."""
    import threading
    S = host.S
    tickers = [host.ticker, S.ms_ticker_new()]
    nper, nt = 6, 30
    graphs = []
    for ti, tk in enumerate(tickers):
        for k in range(nper):
            src, rs, vol, snk = host.source(), host.create(MS_RESAMPLE_ID), host.create(MS_VOLUME_ID), host.sink()
            assert host.call_int(rs, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000) == 0
            assert host.call_int(vol, SET_SAMPLE_RATE, 48000) == 0
            g = C.c_float(0.5)
            assert S.ms_filter_call_method(vol, mid(MS_VOLUME_ID, 2, 4), C.byref(g)) == 0   # MS_VOLUME_SET_GAIN
            host.link(src, 0, rs, 0)
            host.link(rs, 0, vol, 0)
            host.link(vol, 0, snk, 0)
            assert S.ms_ticker_attach(tk, src) == 0
            x = synth_pcm(100 * ti + k, 160 * nt, rate=16000)
            for t in range(nt):
                host.push(src, x[t * 160:(t + 1) * 160])
            graphs.append((tk, src, snk, x))
    errs = []

    def run(tk):
        try:
            for _ in range(nt + 4):
                S.ms_ticker_step(tk)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=(tk,)) for tk in tickers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    for tk, src, snk, x in graphs:
        got = host.drain(snk)
        o, v = oracle.Resampler(16000, 48000), oracle.Volume(48000)
        oracle.lib().orc_volume_set_gain(v.v, 0.5)
        ref = np.concatenate([v.chunk(o.process(x[t * 160:(t + 1) * 160])) for t in range(nt)])
        assert len(got) == len(ref)
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1
        S.ms_ticker_detach(tk, src)


def test_mixer_bypass_and_contributor_timeout(host, oracle):
    # audiomixer.c:244-286: with a single contributing pin the mixer forwards its blocks untouched to the other outputs
    # (not to its own in conference mode) -- in this facade ONE TICK LATER, like mixed blocks, so that its latency does not
    # jump when the mode changes (mixer.inl, MixerState::held).  A linked pin counts as a contributor while it has data
    # or had some -- or was first looked at (the reference's quirk, :257-259) -- less than 1000 ms ago; two
    # contributors bring the mixed path back.
    mx = host.create(MS_AUDIO_MIXER_ID)
    assert host.call_int(mx, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(mx, SET_NCHANNELS, 1) == 0
    assert host.call_int(mx, mid(MS_AUDIO_MIXER_ID, 2, 4), 1) == 0       # ENABLE_CONFERENCE_MODE
    sa, sb, ka, kb = host.source(), host.source(), host.sink(), host.sink()
    host.link(sa, 0, mx, 0)
    host.link(sb, 0, mx, 1)
    host.link(mx, 0, ka, 0)
    host.link(mx, 1, kb, 0)
    host.S.ms_ticker_attach(host.ticker, mx)
    n = 160
    a = synth_pcm(21, n * 160, rate=16000)
    b = synth_pcm(22, n * 160, rate=16000, sigma=2000.0)
    # phase 0: only A talks, but pin 1 was first looked at in tick 0 and so "contributes" for the first second: tick 0
    # is a bypass tick (pin 1's clock only starts), ticks 1..100 take the mixed path; either way B hears A one tick later
    for t in range(101):
        host.push(sa, a[t * n:(t + 1) * n])
        host.step(1)
    host.step(1)
    gb = host.drain(kb)
    np.testing.assert_array_equal(gb[:101 * n], a[:101 * n])
    assert not host.drain(ka).any()
    # phase 1: pin 1 timed out -> bypass: B's output gets A's block of the PREVIOUS tick (constant one-tick latency),
    # A's own output nothing
    for t in range(101, 106):
        before = host.S.ms2shim_sink_blocks(kb)
        host.push(sa, a[t * n:(t + 1) * n])
        host.step(1)
        assert host.S.ms2shim_sink_blocks(kb) == before + (1 if t > 101 else 0)
    host.step(1)
    np.testing.assert_array_equal(host.drain(kb)[-5 * n:], a[101 * n:106 * n])
    assert host.S.ms2shim_sink_size(ka) == 0
    # phase 2: both talk -> the batch mixes (one tick later): each hears the other
    for t in range(106, 116):
        host.push(sa, a[t * n:(t + 1) * n])
        host.push(sb, b[t * n:(t + 1) * n])
    host.step(12)
    ga, gb = host.drain(ka), host.drain(kb)
    assert len(ga) == len(gb) and len(ga) >= 10 * n
    np.testing.assert_array_equal(ga[:10 * n], b[106 * n:116 * n])      # sum - own
    np.testing.assert_array_equal(gb[:10 * n], a[106 * n:116 * n])
    assert not ga[10 * n:].any() and not gb[10 * n:].any()              # ALWAYS_STREAMOUT: silence once both ran dry
    # phase 3: nobody talks for more than a second: no contributor, no output at all
    host.step(110)
    host.drain(ka), host.drain(kb)
    before = host.S.ms2shim_sink_blocks(kb)
    host.step(5)
    assert host.S.ms2shim_sink_blocks(kb) == before
    # A alone again -> bypass again, out one tick later
    host.push(sa, a[120 * n:121 * n])
    host.step(1)
    assert host.S.ms2shim_sink_size(kb) == 0
    host.step(1)
    np.testing.assert_array_equal(host.drain(kb), a[120 * n:121 * n])
    host.S.ms_ticker_detach(host.ticker, mx)


def test_mixer_graph_at_44100_hz(host, oracle):
    # 10 ms of 44.1 kHz audio is 441 samples -- not a multiple of 4: the batch takes the any-length kernel, never aborts
    mx = host.create(MS_AUDIO_MIXER_ID)
    assert host.call_int(mx, SET_SAMPLE_RATE, 44100) == 0 and host.call_int(mx, SET_NCHANNELS, 1) == 0
    assert host.call_int(mx, mid(MS_AUDIO_MIXER_ID, 2, 4), 1) == 0       # ENABLE_CONFERENCE_MODE
    srcs = [host.source() for _ in range(3)]
    snks = [host.sink() for _ in range(3)]
    for i in range(3):
        host.link(srcs[i], 0, mx, i)
        host.link(mx, i, snks[i], 0)
    host.S.ms_ticker_attach(host.ticker, mx)
    n, nt = 441, 8
    x = np.stack([synth_pcm(60 + i, n * nt, rate=44100) for i in range(3)])
    for t in range(nt):
        for i in range(3):
            host.push(srcs[i], x[i, t * n:(t + 1) * n])
    host.step(nt + 2)
    for i in range(3):
        got = host.drain(snks[i])
        want = np.concatenate([oracle.mixer_tick(x[:, t * n:(t + 1) * n])[0][i] for t in range(nt)])
        np.testing.assert_array_equal(got[:n * nt], want)
    host.S.ms_ticker_detach(host.ticker, mx)


def test_overlong_blocks_are_split_not_dropped(host, oracle):
    # 60 ms blocks (2880 samples at 48 kHz, e.g. long codec frames) exceed a batch row: the facades cut them into
    # row-sized pieces over as many rounds / ticks as it takes.  Equalizer: the FIR does not care about the blocking, so
    # the samples equal the oracle fed the whole blocks.  Volume (static gain, light path): every sample comes out
    # scaled, none is lost.
    n, nb = 2880, 4
    x = synth_pcm(77, n * nb, rate=48000, sigma=2500.0)
    src, eq, snk = host.source(), host.create(MS_EQUALIZER_ID), host.sink()
    assert host.call_int(eq, SET_SAMPLE_RATE, 48000) == 0
    assert host.call(eq, mid(MS_EQUALIZER_ID, 0, 12), EqGain(2000.0, 1.5, 800.0)) == 0
    host.link(src, 0, eq, 0)
    host.link(eq, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    for b in range(nb):
        host.push(src, x[b * n:(b + 1) * n])
    host.step(nb + 3)
    got = host.drain(snk)
    o = oracle.Equalizer(48000)
    o.set_gain(2000, 1.5, 800)
    want = np.concatenate([o.run(x[b * n:(b + 1) * n]) for b in range(nb)])
    np.testing.assert_array_equal(got, want)
    host.S.ms_ticker_detach(host.ticker, src)

    src, vol, snk = host.source(), host.create(MS_VOLUME_ID), host.sink()
    assert host.call_int(vol, SET_SAMPLE_RATE, 48000) == 0
    g = C.c_float(0.25)
    assert host.call(vol, mid(MS_VOLUME_ID, 2, 4), g) == 0           # MS_VOLUME_SET_GAIN
    host.link(src, 0, vol, 0)
    host.link(vol, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, src)
    for b in range(nb):
        host.push(src, x[b * n:(b + 1) * n])
    host.step(nb + 3)
    got = host.drain(snk)
    assert len(got) == n * nb
    # gain 0.25 from the first chunk on (set_gain also sets the current gain): (s * 1024) / 4096 with C division
    want = (np.abs(x.astype(np.int64)) * 1024 // 4096 * np.sign(x)).astype(np.int16)
    np.testing.assert_array_equal(got, want)
    host.S.ms_ticker_detach(host.ticker, src)


def test_speex_ec_state_string_carries_convergence_over(host):
    """MS_ECHO_CANCELLER_GET_STATE_STRING / SET_STATE_STRING (speexec.c:361-374 with fetch_config / apply_config :119-167):
    the string taken from a converged canceller, given to a NEW filter before it is attached, makes that one start
    converged; a string that does not fit (another tail length) is refused with an error and the filter starts cold."""
    rate, ns, nt = 16000, 160, 150
    GET_STATE, SET_STATE = mid(EC_IFACE, 5, 8), mid(EC_IFACE, 6, 1)
    rng = np.random.default_rng(8)
    far = np.clip(np.round(rng.normal(0, 3000, ns * nt)), -32767, 32767).astype(np.int16)
    ir = rng.normal(0, 1, 48) * np.exp(-np.arange(48) / 10.0)
    mic = np.clip(np.round(0.4 * np.convolve(far.astype(float), ir)[:ns * nt] + rng.normal(0, 30, ns * nt)), -32767, 32767).astype(np.int16)

    def run(state=None, tail=128, ticks=nt):
        ec = host.create(MS_SPEEX_EC_ID)
        assert host.call_int(ec, SET_SAMPLE_RATE, rate) == 0 and host.call_int(ec, mid(EC_IFACE, 2, 4), tail) == 0
        if state is not None:
            assert host.S.ms_filter_call_method(ec, SET_STATE, C.c_char_p(state)) == 0
        s_ref, s_mic, k_ref, k_mic = host.source(), host.source(), host.sink(), host.sink()
        host.link(s_ref, 0, ec, 0)
        host.link(s_mic, 0, ec, 1)
        host.link(ec, 0, k_ref, 0)
        host.link(ec, 1, k_mic, 0)
        host.S.ms_ticker_attach(host.ticker, ec)
        for t in range(ticks):
            host.push(s_ref, far[t * ns:(t + 1) * ns])
            host.push(s_mic, mic[t * ns:(t + 1) * ns])
            host.step()
        host.step(2)
        p = C.c_char_p()
        assert host.S.ms_filter_call_method(ec, GET_STATE, C.byref(p)) == 0
        txt = p.value
        out = host.drain(k_mic)
        host.S.ms_ticker_detach(host.ticker, ec)
        return out, txt

    out_cold, state = run()
    assert state and len(state) > 10000 and set(state) <= set(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/=")
    head = slice(ns * 5, ns * 45)  # the first 400 ms after start-up
    e_mic = float((mic[head].astype(float) ** 2).mean())
    e_cold = float((out_cold[head].astype(float) ** 2).mean())
    out_warm, _ = run(state=state, ticks=50)
    e_warm = float((out_warm[head].astype(float) ** 2).mean())
    assert e_warm < 0.1 * e_cold and e_warm < 0.05 * e_mic   # converged from the first frames (-13 dB or better at once)
    out_bad, _ = run(state=state, tail=64, ticks=50)       # the blob is for 128 ms: refused, cold start
    e_bad = float((out_bad[head].astype(float) ** 2).mean())
    assert e_bad > 5 * e_warm


@pytest.mark.parametrize("kind", ["simple", "double"])
def test_speex_ec_plugin_meets_the_reference_testers_thresholds(host, kind):
    """The reference's echo-canceller tester scenario (tester/mediastreamer2_aec3_tester.c:654-739) run through the
    drop-in plugin: far-end recording on pin 0, near-end + echo on pin 1, 10 ms ticks, the cleaned output graded with
    the tester's own metric and thresholds (energy in the near-end's silences < 1, similarity in speech > 0.99 / 0.83;
    audiodiff.c:442-576 restated in oracle/audiodiff.py).  See tests/test_aec_recordings.py for the DC-notch caveat."""
    from test_aec_recordings import CASES, RATE, TAIL_MS, grade, load, notch
    near, far, mic = load(kind)
    ns = RATE // 100
    nt = len(mic) // ns
    ec = host.create(MS_SPEEX_EC_ID)
    assert host.call_int(ec, SET_SAMPLE_RATE, RATE) == 0
    assert host.call_int(ec, mid(EC_IFACE, 2, 4), TAIL_MS) == 0
    s_ref, s_mic, k_ref, k_mic = host.source(), host.source(), host.sink(), host.sink()
    host.link(s_ref, 0, ec, 0)
    host.link(s_mic, 0, ec, 1)
    host.link(ec, 0, k_ref, 0)
    host.link(ec, 1, k_mic, 0)
    host.S.ms_ticker_attach(host.ticker, ec)
    for t in range(nt):
        host.push(s_ref, far[t * ns:(t + 1) * ns])
        host.push(s_mic, mic[t * ns:(t + 1) * ns])
        host.step()
    host.step(3)
    out = host.drain(k_mic)
    host.S.ms_ticker_detach(host.ticker, ec)
    assert len(out) >= (nt - 3) * ns
    _, _, _, thr_sim, thr_en = CASES[kind]
    n = min(len(out), len(near))
    sim, energy, _ = grade(notch(near)[:n], out[:n], kind)
    _, energy_unprocessed, _ = grade(near[:n], mic[:n], kind)
    assert energy_unprocessed > 50.0
    assert energy < thr_en, energy
    assert thr_sim < sim <= 1.0, sim


def test_bursts_through_the_batched_facades(host, oracle):
    """A source that hands over everything it holds (an RTP receiver after a hiccup): seven 10 ms blocks in one tick --
    more than a pool's four launch rounds -- through MSResample, MSVolume (AGC and light path) and MSEqualizer.  The
    sample stream must be the oracle's, complete and in order."""
    host.S.ms2shim_source_set_burst.argtypes = [C.c_void_p, C.c_int]

    def run(filt, x, n, per_tick):
        src, snk = host.source(), host.sink()
        host.S.ms2shim_source_set_burst(src, 1)
        host.link(src, 0, filt, 0)
        host.link(filt, 0, snk, 0)
        host.S.ms_ticker_attach(host.ticker, src)
        k = 0
        for cnt in per_tick:
            for _ in range(cnt):
                host.push(src, x[k * n:(k + 1) * n])
                k += 1
            host.step()
        host.step(3)
        out = host.drain(snk)
        host.S.ms_ticker_detach(host.ticker, src)
        return out, k

    pattern = [1, 1, 7, 0, 0, 1, 9, 1, 1]
    nblk = sum(pattern)
    # resampler 16k -> 48k
    x = synth_pcm(90, 160 * nblk, rate=16000)
    rs = host.create(MS_RESAMPLE_ID)
    assert host.call_int(rs, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000) == 0
    got, k = run(rs, x, 160, pattern)
    o = oracle.Resampler(16000, 48000)
    ref = np.concatenate([o.process(x[t * 160:(t + 1) * 160]) for t in range(k)])
    assert len(got) == len(ref) and np.abs(got.astype(int) - ref.astype(int)).max() <= 1
    # volume with AGC (re-framed to 10 ms chunks)
    x = synth_pcm(91, 480 * nblk, sigma=5000.0)
    vol = host.create(MS_VOLUME_ID)
    assert host.call_int(vol, SET_SAMPLE_RATE, 48000) == 0 and host.call_int(vol, mid(MS_VOLUME_ID, 8, 4), 1) == 0
    got, k = run(vol, x, 480, pattern)
    o = oracle.Volume(48000)
    o.v.agc_enabled = 1
    np.testing.assert_array_equal(got, np.concatenate([o.chunk(x[t * 480:(t + 1) * 480]) for t in range(k)]))
    # volume, light path (one chunk per block), static gain
    vol = host.create(MS_VOLUME_ID)
    assert host.call_int(vol, SET_SAMPLE_RATE, 48000) == 0
    g = C.c_float(0.5)
    assert host.call(vol, mid(MS_VOLUME_ID, 2, 4), g) == 0
    got, k = run(vol, x, 480, pattern)
    o = oracle.Volume(48000)
    oracle.lib().orc_volume_set_gain(o.v, 0.5)
    np.testing.assert_array_equal(got, np.concatenate([o.chunk(x[t * 480:(t + 1) * 480]) for t in range(k)]))
    # equalizer
    x = synth_pcm(92, 160 * nblk, sigma=2500.0, rate=16000)
    eq = host.create(MS_EQUALIZER_ID)
    assert host.call_int(eq, SET_SAMPLE_RATE, 16000) == 0
    assert host.call(eq, mid(MS_EQUALIZER_ID, 0, 12), EqGain(1000.0, 2.0, 500.0)) == 0
    got, k = run(eq, x, 160, pattern)
    o = oracle.Equalizer(16000)
    o.set_gain(1000, 2.0, 500)
    np.testing.assert_array_equal(got, np.concatenate([o.run(x[t * 160:(t + 1) * 160]) for t in range(k)]))


def test_speex_ec_burst_of_frames_in_one_tick(host, oracle):
    """Five ticks' worth of microphone and far-end audio delivered in ONE tick (more canceller frames than the pool has
    launch rounds): the filter must still produce exactly the frame sequence the reference's framing produces."""
    rate, F, ns = 16000, 128, 160
    host.S.ms2shim_source_set_burst.argtypes = [C.c_void_p, C.c_int]
    ec = host.create(MS_SPEEX_EC_ID)
    assert host.call_int(ec, SET_SAMPLE_RATE, rate) == 0 and host.call_int(ec, mid(EC_IFACE, 2, 4), 128) == 0
    s_ref, s_mic, k_ref, k_mic = host.source(), host.source(), host.sink(), host.sink()
    host.S.ms2shim_source_set_burst(s_ref, 1)
    host.S.ms2shim_source_set_burst(s_mic, 1)
    host.link(s_ref, 0, ec, 0)
    host.link(s_mic, 0, ec, 1)
    host.link(ec, 0, k_ref, 0)
    host.link(ec, 1, k_mic, 0)
    host.S.ms_ticker_attach(host.ticker, ec)
    pattern = [1, 1, 1, 5, 0, 0, 1, 6, 1, 1]
    nblk = sum(pattern)
    rng = np.random.default_rng(6)
    far = np.clip(np.round(rng.normal(0, 3000, ns * nblk)), -32767, 32767).astype(np.int16)
    ir = rng.normal(0, 1, 48) * np.exp(-np.arange(48) / 10.0)
    mic = np.clip(np.round(0.4 * np.convolve(far.astype(float), ir)[:ns * nblk] + rng.normal(0, 50, ns * nblk)),
                  -32767, 32767).astype(np.int16)
    k = 0
    e = oracle.Echo(F, 128 * rate // 1000, rate)
    pp = oracle.Preproc(F, rate, e)
    echo_fifo, dref_fifo = np.zeros(0, np.int16), np.zeros(0, np.int16)
    started, outs = False, []
    for cnt in pattern:
        for _ in range(cnt):
            host.push(s_ref, far[k * ns:(k + 1) * ns])
            host.push(s_mic, mic[k * ns:(k + 1) * ns])
            k += 1
        host.step()
        # the same framing through the oracle: far-end blocks of the tick first (kept only once the microphone started),
        # then the microphone blocks, then every complete frame (speexec.c:241-305)
        lo, hi = (k - cnt) * ns, k * ns
        if started:
            dref_fifo = np.concatenate([dref_fifo, far[lo:hi]])
        echo_fifo = np.concatenate([echo_fifo, mic[lo:hi]])
        while len(echo_fifo) >= F:
            fr, echo_fifo = echo_fifo[:F], echo_fifo[F:]
            started = True
            if len(dref_fifo) < F:
                dref_fifo = np.concatenate([dref_fifo, np.zeros(F, np.int16)])
            r, dref_fifo = dref_fifo[:F], dref_fifo[F:]
            outs.append(pp.run(e.cancel(fr, r)))
    host.step(3)
    got = host.drain(k_mic)
    ref = np.concatenate(outs)
    host.S.ms_ticker_detach(host.ticker, ec)
    assert len(got) == len(ref)
    d = got.astype(np.float64) - ref.astype(np.float64)
    assert np.sqrt(np.mean(d ** 2)) / 32768.0 <= 1e-4


def _runtime_stats(host):
    P = C.CDLL(os.path.join(PKG, "libmsmi355xfilters.so"))
    P.ms_mi355x_late_events.restype = C.c_ulonglong
    h, b, s = C.c_int(), C.c_int(), C.c_int()
    P.ms_mi355x_runtime_stats(C.byref(h), C.byref(b), C.byref(s))
    return h.value, b.value, s.value, P.ms_mi355x_late_events()


def test_mixer_keeps_mixing_after_detach_and_reattach_on_the_same_ticker(host):
    """MSAudioConference detaches and re-attaches its mixer on the same ticker at every member add / remove
    (src/voip/audioconference.c:325-327,:369-374), and the ticker drops a detached filter's postponed tasks
    (src/base/msticker.c:187-190,:314-324): the flush request the mixer had pending dies with the detach.  The plugin must
    post a new one afterwards -- with a sticky 'flush pending' flag the mixer never emitted another block.  And nothing is lost
    at the detach: the tick in flight is delivered by the graph's first postprocess (filters.cpp facade_detached) -- it waits on
    the sink's queue until the sink is walked again -- so the three rounds are ONE gapless stream, as the reference's synchronous
    mixer would have produced it (msticker.c:197-218)."""
    mx = host.create(MS_AUDIO_MIXER_ID)
    assert host.call_int(mx, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(mx, SET_NCHANNELS, 1) == 0
    assert host.call_int(mx, mid(MS_AUDIO_MIXER_ID, 2, 4), 1) == 0       # ENABLE_CONFERENCE_MODE
    sa, sb, ka, kb = host.source(), host.source(), host.sink(), host.sink()
    host.link(sa, 0, mx, 0)
    host.link(sb, 0, mx, 1)
    host.link(mx, 0, ka, 0)
    host.link(mx, 1, kb, 0)
    n = 160
    a = synth_pcm(31, n * 60, rate=16000)
    b = synth_pcm(32, n * 60, rate=16000, sigma=2000.0)
    pos = 0
    all_a, all_b = [], []
    for round_ in range(3):
        host.S.ms_ticker_attach(host.ticker, mx)
        for t in range(10):
            host.push(sa, a[(pos + t) * n:(pos + t + 1) * n])
            host.push(sb, b[(pos + t) * n:(pos + t + 1) * n])
            host.step(1)
        # detach right after a tick that staged work: the mixer's flush request is pending and is dropped by the ticker
        host.S.ms_ticker_detach(host.ticker, mx)
        ga, gb = host.drain(ka), host.drain(kb)
        assert len(ga) >= 9 * n, f"round {round_}: the mixer delivered {len(ga)} samples"
        all_a.append(ga)
        all_b.append(gb)
        pos += 10
    ga, gb = np.concatenate(all_a), np.concatenate(all_b)
    assert len(ga) == len(gb) == 29 * n   # (the 30th block lies on the sinks' queues: delivered at the last detach, never walked again)
    np.testing.assert_array_equal(ga, b[:29 * n])
    np.testing.assert_array_equal(gb, a[:29 * n])
    for f in (mx, sa, sb, ka, kb):
        host.S.ms_filter_destroy(f)


def test_banks_grow_and_are_freed_with_their_last_slot(host, oracle):
    """40 MSVolume filters on one ticker: banks of 16 and 64 slots (no 'pool exhausted'), every filter served; once the
    filters are destroyed the banks (device objects, pinned buffers) and the ticker's hub are gone -- a ticker per call
    must not leak one pool set per call."""
    h0, b0, s0, late0 = _runtime_stats(host)
    S = host.S
    tk = S.ms_ticker_new()
    chains = []
    for i in range(40):
        v = host.create(MS_VOLUME_ID)
        assert host.call_int(v, SET_SAMPLE_RATE, 16000) == 0
        g = C.c_float(0.5)
        assert host.call(v, mid(MS_VOLUME_ID, 2, 4), g) == 0  # MS_VOLUME_SET_GAIN
        src, snk = host.source(), host.sink()
        host.link(src, 0, v, 0)
        host.link(v, 0, snk, 0)
        S.ms_ticker_attach(tk, v)
        chains.append((src, v, snk))
    h1, b1, s1, _ = _runtime_stats(host)
    assert s1 - s0 == 40 and b1 - b0 == 2 and h1 - h0 == 1
    x = [synth_pcm(100 + i, 160 * 4, rate=16000) for i in range(40)]
    for t in range(4):
        for i, (src, _, _) in enumerate(chains):
            host.push(src, x[i][t * 160:(t + 1) * 160])
        S.ms_ticker_step(tk)
    S.ms_ticker_step(tk)
    for i, (_, _, snk) in enumerate(chains):
        got = host.drain(snk)
        vo = oracle.Volume(16000)
        oracle.lib().orc_volume_set_gain(vo.v, 0.5)
        want = np.concatenate([vo.chunk(x[i][t * 160:(t + 1) * 160].copy()) for t in range(4)])
        np.testing.assert_array_equal(got, want)
    for src, v, snk in chains:
        S.ms_ticker_detach(tk, v)
    for src, v, snk in chains:
        for f in (src, v, snk):
            S.ms_filter_destroy(f)
    S.ms_ticker_destroy(tk)
    h2, b2, s2, late2 = _runtime_stats(host)
    assert (h2, b2, s2) == (h0, b0, s0), "banks / hub of the destroyed ticker are still alive"
    assert late2 == late0


def test_hubs_come_and_go_under_a_thread_that_walks_them(host):
    """Three threads each run 'calls': a ticker of their own, a few MSVolume graphs, some ticks, everything destroyed --
    the ticker's hub (lock, HIP stream, banks) dies with its last slot -- while the main thread keeps walking every hub
    (ms_mi355x_runtime_stats, ms_mi355x_flush).  A hub found in the registry must stay valid until the walker has locked
    and released it (reference taken under the registry lock), and a walker that gets a hub whose last bank has just gone
    must leave it alone."""
    import threading
    S = host.S
    P = C.CDLL(os.path.join(PKG, "libmsmi355xfilters.so"))
    h0, b0, s0, late0 = _runtime_stats(host)
    stop = threading.Event()
    errors = []
    x = synth_pcm(77, 160, rate=16000)

    def calls(seed):
        try:
            for rep in range(12):
                tk = S.ms_ticker_new()
                chains = []
                for i in range(3 + (seed + rep) % 3):
                    v = host.create(MS_VOLUME_ID)
                    host.call_int(v, SET_SAMPLE_RATE, 16000)
                    src, snk = host.source(), host.sink()
                    host.link(src, 0, v, 0)
                    host.link(v, 0, snk, 0)
                    S.ms_ticker_attach(tk, v)
                    chains.append((src, v, snk))
                for t in range(3):
                    for src, _, _ in chains:
                        S.ms2shim_source_push(src, x.ctypes.data, x.nbytes)
                    S.ms_ticker_step(tk)
                for src, v, snk in chains:
                    S.ms_ticker_detach(tk, v)
                for src, v, snk in chains:
                    for f in (src, v, snk):
                        S.ms_filter_destroy(f)
                S.ms_ticker_destroy(tk)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=calls, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    walks = 0
    while any(t.is_alive() for t in th):
        hh, bb, ss = C.c_int(), C.c_int(), C.c_int()
        P.ms_mi355x_runtime_stats(C.byref(hh), C.byref(bb), C.byref(ss))
        assert hh.value >= 0 and bb.value >= 0 and ss.value >= 0
        walks += 1
    for t in th:
        t.join()
    stop.set()
    assert not errors, errors
    assert walks > 10
    h2, b2, s2, late2 = _runtime_stats(host)
    assert (h2, b2, s2) == (h0, b0, s0) and late2 == late0


def test_a_chain_of_gpu_facades_costs_one_tick_in_total(host, oracle):
    """source -> MSResample 8k->16k -> MSVolume (gain 0.5) -> MSEqualizer (flat) -> sink: the flush task at the start of
    the next tick runs the resampler's bank, hands its output to the volume facade, runs that bank, and so on -- the
    block pushed in tick t is at the sink after tick t+1, not t+3 (src/base/msticker.c:301-312 runs tasks before graphs).
    Content: the chain of oracles."""
    rs, vol, eq = host.create(MS_RESAMPLE_ID), host.create(MS_VOLUME_ID), host.create(MS_EQUALIZER_ID)
    assert host.call_int(rs, SET_SAMPLE_RATE, 8000) == 0 and host.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 16000) == 0
    assert host.call_int(vol, SET_SAMPLE_RATE, 16000) == 0 and host.call_int(eq, SET_SAMPLE_RATE, 16000) == 0
    g = C.c_float(0.5)
    assert host.call(vol, mid(MS_VOLUME_ID, 2, 4), g) == 0  # MS_VOLUME_SET_GAIN
    src, snk = host.source(), host.sink()
    host.link(src, 0, rs, 0)
    host.link(rs, 0, vol, 0)
    host.link(vol, 0, eq, 0)
    host.link(eq, 0, snk, 0)
    host.S.ms_ticker_attach(host.ticker, rs)
    nt = 12
    x = synth_pcm(77, 80 * nt, rate=8000)
    arrived = []
    for t in range(nt):
        host.push(src, x[t * 80:(t + 1) * 80])
        host.step(1)
        arrived.append(host.S.ms2shim_sink_blocks(snk))
    host.step(1)
    arrived.append(host.S.ms2shim_sink_blocks(snk))
    assert arrived[0] == 0 and arrived[1] == 1, f"blocks at the sink after each tick: {arrived}"   # one tick for three facades
    assert arrived[-1] == nt
    got = host.drain(snk)
    r = oracle.Resampler(8000, 16000)
    vo = oracle.Volume(16000)
    oracle.lib().orc_volume_set_gain(vo.v, 0.5)
    eqo = oracle.Equalizer(16000)
    want = []
    for t in range(nt):
        up = r.process(x[t * 80:(t + 1) * 80])
        want.append(eqo.run(vo.chunk(up.copy())))
    want = np.concatenate(want)
    assert len(got) == len(want)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1   # the resampler's 1 LSB (FMA order)
    host.S.ms_ticker_detach(host.ticker, rs)
    for f in (rs, vol, eq, src, snk):
        host.S.ms_filter_destroy(f)
