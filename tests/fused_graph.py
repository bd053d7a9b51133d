"""Conference graphs of the plugin's facades in the test runtime (tests/host/ms2shim.c), as the reference builds a call
leg's sending side (src/voip/audiostream.c:1798-1810) in front of a conference mixer (src/voip/audioconference.c:209-257):

    mic source -> MSResample (16k -> 48k) -> MSSpeexEC pin 1 -> MSVolume (AGC) -> MSAudioMixer pin k -> sink k
    far-end source -------------------------> MSSpeexEC pin 0 -> speaker sink

run twice on the same inputs -- with the fused call-leg batch (mediastreamer2_amd/host/filters/leg_chain.inl) and with
MSMI355X_NO_FUSE=1 (every facade on its own bank: the path the oracle tests pin) -- for tests/test_gpu_plugin_fused.py (the
product plugin on the GPU) and tests/test_plugin_fused_cpu.py (the same host code against the host-memory double of the
kernel library: `python tests/fused_graph.py --double` in a process of its own)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")

MS_SPEEX_EC_ID, MS_RESAMPLE_ID, MS_VOLUME_ID, MS_AUDIO_MIXER_ID = 28, 41, 43, 68
MS_EQUALIZER_ID = 61
MS_ULAW_ENC_ID, MS_ULAW_DEC_ID, MS_GENERIC_PLC_ID = 7, 8, 111
MS_ALAW_ENC_ID, MS_ALAW_DEC_ID, MS_AUDIO_FLOW_CONTROL_ID = 9, 10, 141
MS_FILTER_BASE_ID = 2
EC_IFACE = 16384 + 4


def mid(fid, idx, argsize):
    return ((fid & 0xFFFF) << 16) | (idx << 8) | (argsize & 0xFF)


def _method_ids():
    """the method ids as include/ms2_plugin_abi.h spells them (the header is the contract; indices are not guessed)"""
    import re
    txt = open(os.path.join(ROOT, "include", "ms2_plugin_abi.h")).read()
    owner = {"MS_VOLUME_ID": MS_VOLUME_ID, "MS_AUDIO_MIXER_ID": MS_AUDIO_MIXER_ID, "MSFilterEchoCancellerInterface": EC_IFACE}
    size = {"int": 4, "float": 4, "bool_t": 1, "char *": 8, "const char": 1}
    out = {}
    for name, idx, typ in re.findall(r"#define\s+(MS_\w+)\s+MS_FILTER_BASE_METHOD\(\s*(\d+)\s*,\s*([\w \*]+?)\s*\)", txt):
        if typ in size:
            out[name] = mid(MS_FILTER_BASE_ID, int(idx), size[typ])
    for name, own, idx, typ in re.findall(r"#define\s+(MS_\w+)\s+MS_FILTER_METHOD\(\s*(\w+)\s*,\s*(\d+)\s*,\s*([\w \*]+?)\s*\)", txt):
        if own in owner and typ in size:
            out[name] = mid(owner[own], int(idx), size[typ])
    return out


IDS = _method_ids()
EC_SET_TAIL = IDS["MS_ECHO_CANCELLER_SET_TAIL_LENGTH"]
EC_SET_DELAY = IDS["MS_ECHO_CANCELLER_SET_DELAY"]
EC_SET_BYPASS = IDS["MS_ECHO_CANCELLER_SET_BYPASS_MODE"]
VOL_SET_GAIN = IDS["MS_VOLUME_SET_GAIN"]
VOL_GET_LINEAR = IDS["MS_VOLUME_GET_LINEAR"]
VOL_ENABLE_AGC = IDS["MS_VOLUME_ENABLE_AGC"]
MIX_CONF_MODE = IDS["MS_AUDIO_MIXER_ENABLE_CONFERENCE_MODE"]


class Host:
    """ctypes view of the shim runtime with the plugin loaded (product or double build)"""

    def __init__(self, plugin_dir):
        self.S = S = C.CDLL(os.path.join(HOST, "libms2shim.so"), mode=C.RTLD_GLOBAL)
        vp = C.c_void_p
        for fn in ("ms_factory_new", "ms_factory_create_filter", "ms_ticker_new", "ms2shim_new_source", "ms2shim_new_sink"):
            getattr(S, fn).restype = vp
        S.ms_factory_create_filter.argtypes = [vp, C.c_int]
        S.ms_factory_load_plugin.argtypes = [vp, C.c_char_p]
        S.ms2shim_register_test_filters.argtypes = [vp]
        S.ms2shim_new_source.argtypes = [vp]
        S.ms2shim_new_sink.argtypes = [vp]
        S.ms2shim_source_push.argtypes = [vp, vp, C.c_size_t]
        S.ms2shim_source_set_burst.argtypes = [vp, C.c_int]
        S.ms2shim_volume_set_peer.argtypes = [vp, vp]
        S.ms2shim_new_pass.restype = vp
        S.ms2shim_new_pass.argtypes = [vp]
        S.ms2shim_equalizer_set_gain.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        S.ms2shim_equalizer_set_active.argtypes = [vp, C.c_int]
        S.ms2shim_flow_control_drop.argtypes = [vp, C.c_uint, C.c_uint]
        S.ms2shim_sink_read.restype = C.c_size_t
        S.ms2shim_sink_read.argtypes = [vp, vp, C.c_size_t]
        S.ms2shim_sink_size.restype = C.c_size_t
        S.ms2shim_sink_size.argtypes = [vp]
        S.ms2shim_sink_blocks.argtypes = [vp]
        S.ms_filter_link.argtypes = [vp, C.c_int, vp, C.c_int]
        S.ms_filter_call_method.argtypes = [vp, C.c_uint, vp]
        S.ms_filter_destroy.argtypes = [vp]
        S.ms_ticker_attach.argtypes = [vp, vp]
        S.ms_ticker_detach.argtypes = [vp, vp]
        S.ms_ticker_step.argtypes = [vp]
        S.ms_ticker_destroy.argtypes = [vp]
        self.fac = S.ms_factory_new()
        S.ms2shim_register_test_filters(self.fac)
        plugin = os.path.join(plugin_dir, "libmsmi355xfilters.so")
        assert S.ms_factory_load_plugin(self.fac, plugin.encode()) == 0
        self.P = C.CDLL(plugin)
        self.P.ms_mi355x_late_events.restype = C.c_ulonglong

    def recv_streams(self):
        return int(self.P.ms_mi355x_recv_stats())

    def fused_stats(self):
        c, l = C.c_int(), C.c_int()
        la, fr = C.c_ulonglong(), C.c_ulonglong()
        self.P.ms_mi355x_fused_stats(C.byref(c), C.byref(l), C.byref(la), C.byref(fr))
        return {"conferences": c.value, "legs": l.value, "launches": la.value, "flush_rounds": fr.value}

    def runtime_stats(self):
        h, b, s = C.c_int(), C.c_int(), C.c_int()
        self.P.ms_mi355x_runtime_stats(C.byref(h), C.byref(b), C.byref(s))
        return h.value, b.value, s.value

    def call_int(self, f, method, val):
        v = C.c_int(val)
        return self.S.ms_filter_call_method(f, method, C.byref(v))

    def call_float(self, f, method, val):
        v = C.c_float(val)
        return self.S.ms_filter_call_method(f, method, C.byref(v))

    def call_bool(self, f, method, val):
        v = C.c_ubyte(val)
        return self.S.ms_filter_call_method(f, method, C.byref(v))

    def get_float(self, f, method):
        v = C.c_float()
        assert self.S.ms_filter_call_method(f, method, C.byref(v)) == 0
        return v.value

    def push(self, src, samples):
        a = np.ascontiguousarray(samples, np.int16)
        self.S.ms2shim_source_push(src, a.ctypes.data, a.nbytes)

    def drain(self, sink):
        n = self.S.ms2shim_sink_size(sink)
        buf = np.zeros(n // 2, np.int16)
        if n:
            self.S.ms2shim_sink_read(sink, buf.ctypes.data, n)
        return buf


class Conferences:
    """nconf conferences of `members` legs each on one ticker"""

    def __init__(self, h, nconf, members, in_rate=16000, rate=48000, tail_ms=128, delay_ms=0, agc=True, pins=None, gain=None, mixer=True, resampler=True,
                 endpoint_resamplers=False, echo_limiter=False, mic_equalizer=False, volrecv=False, cpu_filters=False, g711=False, spk_equalizer=False,
                 flowcontrol=False, dtmfgen_rtp=True, encoder=True, local_mixer=0, outbound_mixer=False, alaw=False, idle_equalizers=False, recv_tee=True):
        self.h, self.S = h, h.S
        S = h.S
        self.ticker = S.ms_ticker_new()
        self.nconf, self.members, self.in_rate, self.rate = nconf, members, in_rate, rate
        self.pins = list(range(members)) if pins is None else list(pins)
        base = lambda name: IDS[name]
        self.legs, self.mixers = [], []
        self.with_mixer = mixer
        for c in range(nconf):
            mx = None
            if mixer:  # else: an AudioStream's sending side on a shared ticker -- MSVolume's output goes straight to the sink
                mx = S.ms_factory_create_filter(h.fac, MS_AUDIO_MIXER_ID)
                h.call_int(mx, base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                h.call_int(mx, MIX_CONF_MODE, 1)
                self.mixers.append(mx)
            for k in range(members):
                leg = {"mic": S.ms2shim_new_source(h.fac), "far": S.ms2shim_new_source(h.fac), "spk": S.ms2shim_new_sink(h.fac),
                       "out": S.ms2shim_new_sink(h.fac), "rs": S.ms_factory_create_filter(h.fac, MS_RESAMPLE_ID),
                       "ec": S.ms_factory_create_filter(h.fac, MS_SPEEX_EC_ID), "vol": S.ms_factory_create_filter(h.fac, MS_VOLUME_ID),
                       "mixer": mx, "pin": self.pins[k]}
                h.call_int(leg["rs"], base("MS_FILTER_SET_SAMPLE_RATE"), in_rate)
                h.call_int(leg["rs"], base("MS_FILTER_SET_OUTPUT_SAMPLE_RATE"), rate)
                h.call_int(leg["ec"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                h.call_int(leg["ec"], EC_SET_TAIL, tail_ms)
                h.call_int(leg["ec"], EC_SET_DELAY, delay_ms)
                h.call_int(leg["vol"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                if agc:
                    h.call_int(leg["vol"], VOL_ENABLE_AGC, 1)
                if gain is not None:
                    h.call_float(leg["vol"], VOL_SET_GAIN, gain)
                # (resampler=False: the sound card / decoder already runs at the canceller's rate -- MSSpeexEC is the head of the leg; the
                # MSResample is created all the same and stays unlinked)
                links = [(leg["mic"], 0, leg["rs"], 0), (leg["rs"], 0, leg["ec"], 1)] if resampler else [(leg["mic"], 0, leg["ec"], 1)]
                if idle_equalizers:   # AUDIO_STREAM_FEATURE_EQUALIZER (part of AUDIO_STREAM_FEATURE_ALL, audiostream.c:1623-1640): BOTH equalizers exist, neither is active
                    leg["eq"], leg["spk_eq"] = (S.ms_factory_create_filter(h.fac, MS_EQUALIZER_ID) for _ in range(2))
                    for e in (leg["eq"], leg["spk_eq"]):
                        h.call_int(e, base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                        assert S.ms2shim_equalizer_set_gain(e, 1200.0, 2.0, 500.0) == 0   # (a response that WOULD be heard)
                        assert S.ms2shim_equalizer_set_active(e, 0) == 0
                    links = ([(leg["mic"], 0, leg["rs"], 0), (leg["rs"], 0, leg["eq"], 0)] if resampler else [(leg["mic"], 0, leg["eq"], 0)]) + [(leg["eq"], 0, leg["ec"], 1)]
                if mic_equalizer:   # audiostream.c:1801: between read_resampler and ec, a response of its own per leg
                    leg["eq"] = S.ms_factory_create_filter(h.fac, MS_EQUALIZER_ID)
                    h.call_int(leg["eq"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                    assert S.ms2shim_equalizer_set_gain(leg["eq"], 1000.0 + 300.0 * k, 2.5, 600.0) == 0
                    assert S.ms2shim_equalizer_set_gain(leg["eq"], 4000.0, 0.4, 1500.0) == 0
                    links = [(leg["mic"], 0, leg["rs"], 0), (leg["rs"], 0, leg["eq"], 0), (leg["eq"], 0, leg["ec"], 1)]
                links += [(leg["ec"], 1, leg["vol"], 0), (leg["ec"], 0, leg["spk"], 0)]
                if echo_limiter or volrecv:   # audio_stream_enable_echo_limiter (audiostream.c:2236-2240): volrecv upstream of the canceller's far end, volsend's peer
                    leg["volrecv"] = S.ms_factory_create_filter(h.fac, MS_VOLUME_ID)
                    h.call_int(leg["volrecv"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                    if echo_limiter:
                        assert S.ms2shim_volume_set_peer(leg["vol"], leg["volrecv"]) == 0
                        h.call_float(leg["vol"], IDS["MS_VOLUME_SET_EA_THRESHOLD"], 0.002)   # (the scene's far end meters ~0.01: the limiter works)
                        h.call_float(leg["vol"], IDS["MS_VOLUME_SET_EA_FORCE"], 20.0)
                    if cpu_filters:   # dtmfgen in front of volrecv, recv_tee behind it (audiostream.c:1826-1827): the application's own filters
                        leg["dtmfgen"], leg["recv_tee"] = S.ms2shim_new_pass(h.fac), S.ms2shim_new_pass(h.fac)
                        head = leg["far"]
                        if g711:   # rtprecv's packets -> MSUlawDec -> [local_mixer] -> MSGenericPLC -> [MSAudioFlowControl] (audiostream.c:1813-1824), facades of the plugin
                            leg["dec"], leg["plc"] = S.ms_factory_create_filter(h.fac, MS_ALAW_DEC_ID if alaw else MS_ULAW_DEC_ID), S.ms_factory_create_filter(h.fac, MS_GENERIC_PLC_ID)
                            h.call_int(leg["plc"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                            up = leg["dec"]
                            links += [(leg["far"], 0, leg["dec"], 0)]
                            if local_mixer:   # AUDIO_STREAM_FEATURE_LOCAL_PLAYING (audiostream.c:1770-1772,1815): not a conference; 2 = its local player is linked too (and idle)
                                leg["local_mixer"] = S.ms_factory_create_filter(h.fac, MS_AUDIO_MIXER_ID)
                                h.call_int(leg["local_mixer"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                                links += [(up, 0, leg["local_mixer"], 0)]
                                if local_mixer == 2:
                                    leg["local_player"] = S.ms2shim_new_source(h.fac)
                                    links += [(leg["local_player"], 0, leg["local_mixer"], 1)]
                                up = leg["local_mixer"]
                            links += [(up, 0, leg["plc"], 0)]
                            head = leg["plc"]
                            if flowcontrol:   # AUDIO_STREAM_FEATURE_FLOW_CONTROL (audiostream.c:1754-1766,1824)
                                leg["fc"] = S.ms_factory_create_filter(h.fac, MS_AUDIO_FLOW_CONTROL_ID)
                                h.call_int(leg["fc"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                                h.call_int(leg["fc"], base("MS_FILTER_SET_NCHANNELS"), 1)
                                links += [(leg["plc"], 0, leg["fc"], 0)]
                                head = leg["fc"]
                        links += [(head, 0, leg["dtmfgen"], 0), (leg["dtmfgen"], 0, leg["volrecv"], 0)]
                        behind = leg["volrecv"]
                        if recv_tee:   # (only with a recording feature, audiostream.c:1776-1786,1827: without one volrecv stands right in front of spk_equalizer)
                            links += [(leg["volrecv"], 0, leg["recv_tee"], 0)]
                            behind = leg["recv_tee"]
                        else:
                            S.ms_filter_destroy(leg.pop("recv_tee"))
                        links += [(behind, 0, leg["spk_eq"], 0), (leg["spk_eq"], 0, leg["ec"], 0)] if idle_equalizers else [(behind, 0, leg["ec"], 0)]
                    elif spk_equalizer:   # audiostream.c:1828: an MSEqualizer of ours right in front of the canceller's far end -- it delivers with the flush
                        leg["spk_eq"] = S.ms_factory_create_filter(h.fac, MS_EQUALIZER_ID)
                        h.call_int(leg["spk_eq"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                        assert S.ms2shim_equalizer_set_gain(leg["spk_eq"], 1500.0, 2.0, 700.0) == 0
                        links += [(leg["far"], 0, leg["volrecv"], 0), (leg["volrecv"], 0, leg["spk_eq"], 0), (leg["spk_eq"], 0, leg["ec"], 0)]
                    else:
                        links += [(leg["far"], 0, leg["volrecv"], 0), (leg["volrecv"], 0, leg["ec"], 0)]
                else:
                    links += [(leg["far"], 0, leg["ec"], 0)]
                if mixer and endpoint_resamplers:
                    # MSAudioConference's plumbing (audioconference.c:209-257): in_resampler in front of the pin, out_resampler behind it --
                    # both at the conference's rate here: they forward (msresample.c:126-135)
                    leg["in_rs"], leg["out_rs"] = (S.ms_factory_create_filter(h.fac, MS_RESAMPLE_ID) for _ in range(2))
                    for f in (leg["in_rs"], leg["out_rs"]):
                        h.call_int(f, base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                        h.call_int(f, base("MS_FILTER_SET_OUTPUT_SAMPLE_RATE"), rate)
                    links += [(leg["vol"], 0, leg["in_rs"], 0), (leg["in_rs"], 0, mx, leg["pin"]), (mx, leg["pin"], leg["out_rs"], 0), (leg["out_rs"], 0, leg["out"], 0)]
                else:
                    if not mixer and cpu_filters:   # volsend -> [dtmfgen_rtp] -> [outbound_mixer] -> (encoder, rtpsend: the sink)
                        up = leg["vol"]
                        if dtmfgen_rtp:   # (only where no telephone-event payload is negotiated, audiostream.c:1396-1404)
                            leg["dtmfgen_rtp"] = S.ms2shim_new_pass(h.fac)
                            links += [(up, 0, leg["dtmfgen_rtp"], 0)]
                            up = leg["dtmfgen_rtp"]
                        if outbound_mixer:   # AUDIO_STREAM_FEATURE_REMOTE_PLAYING (audiostream.c:1585-1588,1807): one linked input unless a remote player is open
                            leg["outbound_mixer"] = S.ms_factory_create_filter(h.fac, MS_AUDIO_MIXER_ID)
                            h.call_int(leg["outbound_mixer"], base("MS_FILTER_SET_SAMPLE_RATE"), rate)
                            links += [(up, 0, leg["outbound_mixer"], 0)]
                            up = leg["outbound_mixer"]
                        if g711 and encoder:   # .. -> MSUlawEnc (its default 20 ms packets) -> rtpsend
                            leg["enc"] = S.ms_factory_create_filter(h.fac, MS_ALAW_ENC_ID if alaw else MS_ULAW_ENC_ID)
                            links += [(up, 0, leg["enc"], 0), (leg["enc"], 0, leg["out"], 0)]
                        else:
                            links += [(up, 0, leg["out"], 0)]
                    else:
                        links += [(leg["vol"], 0, mx, leg["pin"]), (mx, leg["pin"], leg["out"], 0)] if mixer else [(leg["vol"], 0, leg["out"], 0)]
                for a, pa, b, pb in links:
                    assert S.ms_filter_link(a, pa, b, pb) == 0
                self.legs.append(leg)
        self.attached = False

    def roots(self):
        return self.mixers if self.with_mixer else [leg["mic"] for leg in self.legs]

    def attach(self):
        for f in self.roots():
            self.S.ms_ticker_attach(self.ticker, f)
        self.attached = True

    def detach(self):
        for f in self.roots():
            self.S.ms_ticker_detach(self.ticker, f)
        self.attached = False

    def step(self, n=1):
        for _ in range(n):
            self.S.ms_ticker_step(self.ticker)

    def close(self):
        if self.attached:
            self.detach()
        for leg in self.legs:
            for k in ("mic", "far", "spk", "out", "rs", "ec", "vol", "in_rs", "out_rs", "volrecv", "eq", "dtmfgen", "recv_tee", "dtmfgen_rtp", "dec", "plc", "enc", "spk_eq",
                      "fc", "local_mixer", "local_player", "outbound_mixer"):
                if k in leg:
                    self.S.ms_filter_destroy(leg[k])
        for mx in self.mixers:
            self.S.ms_filter_destroy(mx)
        self.S.ms_ticker_destroy(self.ticker)


def scene(nlegs, nticks, in_rate, rate, seed=7):
    """per leg: far end (noise + tone) at `rate`, microphone = 0.5 x the far end 20 ms late + near-end noise, at in_rate"""
    rng = np.random.default_rng(seed)
    ns, ni = rate // 100, in_rate // 100
    t = np.arange(nticks * ns)
    far = (rng.normal(0, 3000, (nlegs, nticks * ns)) + 3276 * np.sin(2 * np.pi * 1000 * t / rate)).round().clip(-32767, 32767).astype(np.int16)
    late = np.concatenate([np.zeros((nlegs, ns * 2), np.int16), far[:, :-ns * 2]], axis=1).astype(np.float64)
    q = rate // in_rate
    mic = (0.5 * late.reshape(nlegs, -1, q).mean(axis=2) + rng.normal(0, 300, (nlegs, nticks * ni))).round().clip(-32767, 32767).astype(np.int16)
    return mic, far


def run(plugin_dir, fuse, scenario, h=None):
    """one scenario, returns {"out": [per leg int16], "spk": [...], "stats": ..}"""
    if fuse:
        os.environ.pop("MSMI355X_NO_FUSE", None)
    else:
        os.environ["MSMI355X_NO_FUSE"] = "1"
    # a method call lands between two ticks; with the bank's work leaving at the END of a graph walk it takes effect one tick
    # later than with the facades one by one (still within a tick of the call): compared sample for sample without that
    if scenario.get("no_early_launch"):
        os.environ["MSMI355X_NO_EARLY_LAUNCH"] = "1"
    else:
        os.environ.pop("MSMI355X_NO_EARLY_LAUNCH", None)
    os.environ["MSMI355X_CHECK_LEVELS"] = "1"
    h = h or Host(plugin_dir)
    sc = dict(nconf=2, members=4, nticks=120, in_rate=16000, rate=48000, tail_ms=128, delay_ms=0, pins=None)
    sc.update(scenario)
    conf = Conferences(h, sc["nconf"], sc["members"], sc["in_rate"], sc["rate"], sc["tail_ms"], sc["delay_ms"], pins=sc["pins"],
                       gain=sc.get("gain"), mixer=not sc.get("no_mixer"), resampler=not sc.get("no_resampler"), agc=not sc.get("no_agc"),
                       endpoint_resamplers=bool(sc.get("endpoint_resamplers")), echo_limiter=bool(sc.get("echo_limiter")), mic_equalizer=bool(sc.get("mic_equalizer")), volrecv=bool(sc.get("volrecv")), cpu_filters=bool(sc.get("cpu_filters")), g711=bool(sc.get("g711")), spk_equalizer=bool(sc.get("spk_equalizer")),
                       flowcontrol=bool(sc.get("flowcontrol")), dtmfgen_rtp=sc.get("dtmfgen_rtp", True), encoder=sc.get("encoder", True), local_mixer=int(sc.get("local_mixer", 0)),
                       outbound_mixer=bool(sc.get("outbound_mixer")), alaw=bool(sc.get("alaw")), idle_equalizers=bool(sc.get("idle_equalizers")), recv_tee=not sc.get("no_recv_tee"))
    n = sc["nconf"] * sc["members"]
    nt, ni, ns = sc["nticks"], sc["in_rate"] // 100, sc["rate"] // 100
    mic, far = scene(n, nt, sc["in_rate"], sc["rate"], seed=sc.get("seed", 7))
    far_codes = None
    if sc.get("g711"):   # what the far endpoints send: their audio as PCMU (the oracle's encoder, pinned against the reference's g711.c)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import oracle
        oracle.build()
        far_codes = [np.ascontiguousarray(oracle.g711_encode(0 if sc.get("alaw") else 1, far[s])) for s in range(n)]
    late0 = h.P.ms_mi355x_late_events()
    before = h.runtime_stats()   # (other graphs of the same process may be alive: what this run leaves behind is the difference)
    conf.attach()
    mid_stats = None
    for t in range(nt):
        for s, leg in enumerate(conf.legs):
            # microphone: 10 ms blocks, or 20 ms packets every other tick (ptime 20)
            if sc.get("ptime20"):
                if t % 2 == 0:
                    h.push(leg["mic"], mic[s, t * ni:(t + 2) * ni])
                else:
                    h.push(leg["mic"], np.zeros(0, np.int16))
            else:
                h.push(leg["mic"], mic[s, t * ni:(t + 1) * ni])
            # far end: regular, or with a late packet every 17th tick per leg (nothing, then two blocks at once)
            if sc.get("g711"):   # G.711 packets of 10 ms, one in 19 lost (the PLC conceals it) unless the scenario says "lossless"
                if sc.get("lossless") or (t + 2 * s) % 19 != 7:
                    pk = far_codes[s][t * ns:(t + 1) * ns]
                    h.S.ms2shim_source_push(leg["far"], pk.ctypes.data, pk.nbytes)
                if "local_player" in leg:
                    h.push(leg["local_player"], np.zeros(0, np.int16))   # (an idle player: nothing)
            elif sc.get("far_gaps") and (t + 3 * s) % 17 == 5:
                h.push(leg["far"], np.zeros(0, np.int16))
            elif sc.get("far_gaps") and (t + 3 * s) % 17 == 6:
                h.push(leg["far"], far[s, (t - 1) * ns:(t + 1) * ns])
            else:
                h.push(leg["far"], far[s, t * ns:(t + 1) * ns])
        for ev in sc.get("events", []):
            if ev[0] == t:
                kind, s, val = ev[1], ev[2], ev[3]
                leg = conf.legs[s]
                if kind == "gain":
                    h.call_float(leg["vol"], VOL_SET_GAIN, val)
                elif kind == "bypass":
                    h.call_bool(leg["ec"], EC_SET_BYPASS, val)
                elif kind == "eq_gain":     # MS_EQUALIZER_SET_GAIN in mid-call
                    assert h.S.ms2shim_equalizer_set_gain(leg["eq"], 2000.0, val, 800.0) == 0
                elif kind == "eq_active":
                    assert h.S.ms2shim_equalizer_set_active(leg["eq"], int(val)) == 0
                elif kind == "eq_rate":     # MS_FILTER_SET_SAMPLE_RATE at the rate it has: the response is flat again (equalizer.c:305-309), the leg stays fused
                    h.call_int(leg["eq"], IDS["MS_FILTER_SET_SAMPLE_RATE"], int(val))
                elif kind == "spk_eq_active":   # the speaker's equalizer switched on in mid-call: no longer transparent
                    assert h.S.ms2shim_equalizer_set_active(leg["spk_eq"], int(val)) == 0
                elif kind == "recv_gain":   # volrecv stops being a meter only: the leg goes back to its facades
                    h.call_float(leg["volrecv"], VOL_SET_GAIN, val)
                elif kind == "agc":
                    h.call_int(leg["vol"], VOL_ENABLE_AGC, val)
                elif kind == "in_rs_rate":   # the endpoint's in_resampler is told to resample after all
                    h.call_int(leg["in_rs"], IDS["MS_FILTER_SET_SAMPLE_RATE"], val)
                elif kind == "flow_drop":   # MS_AUDIO_FLOW_CONTROL_DROP: val ms out of the next second (what the canceller's / the card's drop event asks for)
                    assert h.S.ms2shim_flow_control_drop(leg["fc"], 1000, int(val)) == 0
                elif kind == "reattach":
                    conf.detach()
                    conf.attach()
        conf.step()
        if t == nt // 2:
            mid_stats = h.fused_stats()
            mid_stats["recv_streams"] = h.recv_streams()
    levels = [h.get_float(leg["vol"], VOL_GET_LINEAR) for leg in conf.legs]
    if sc.get("echo_limiter") or sc.get("volrecv"):
        levels += [h.get_float(leg["volrecv"], VOL_GET_LINEAR) for leg in conf.legs]
    res = {"out": [h.drain(leg["out"]) for leg in conf.legs], "spk": [h.drain(leg["spk"]) for leg in conf.legs], "stats": mid_stats,
           "late": h.P.ms_mi355x_late_events() - late0, "levels": levels}
    conf.close()
    res["after"] = tuple(a - b for a, b in zip(h.runtime_stats(), before))
    return res


SCENARIOS = {
    "plain": {},
    "delay_and_far_gaps": {"delay_ms": 20, "far_gaps": True, "nticks": 150},
    "ptime20": {"ptime20": True, "nticks": 100},
    "odd_pins": {"members": 3, "pins": [0, 5, 9], "nconf": 3, "tail_ms": 64},
    "gain_method": {"events": [(40, "gain", 1, 0.5), (70, "gain", 5, 2.0)], "no_early_launch": True},
    "gain_method_early": {"events": [(40, "gain", 1, 0.5), (70, "gain", 5, 2.0)]},
    "wideband_8k_16k": {"in_rate": 8000, "rate": 16000, "tail_ms": 128, "nticks": 100},
    # the conference as MSAudioConference plumbs it: a forwarding in_resampler in front of every pin, an out_resampler behind it
    "endpoint_resamplers": {"endpoint_resamplers": True, "delay_ms": 10, "far_gaps": True},
    "endpoint_resamplers_no_agc_no_resampler": {"endpoint_resamplers": True, "no_agc": True, "no_resampler": True, "in_rate": 48000, "members": 3, "pins": [1, 4, 6]},
    # an AudioStream's sending side, several streams on one ticker: MSVolume's chunks go on to another filter, there is no mixer
    "no_mixer": {"no_mixer": True, "nconf": 1, "members": 6, "delay_ms": 10, "far_gaps": True},
    "no_mixer_ptime20": {"no_mixer": True, "nconf": 1, "members": 5, "ptime20": True, "nticks": 100},
    # no MSResample in front of the canceller (the card or the decoder runs at its rate already): MSSpeexEC is the head of the leg
    "no_resampler": {"no_resampler": True, "in_rate": 48000, "delay_ms": 10, "far_gaps": True},
    "no_resampler_16k_ptime20": {"no_resampler": True, "in_rate": 16000, "rate": 16000, "ptime20": True, "nticks": 100, "members": 3, "pins": [0, 2, 7]},
    "no_resampler_no_mixer": {"no_resampler": True, "in_rate": 48000, "no_mixer": True, "nconf": 1, "members": 5, "far_gaps": True},
    # MSVolume WITHOUT AGC (the reference's default): it meters and levels the canceller's frames one by one, no 10 ms re-framing
    "no_agc": {"no_agc": True, "gain": 0.7, "delay_ms": 10, "far_gaps": True},
    "no_agc_ptime20_16k": {"no_agc": True, "in_rate": 8000, "rate": 16000, "ptime20": True, "nticks": 100},
    # ... the sending side of a default AudioStream: sound card at the stream's rate -> MSSpeexEC -> MSVolume (meter only) -> encoder
    "no_agc_no_resampler_no_mixer": {"no_agc": True, "no_resampler": True, "in_rate": 48000, "no_mixer": True, "nconf": 1, "members": 5, "far_gaps": True},
    # a default AudioStream with the echo limiter on (audiostream.c:2236-2240): volsend's peer is volrecv, which stands upstream of the
    # canceller's far end -- metered beside the leg (LegBank::vol_peer), its blocks handed on in the walk
    "echo_limiter_no_mixer": {"echo_limiter": True, "no_agc": True, "no_mixer": True, "nconf": 1, "members": 5, "far_gaps": True, "nticks": 150},
    "echo_limiter_agc_20ms": {"echo_limiter": True, "no_mixer": True, "nconf": 1, "members": 4, "delay_ms": 10, "ptime20": True, "nticks": 150},
    "echo_limiter_replumbed": {"echo_limiter": True, "no_agc": True, "no_mixer": True, "nconf": 1, "members": 4, "nticks": 120,
                               "events": [(41, "reattach", 0, 0), (42, "reattach", 0, 0)], "tail_blocks": 1},
    # volrecv is given a gain: no longer a meter only, its leg goes back to the facades (where the canceller starts over: compared up to there)
    "echo_limiter_peer_reconfigured": {"echo_limiter": True, "no_agc": True, "no_mixer": True, "nconf": 1, "members": 4, "nticks": 120,
                                       "events": [(80, "recv_gain", 1, 0.5)], "compare_ticks": 78},
    "echo_limiter_conference_keeps_its_facades": {"echo_limiter": True, "nticks": 60, "expect_unfused": True},
    # the far end as an AudioStream brings it: through volrecv (audiostream.c:1812-1832), a meter only by default -- it hands its blocks on
    # in the walk, so that the far end meets the microphone block of the same walk in the fused canceller; given a gain (the speaker's
    # volume) its blocks come back with the flush, and the leg goes back to its facades, whose queues pair the streams by count
    "far_end_through_volrecv": {"volrecv": True, "far_gaps": True, "delay_ms": 20, "nticks": 110, "events": [(41, "reattach", 0, 0), (70, "recv_gain", 1, 0.5)], "tail_blocks": 1},
    "far_end_through_volrecv_no_mixer": {"volrecv": True, "no_mixer": True, "no_agc": True, "nconf": 1, "members": 5, "ptime20": True, "nticks": 100,
                                         "events": [(61, "recv_gain", 2, 2.0)], "tail_blocks": 1},
    # an AudioStream as audiostream.c:1798-1832 plumbs it, the application's CPU filters included: card at the stream's rate -> ec ->
    # volsend -> dtmfgen_rtp -> ..;  .. -> dtmfgen -> volrecv -> recv_tee -> ec
    "audiostream_16k_with_the_applications_filters": {"volrecv": True, "cpu_filters": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 16000,
                                                      "rate": 16000, "nconf": 1, "members": 6, "far_gaps": True, "nticks": 110,
                                                      "events": [(41, "reattach", 0, 0), (80, "recv_gain", 2, 0.5)], "tail_blocks": 1},
    # ... and a narrow-band G.711 call end to end: packets in through MSUlawDec -> MSGenericPLC (one in 19 lost), the card at 8 kHz, packets out of MSUlawEnc
    "audiostream_8k_g711": {"volrecv": True, "cpu_filters": True, "g711": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000,
                            "nconf": 1, "members": 6, "nticks": 120, "events": [(61, "reattach", 0, 0)], "tail_blocks": 2},
    # ... the same call without losses: both forms add the same tick in either direction -- packets and speaker frames bit for bit
    # (with a loss the fused receiving side conceals in the tick the packet is missing in, as the reference does; the facades one by one a tick later)
    "audiostream_8k_g711_lossless": {"volrecv": True, "cpu_filters": True, "g711": True, "lossless": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000,
                                     "nconf": 1, "members": 6, "nticks": 120, "events": [(61, "reattach", 0, 0)], "tail_blocks": 2},
    # ... with MSAudioFlowControl behind the PLC (AUDIO_STREAM_FEATURE_FLOW_CONTROL) and a drop request in mid-call, A-law, and no dtmfgen_rtp
    # (a telephone-event payload is negotiated, audiostream.c:1396-1404): volsend's frames are ENCODED in the leg's batch
    "audiostream_8k_pcma_flowcontrol_encoder_in_the_leg": {"volrecv": True, "cpu_filters": True, "g711": True, "alaw": True, "lossless": True, "flowcontrol": True, "dtmfgen_rtp": False,
                                                           "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000, "nconf": 1, "members": 5, "nticks": 130,
                                                           "events": [(40, "flow_drop", 1, 20), (41, "flow_drop", 3, 35), (81, "reattach", 0, 0)], "tail_blocks": 2},
    # ... and the reference's DEFAULT features (AUDIO_STREAM_FEATURE_ALL, audiostream.c:1585-1588,1770-1772,1807,1815): an outbound_mixer in front of
    # the encoder and a local_mixer behind the decoder, each with one linked input (no player open): they can only forward, in the walk
    "audiostream_8k_default_features": {"volrecv": True, "cpu_filters": True, "g711": True, "lossless": True, "flowcontrol": True, "dtmfgen_rtp": False, "local_mixer": 1, "outbound_mixer": True,
                                        "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000, "nconf": 1, "members": 5, "nticks": 120,
                                        "events": [(61, "reattach", 0, 0)], "tail_blocks": 2},
    # ... ALL of AUDIO_STREAM_FEATURE_ALL's filters of this plugin: the two equalizers exist too, neither active (audiostream.c:1623-1640,1801,1828) -- they hand their
    # blocks on in the walk and the stream fuses THROUGH them; the speaker's switched on in mid-call sends one stream's leg back to its facades (compared up to there)
    "audiostream_8k_all_features_idle_equalizers": {"volrecv": True, "cpu_filters": True, "g711": True, "lossless": True, "flowcontrol": True, "dtmfgen_rtp": False, "local_mixer": 1,
                                                    "outbound_mixer": True, "idle_equalizers": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000,
                                                    "nconf": 1, "members": 5, "nticks": 120, "events": [(41, "reattach", 0, 0), (90, "spk_eq_active", 2, 1)], "tail_blocks": 2, "compare_ticks": 88},
    # ... without a recording feature there is no recv_tee (audiostream.c:1776-1786): volrecv hands its blocks straight to the idle spk_equalizer -- it is still a meter that
    # passes in the walk (volume_preprocess finds the canceller's far end THROUGH the equalizer) and the stream still fuses
    "audiostream_8k_idle_equalizers_without_recv_tee": {"volrecv": True, "cpu_filters": True, "g711": True, "lossless": True, "flowcontrol": True, "dtmfgen_rtp": False, "local_mixer": 1,
                                                        "outbound_mixer": True, "idle_equalizers": True, "no_recv_tee": True, "no_mixer": True, "no_agc": True, "no_resampler": True,
                                                        "in_rate": 8000, "rate": 8000, "nconf": 1, "members": 4, "nticks": 100, "events": [(41, "reattach", 0, 0)], "tail_blocks": 2},
    # ... with the local player linked (and idle): the local_mixer has two inputs, mixes for the first second and forwards afterwards (audiomixer.c:244-286)
    "audiostream_8k_default_features_local_player_linked": {"volrecv": True, "cpu_filters": True, "g711": True, "lossless": True, "flowcontrol": True, "dtmfgen_rtp": False, "local_mixer": 2,
                                                            "outbound_mixer": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000, "nconf": 1,
                                                            "members": 4, "nticks": 150, "tail_blocks": 2},
    "spk_equalizer_keeps_the_leg_on_its_facades": {"volrecv": True, "spk_equalizer": True, "nticks": 60, "expect_unfused": True},
    "volrecv_with_a_gain_from_the_start": {"volrecv": True, "nticks": 60, "events": [(0, "recv_gain", 1, 0.5)], "tail_blocks": 1},
    # mic_equalizer between MSResample and MSSpeexEC (audiostream.c:1801): it moves into the leg's bank with its gains and its FIR's
    # memory, the up-sampler un-folds from the canceller's launch (resample, equalize, cancel: all on the device)
    "mic_equalizer": {"mic_equalizer": True, "delay_ms": 10, "far_gaps": True, "nticks": 110, "events": [(40, "eq_gain", 1, 3.0), (55, "eq_rate", 1, 48000), (70, "eq_active", 2, 0), (90, "eq_active", 2, 1)]},
    "mic_equalizer_no_mixer_8k_16k": {"mic_equalizer": True, "no_mixer": True, "nconf": 1, "members": 5, "in_rate": 8000, "rate": 16000, "tail_ms": 128, "ptime20": True,
                                      "nticks": 100, "events": [(41, "eq_gain", 3, 0.3)]},
    "mic_equalizer_replumbed_then_leaves": {"mic_equalizer": True, "no_agc": True, "nticks": 110, "events": [(41, "reattach", 0, 0), (42, "reattach", 0, 0), (75, "agc", 1, 1)],
                                            "tail_blocks": 1},
    # a leg / a conference leaves its batch WHILE ATTACHED (a method makes a member stop qualifying): nothing is heard of it -- the
    # canceller's adapted state and queues, the mixer channels' queues and clocks, MSVolume's state and framing go with the filters,
    # and the method meets the walk it preceded in both forms
    "agc_switched_off_midcall": {"ptime20": True, "nticks": 100, "events": [(41, "agc", 1, 0), (61, "agc", 2, 0)], "tail_blocks": 1},
    "bypass_switched_midcall": {"far_gaps": True, "delay_ms": 10, "nticks": 110, "events": [(41, "bypass", 1, True), (60, "bypass", 6, True), (80, "bypass", 1, False)],
                                "tail_blocks": 1},
    "agc_switched_on_midcall_no_mixer": {"no_agc": True, "no_mixer": True, "nconf": 1, "members": 5, "nticks": 100, "events": [(41, "agc", 1, 1)], "tail_blocks": 1},
    "in_resampler_told_to_resample_midcall": {"endpoint_resamplers": True, "nticks": 100, "events": [(41, "in_rs_rate", 1, 16000)], "tail_blocks": 1},
    # the conferences re-plumbed (detached, attached again) while chunks WAIT in the mixer channels -- 20 ms packets leave one there
    # every other tick; the channel's bufferizer outlives the detach (audiomixer.c:64-76,132-135,200-208) and so must the batch's queues
    "replumbed": {"nticks": 100, "events": [(41, "reattach", 0, 0), (70, "reattach", 0, 0)], "tail_blocks": 1},
    "ptime20_replumbed": {"ptime20": True, "nticks": 100, "events": [(41, "reattach", 0, 0), (60, "reattach", 0, 0)], "tail_blocks": 1},
    "ptime20_replumbed_no_early_launch": {"ptime20": True, "nticks": 100, "no_early_launch": True, "events": [(41, "reattach", 0, 0), (42, "reattach", 0, 0), (60, "reattach", 0, 0)],
                                          "tail_blocks": 1},
    # ... eleven times within a quarter of a second (an application adding and removing members one after the other, audioconference.c:322-374):
    # a conference can come back from EVERY re-plumbing with one more chunk waiting (the mixer skips the walk in which no pin delivers,
    # audiomixer.c:244-286, its flow control trims after 5 s, :92-111) -- the batch's queues take what the channels' bufferizers held (LegBank:
    # kLegHeldChunks) and the device queues hold what the host's framing says throughout
    "replumbed_eleven_times": {"nticks": 130, "events": [(41 + 2 * k, "reattach", 0, 0) for k in range(11)], "tail_blocks": 1},
    "ptime20_replumbed_eleven_times": {"ptime20": True, "nticks": 130, "events": [(41 + 2 * k, "reattach", 0, 0) for k in range(11)], "tail_blocks": 1},
    "no_agc_replumbed": {"no_agc": True, "gain": 0.7, "nticks": 100, "events": [(41, "reattach", 0, 0)], "tail_blocks": 1},
    "no_agc_ptime20_16k_replumbed": {"no_agc": True, "in_rate": 8000, "rate": 16000, "ptime20": True, "nticks": 100, "events": [(41, "reattach", 0, 0), (52, "reattach", 0, 0)],
                                     "tail_blocks": 1},
}


def compare(a, b, tail_blocks=0, block=480, ticks=None):
    """fused result a against the facades one by one b: every leg's mix and speaker audio, bit for bit.  tail_blocks: the two
    forms' latencies through a re-plumbing may differ by a tick (the one-by-one mixer is pumped by the flush that brings it blocks):
    the shorter stream must be the other's beginning, short by at most that many blocks at the END of the run"""
    bad = []
    for k in ("out", "spk"):
        for s, (x, y) in enumerate(zip(a[k], b[k])):
            if ticks is not None:   # (only the run's first `ticks` ticks are held equal: see the scenario)
                x, y = x[:ticks * block], y[:ticks * block]
            n = min(len(x), len(y))
            if tail_blocks and abs(len(x) - len(y)) <= tail_blocks * block and n > 0 and np.array_equal(x[:n], y[:n]):
                continue
            if len(x) != len(y) or not np.array_equal(x, y):
                first = int(np.argmax(x[:n] != y[:n])) if n and (x[:n] != y[:n]).any() else n
                bad.append((k, s, len(x), len(y), first))
    return bad


if __name__ == "__main__":
    d = os.path.join(HOST, "double") if "--double" in sys.argv else os.path.join(ROOT, "mediastreamer2_amd")
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(SCENARIOS)
    h = Host(d)
    verdict = {}
    for name in names:
        fused = run(d, True, SCENARIOS[name], h)
        plain = run(d, False, SCENARIOS[name], h)
        sc_ = SCENARIOS[name]
        verdict[name] = {"bad": compare(fused, plain, sc_.get("tail_blocks", 0), sc_.get("rate", 48000) // 100, sc_.get("compare_ticks")), "fused_stats": fused["stats"], "plain_stats": plain["stats"], "late": [fused["late"], plain["late"]],
                         "samples": int(sum(len(x) for x in fused["out"])), "nonzero": bool(any(x.any() for x in fused["out"])),
                         "levels_equal": bool(np.allclose(fused["levels"], plain["levels"], rtol=0, atol=0)), "after": [fused["after"], plain["after"]]}
    print(json.dumps(verdict))
