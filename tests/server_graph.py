"""A conference SERVER's graphs over the plugin's filters (test runtime: tests/host/ms2shim.c): conferences whose members are REMOTE
endpoints as MSAudioConference plumbs them (src/voip/audioconference.c:121-179,209-257) --

    source (stands for rtprecv -> decoder -> .. -> dtmfgen, audiostream.c:1812-1826) -> MSVolume (volrecv) -> [in_resampler] -> mixer pin k
    mixer pin k -> [out_resampler] -> MSUlawEnc / MSAlawEnc -> sink (stands for rtpsend)        (or straight to a sink: a PCM listener)

-- no echo canceller anywhere.  run() plays one scenario and returns every sink's bytes and every member's meter; used fused
(filters/server_leg.inl: the conference as one device-resident batch, the mixes encoded in it) against MSMI355X_NO_FUSE=1 (the
facades one by one) by tests/test_plugin_server_cpu.py (host-memory double) and tests/test_gpu_plugin_server.py (real kernels, and
against the chain of oracle objects).

    python tests/server_graph.py [--double] [scenario ..]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fused_graph as fg  # noqa: E402

IDS = fg.IDS
MS_ULAW_ENC_ID, MS_ULAW_DEC_ID, MS_ALAW_ENC_ID, MS_ALAW_DEC_ID = 7, 8, 9, 10
SET_RATE, SET_OUT_RATE = IDS["MS_FILTER_SET_SAMPLE_RATE"], IDS["MS_FILTER_SET_OUTPUT_SAMPLE_RATE"]
ADD_FMTP = IDS["MS_FILTER_ADD_FMTP"]
MIX_SET_ACTIVE = fg.mid(fg.MS_AUDIO_MIXER_ID, 1, 8)


class MixerCtl(C.Structure):
    _fields_ = [("pin", C.c_int), ("active", C.c_int)]


class ServerConferences:
    """nconf conferences of `members` remote endpoints each on one ticker"""

    def __init__(self, h, nconf, members, rate=8000, law="u", ptime=0, endpoint_resamplers=True, pcm_pins=(), pins=None, gain=None, listener=False,
                 decoders=(), endpoint_rate=None):
        self.h, self.S = h, h.S
        S = h.S
        self.ticker = S.ms_ticker_new()
        self.rate, self.nconf, self.members = rate, nconf, members
        erate = endpoint_rate or rate   # the endpoints' own rate: other than the conference's, their resamplers work (audioconference.c:209-257)
        self.erate = erate
        self.pins = list(range(members)) if pins is None else list(pins)
        self.legs, self.mixers, self.extra = [], [], []
        for c in range(nconf):
            mx = S.ms_factory_create_filter(h.fac, fg.MS_AUDIO_MIXER_ID)
            h.call_int(mx, SET_RATE, rate)
            h.call_int(mx, fg.MIX_CONF_MODE, 1)
            self.mixers.append(mx)
            for k in range(members):
                pin = self.pins[k]
                leg = {"src": self.new_source(), "vol": S.ms_factory_create_filter(h.fac, fg.MS_VOLUME_ID), "out": S.ms2shim_new_sink(h.fac),
                       "mixer": mx, "pin": pin, "enc": None, "dec": None}
                leg["law"] = law if law in ("a", "u") else ("a" if k % 2 else "u")   # "mixed": alternate
                if erate == 8000 and (decoders is True or k in decoders):   # the source hands over G.711 packets (rtprecv): MSAlawDec / MSUlawDec of the plugin in front of volrecv
                    leg["dec"] = S.ms_factory_create_filter(h.fac, MS_ALAW_DEC_ID if leg["law"] == "a" else MS_ULAW_DEC_ID)
                h.call_int(leg["vol"], SET_RATE, erate)
                if gain is not None:
                    h.call_float(leg["vol"], fg.VOL_SET_GAIN, gain)
                links = []
                head, tail = (leg["vol"], 0), None
                if endpoint_resamplers:   # audioconference.c:209-257: both at the conference's rate here -- they forward (msresample.c:126-135)
                    leg["in_rs"], leg["out_rs"] = (S.ms_factory_create_filter(h.fac, fg.MS_RESAMPLE_ID) for _ in range(2))
                    h.call_int(leg["in_rs"], SET_RATE, erate), h.call_int(leg["in_rs"], SET_OUT_RATE, rate)
                    h.call_int(leg["out_rs"], SET_RATE, rate), h.call_int(leg["out_rs"], SET_OUT_RATE, erate)
                    links += [(leg["vol"], 0, leg["in_rs"], 0), (leg["in_rs"], 0, mx, pin), (mx, pin, leg["out_rs"], 0)]
                    tail = (leg["out_rs"], 0)
                else:
                    links += [(leg["vol"], 0, mx, pin)]
                    tail = (mx, pin)
                if erate == 8000 and k not in pcm_pins:
                    this_law = leg["law"]
                    leg["enc"] = S.ms_factory_create_filter(h.fac, MS_ALAW_ENC_ID if this_law == "a" else MS_ULAW_ENC_ID)
                    if ptime:
                        fmtp = f"ptime={ptime}".encode()
                        assert S.ms_filter_call_method(leg["enc"], ADD_FMTP, C.c_char_p(fmtp)) == 0
                    links += [(tail[0], tail[1], leg["enc"], 0), (leg["enc"], 0, leg["out"], 0)]
                else:
                    links += [(tail[0], tail[1], leg["out"], 0)]
                links = ([(leg["src"], 0, leg["dec"], 0), (leg["dec"], 0, leg["vol"], 0)] if leg["dec"] else [(leg["src"], 0, leg["vol"], 0)]) + links
                for a, pa, b, pb in links:
                    assert S.ms_filter_link(a, pa, b, pb) == 0, (a, pa, b, pb)
                self.legs.append(leg)
            if listener:   # an output-only pin above the members (a recorder's tap): it hears everybody
                tap = S.ms2shim_new_sink(h.fac)
                assert S.ms_filter_link(mx, max(self.pins) + 2, tap, 0) == 0
                self.extra.append(tap)
        self.attached = False

    def new_source(self):
        """an RTP receiver / decoder: it hands on everything it has in a walk (two packets after a late one)"""
        src = self.S.ms2shim_new_source(self.h.fac)
        self.S.ms2shim_source_set_burst(src, 1)
        return src

    def attach(self):
        for f in self.mixers:
            self.S.ms_ticker_attach(self.ticker, f)
        self.attached = True

    def detach(self):
        for f in self.mixers:
            self.S.ms_ticker_detach(self.ticker, f)
        self.attached = False

    def step(self):
        self.S.ms_ticker_step(self.ticker)

    def close(self):
        if self.attached:
            self.detach()
        for leg in self.legs:
            for k in ("src", "vol", "out", "in_rs", "out_rs", "enc", "dec"):
                if leg.get(k):
                    self.S.ms_filter_destroy(leg[k])
        for f in self.mixers + self.extra:
            self.S.ms_filter_destroy(f)
        self.S.ms_ticker_destroy(self.ticker)


def signals(nlegs, nticks, rate, seed=5):
    """every member's decoded audio: a talker whose loudness changes over the call (so that the census and the election see silences)"""
    rng = np.random.default_rng(seed)
    ns = rate // 100
    t = np.arange(nticks * ns)
    out = np.zeros((nlegs, nticks * ns), np.int16)
    for s in range(nlegs):
        env = np.repeat(rng.choice([200.0, 1500.0, 6000.0, 14000.0], size=nticks // 20 + 1), 20 * ns)[:nticks * ns]
        x = env * (0.5 * rng.normal(0, 1, nticks * ns) + np.sin(2 * np.pi * (200 + 37 * s) * t / rate))
        out[s] = x.round().clip(-32767, 32767).astype(np.int16)
    return out


def run(plugin_dir, fuse, scenario, h=None):
    if fuse:
        os.environ.pop("MSMI355X_NO_FUSE", None)
    else:
        os.environ["MSMI355X_NO_FUSE"] = "1"
    os.environ.pop("MSMI355X_NO_EARLY_LAUNCH", None)
    if scenario.get("no_early_launch"):
        os.environ["MSMI355X_NO_EARLY_LAUNCH"] = "1"
    os.environ["MSMI355X_CHECK_LEVELS"] = "1"
    h = h or fg.Host(plugin_dir)
    sc = dict(nconf=2, members=4, nticks=120, rate=8000, law="u", ptime=0, endpoint_resamplers=True, pcm_pins=(), pins=None, gain=None, listener=False,
              decoders=(), endpoint_rate=None)
    sc.update(scenario)
    conf = ServerConferences(h, sc["nconf"], sc["members"], sc["rate"], sc["law"], sc["ptime"], sc["endpoint_resamplers"], sc["pcm_pins"], sc["pins"], sc["gain"],
                             sc["listener"], sc["decoders"], sc["endpoint_rate"])
    n, nt, ns = sc["nconf"] * sc["members"], sc["nticks"], conf.erate // 100   # (the sources run at the endpoints' rate)
    pcm = signals(n, nt, conf.erate, seed=sc.get("seed", 5))
    codes = {}
    if any(leg["dec"] for leg in conf.legs):   # what the endpoints send: their audio as G.711 (the oracle's encoder, pinned against the reference's g711.c)
        import oracle
        oracle.build()
        codes = {s: oracle.g711_encode(0 if leg["law"] == "a" else 1, pcm[s]) for s, leg in enumerate(conf.legs) if leg["dec"]}

    def push(leg, s, lo, hi):
        a = np.ascontiguousarray(codes[s][lo:hi] if leg["dec"] else pcm[s, lo:hi])
        h.S.ms2shim_source_push(leg["src"], a.ctypes.data, a.nbytes)
    late0, before = h.P.ms_mi355x_late_events(), h.runtime_stats()
    conf.attach()
    mid_stats, meters = None, []
    for t in range(nt):
        for s, leg in enumerate(conf.legs):
            quiet = sc.get("silent") and s in sc["silent"][0] and sc["silent"][1] <= t < sc["silent"][2]   # the endpoint sends nothing (DTX, a hole in the network)
            if quiet:
                continue
            if sc.get("ptime20_in"):   # 20 ms packets: a block of two ticks every other tick
                if (t + s) % 2 == 0:
                    push(leg, s, t * ns, (t + 2) * ns)
            elif sc.get("burst") and (t + 5 * s) % 23 == 7:
                continue                                                         # a late packet ...
            elif sc.get("burst") and (t + 5 * s) % 23 == 8:
                push(leg, s, (t - 1) * ns, t * ns)                               # ... arrives with the next one: two blocks in one tick
                push(leg, s, t * ns, (t + 1) * ns)
            else:
                push(leg, s, t * ns, (t + 1) * ns)
        for ev in sc.get("events", []):
            if ev[0] != t:
                continue
            kind, s, val = ev[1], ev[2], ev[3]
            leg = conf.legs[s]
            if kind == "gain":
                h.call_float(leg["vol"], fg.VOL_SET_GAIN, val)
            elif kind == "agc":
                h.call_int(leg["vol"], fg.VOL_ENABLE_AGC, val)
            elif kind == "mute":
                ctl = MixerCtl(leg["pin"], 0 if val else 1)
                assert h.S.ms_filter_call_method(leg["mixer"], MIX_SET_ACTIVE, C.byref(ctl)) == 0
            elif kind == "reattach":
                conf.detach()
                conf.attach()
        conf.step()
        if t == nt // 2:
            mid_stats = h.fused_stats()
        if t % 10 == 9:
            meters.append([h.get_float(leg["vol"], IDS["MS_VOLUME_GET_MAX"]) for leg in conf.legs])
    res = {"out": [h.drain(leg["out"]).view(np.uint8) if leg["enc"] else h.drain(leg["out"]) for leg in conf.legs],
           "taps": [h.drain(t_) for t_ in conf.extra], "stats": mid_stats, "late": h.P.ms_mi355x_late_events() - late0,
           "levels": [h.get_float(leg["vol"], IDS["MS_VOLUME_GET_LINEAR"]) for leg in conf.legs], "meters": meters,
           "laws": [leg.get("law") for leg in conf.legs], "decoded": sorted(codes)}
    conf.close()
    res["after"] = tuple(a - b for a, b in zip(h.runtime_stats(), before))
    res["pcm"] = pcm
    return res


SCENARIOS = {
    "ulaw_ptime20": {},                                                      # G.711 endpoints, the encoder's default 20 ms packets
    "alaw_ptime10_direct": {"law": "a", "ptime": 10, "endpoint_resamplers": False, "members": 3, "pins": [0, 2, 5], "nconf": 3},
    "mixed_laws_and_a_pcm_pin": {"law": "mixed", "pcm_pins": (1,), "listener": True, "gain": 0.8},
    "packets_of_20ms_in": {"ptime20_in": True, "nticks": 100},
    "late_packets": {"burst": True, "nticks": 150},
    "a_member_falls_silent": {"silent": ((1, 2), 30, 70), "nticks": 200},    # one member quiet for 400 ms (census: contributes for a second), then ...
    "all_but_one_fall_silent": {"silent": ((1, 2, 3, 5, 6, 7), 20, 150), "nticks": 200},  # ... more than a second: a lone contributor, then nobody left but it
    # a lone contributor is forwarded AS IT IS by the reference's bypass mode (audiomixer.c:219-242): muted or not, whatever its input gain
    "a_lone_contributor_is_heard_even_muted": {"silent": ((1, 2, 3, 5, 6, 7), 20, 170), "nticks": 200,
                                               "events": [(100, "mute", 0, True), (130, "mute", 0, False), (150, "mute", 4, True)]},
    "mute_and_gain": {"events": [(30, "mute", 1, True), (60, "mute", 1, False), (45, "gain", 2, 0.25)], "no_early_launch": True},
    "mute_and_gain_early": {"events": [(30, "mute", 1, True), (60, "mute", 1, False)]},
    "reattach": {"events": [(41, "reattach", 0, 0), (77, "reattach", 0, 0)]},  # 41: half a 20 ms packet is filled when the graph is re-plumbed
    "agc_switched_on": {"events": [(50, "agc", 2, 1)]},                      # the conference leaves its batch and carries on one by one
    "wideband_pcm_48k": {"rate": 48000, "nticks": 80},                       # no encoder runs at this rate: every pin gets its PCM from the slab
    # a G.711 bridge end to end: the endpoints' PACKETS in (MSUlawDec / MSAlawDec of the plugin head the legs), packets out; 80 + 80 bytes per member and tick
    "g711_bridge_packets_of_20ms": {"decoders": True, "ptime20_in": True, "law": "mixed", "nticks": 100},
    # late packets leave a block waiting in a mixer channel: it outlives a re-plumbing (audiomixer.c:64-76,132-135,200-208), and the
    # conference leaving its batch while attached (a member's AGC switched on)
    "late_packets_replumbed": {"burst": True, "nticks": 140, "events": [(50, "reattach", 0, 0), (51, "reattach", 0, 0), (90, "reattach", 0, 0)]},
    "late_packets_replumbed_no_early_launch": {"burst": True, "nticks": 120, "no_early_launch": True, "events": [(50, "reattach", 0, 0), (51, "reattach", 0, 0)]},
    "late_packets_agc_switched_on": {"burst": True, "nticks": 140, "events": [(50, "agc", 2, 1)]},
    # G.711 endpoints in a WIDEBAND conference (audioconference.c:209-257: the endpoints' resamplers work): levelled at 8 kHz, up-sampled in
    # the batch, mixed at 16 kHz, every pin's mix down-sampled and encoded -- the resamplers' states move into the bank and back
    "g711_endpoints_in_a_16k_conference": {"rate": 16000, "endpoint_rate": 8000, "law": "mixed", "listener": True, "nticks": 140,
                                           "events": [(40, "mute", 1, True), (60, "reattach", 0, 0), (61, "reattach", 0, 0), (90, "agc", 2, 1)]},
    # wide-band endpoints (a CPU codec at 16 kHz) in a 48 kHz conference: no encoder of ours behind the out_resampler -- its PCM at 16 kHz comes from the batch
    "wideband_endpoints_in_a_48k_conference": {"rate": 48000, "endpoint_rate": 16000, "listener": True, "nticks": 110,
                                               "events": [(30, "mute", 1, True), (50, "reattach", 0, 0), (70, "gain", 5, 0.5)]},
    "g711_packets_of_20ms_into_a_48k_conference": {"rate": 48000, "endpoint_rate": 8000, "decoders": True, "ptime20_in": True, "nticks": 100},
    "g711_bridge_some_members_pcm": {"decoders": (0, 2), "burst": True, "nticks": 140, "events": [(60, "reattach", 0, 0)]},
}


def compare(a, b):
    bad = []
    for k in ("out", "taps"):
        for s, (x, y) in enumerate(zip(a[k], b[k])):
            if len(x) != len(y) or not np.array_equal(x, y):
                n = min(len(x), len(y))
                first = int(np.argmax(x[:n] != y[:n])) if n and (x[:n] != y[:n]).any() else n
                bad.append((k, s, len(x), len(y), first))
    return bad


if __name__ == "__main__":
    d = os.path.join(fg.HOST, "double") if "--double" in sys.argv else os.path.join(ROOT, "mediastreamer2_amd")
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(SCENARIOS)
    h = fg.Host(d)
    verdict = {}
    for name in names:
        fused, plain = run(d, True, SCENARIOS[name], h), run(d, False, SCENARIOS[name], h)
        verdict[name] = {"bad": compare(fused, plain), "fused_stats": fused["stats"], "plain_stats": plain["stats"], "late": [fused["late"], plain["late"]],
                         "bytes": int(sum(len(x) for x in fused["out"])), "nonzero": bool(any(x.any() for x in fused["out"])),
                         "levels_equal": bool(np.array_equal(fused["levels"], plain["levels"])), "meters_equal": bool(np.array_equal(fused["meters"], plain["meters"])),
                         "after": [fused["after"], plain["after"]]}
    print(json.dumps(verdict))
