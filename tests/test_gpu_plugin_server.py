"""A conference SERVER's members through the plugin with the real kernels (tests/server_graph.py): conferences of REMOTE endpoints --
volrecv -> [in_resampler] -> mixer pin, pin -> [out_resampler] -> MSUlawEnc / MSAlawEnc, no canceller (audioconference.c:121-179,
209-257) -- (1) fused into a ServerBank (filters/server_leg.inl: metered, queued, mixed and ENCODED in one batch), (2) the facades one
by one, and (3) as the chain of ORACLE objects predicts them: oracle.Volume per member (the meter and gain of msvolume.c, no AGC: every
block as it comes), conference_glue.OracleMixer (audiomixer.c's census, queues and arithmetic), oracle.g711_encode -- which is pinned
against the reference's own g711.c (oracle/_ref) -- packed to the encoder's ptime; where MSAlawDec / MSUlawDec of the plugin head the
legs (a G.711 bridge end to end) the oracle decodes the same packets first.  Bit for bit: this path is integer work and MSVolume's
float chain, both bit-exact by north_star."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conference_glue as cg  # noqa: E402
import fused_graph as fg  # noqa: E402
import server_graph as sg  # noqa: E402

pytestmark = pytest.mark.gpu
PKG = os.path.join(fg.ROOT, "mediastreamer2_amd")
NAMES = list(sg.SCENARIOS)


@pytest.fixture(scope="module")
def host():
    import torch  # noqa: F401  (one HIP runtime per process, see mediastreamer2_amd/_lib.py)
    return fg.Host(PKG)


@pytest.fixture(scope="module")
def runs(host):
    return {}


def both(host, runs, name):
    if name not in runs:
        runs[name] = (sg.run(PKG, True, sg.SCENARIOS[name], host), sg.run(PKG, False, sg.SCENARIOS[name], host))
    return runs[name]


@pytest.mark.parametrize("name", NAMES)
def test_fused_server_conference_equals_the_facades_one_by_one(host, runs, name):
    fused, plain = both(host, runs, name)
    assert fused["stats"]["legs"] > 0 and plain["stats"]["legs"] == 0
    assert sg.compare(fused, plain) == []
    assert fused["late"] == 0 and plain["late"] == 0 and fused["after"] == (0, 0, 0) and plain["after"] == (0, 0, 0)
    assert np.array_equal(fused["levels"], plain["levels"]) and np.array_equal(fused["meters"], plain["meters"])
    assert any(x.any() for x in fused["out"])


def oracle_server(oracle, sc_in):
    """the scenario through the chain of oracle objects -> every member's output stream (G.711 bytes or PCM) as run() returns them"""
    sc = dict(nconf=2, members=4, nticks=120, rate=8000, law="u", ptime=0, pcm_pins=(), pins=None, gain=None, decoders=(), endpoint_rate=None)
    sc.update(sc_in)
    conf_rate = sc["rate"]
    rate = sc["endpoint_rate"] or conf_rate   # the endpoints' own rate: sources, MSVolume and the encoders run at it (audioconference.c:209-257)
    n, nt, ns = sc["nconf"] * sc["members"], sc["nticks"], rate // 100
    pins = list(range(sc["members"])) if sc["pins"] is None else list(sc["pins"])
    pcm = sg.signals(n, nt, rate, seed=sc.get("seed", 5))
    # endpoints at another rate than their conference: every member's in_resampler and out_resampler work (msresample.c:122-179), 10 ms at a time
    up = [oracle.Resampler(rate, conf_rate) if rate != conf_rate else None for _ in range(n)]
    down = [oracle.Resampler(conf_rate, rate) if rate != conf_rate else None for _ in range(n)]
    hold = [np.zeros(0, np.int16) for _ in range(n)]   # MSVolume's bufferizer once AGC re-frames to 10 ms chunks (msvolume.c:480-486)
    law_of = lambda k: sc["law"] if sc["law"] in ("a", "u") else ("a" if k % 2 else "u")
    for s in range(n):   # a member whose packets pass MSAlawDec / MSUlawDec: what volrecv sees is the decoded audio (g711.c:113-166,200-255)
        k = s % sc["members"]
        if rate == 8000 and (sc["decoders"] is True or k in sc["decoders"]):
            L = 0 if law_of(k) == "a" else 1
            pcm[s] = oracle.g711_decode(L, oracle.g711_encode(L, pcm[s]))
    vols = [oracle.Volume(rate) for _ in range(n)]
    for v in vols:
        if sc["gain"] is not None:   # MS_VOLUME_SET_GAIN before the attach (msvolume.c:270-276)
            v.v.gain = v.v.target_gain = v.v.static_gain = sc["gain"]
    mixers = [cg.OracleMixer(oracle, conf_rate // 100) for _ in range(sc["nconf"])]
    for c in range(sc["nconf"]):
        for k in range(sc["members"]):
            mixers[c].link(pins[k])
    heard = [[] for _ in range(n)]
    for t in range(nt):
        for ev in sc.get("events", []):
            if ev[0] == t and ev[1] == "mute":
                mixers[ev[2] // sc["members"]].active[pins[ev[2] % sc["members"]]] = not ev[3]
            elif ev[0] == t and ev[1] == "gain":
                v = vols[ev[2]]
                v.v.gain = v.v.target_gain = v.v.static_gain = ev[3]
            elif ev[0] == t and ev[1] == "agc":   # MS_VOLUME_ENABLE_AGC: from the next walk's blocks on MSVolume works on 10 ms chunks, with the AGC's target gain (msvolume.c:172-184,480-503)
                vols[ev[2]].v.agc_enabled = int(ev[3])
            elif ev[0] == t and ev[1] == "reattach":
                for m in mixers:   # mixer_postprocess keeps the channels' queues (audiomixer.c:132-135,200-208), preprocess restarts the clocks; MSVolume lives on
                    m.reattached()
        for c in range(sc["nconf"]):
            arrived = {}
            for k in range(sc["members"]):
                s = c * sc["members"] + k
                blocks = []
                quiet = sc.get("silent") and s in sc["silent"][0] and sc["silent"][1] <= t < sc["silent"][2]
                if quiet:
                    pass
                elif sc.get("ptime20_in"):
                    if (t + s) % 2 == 0:
                        blocks.append(pcm[s, t * ns:(t + 2) * ns])
                elif sc.get("burst") and (t + 5 * s) % 23 == 7:
                    pass
                elif sc.get("burst") and (t + 5 * s) % 23 == 8:
                    blocks += [pcm[s, (t - 1) * ns:t * ns], pcm[s, t * ns:(t + 1) * ns]]
                else:
                    blocks.append(pcm[s, t * ns:(t + 1) * ns])
                if vols[s].v.agc_enabled:   # :480-503: re-framed to 10 ms chunks
                    hold[s] = np.concatenate([hold[s]] + blocks)
                    lev = []
                    while len(hold[s]) >= ns:
                        lev.append(vols[s].chunk(hold[s][:ns]))
                        hold[s] = hold[s][ns:]
                else:
                    lev = [vols[s].chunk(b) for b in blocks]   # :505-512: every block as it is
                x = np.concatenate(lev) if lev else np.zeros(0, np.int16)
                if up[s] is not None and len(x):   # the in_resampler: its input re-framed to 10 ms blocks (resample.inl / msresample.c:122-179 per block)
                    x = np.concatenate([up[s].process(x[i:i + ns]) for i in range(0, len(x), ns)])
                arrived[pins[k]] = x
            for pin, row in mixers[c].tick(10 * t, arrived).items():
                s = c * sc["members"] + pins.index(pin)
                heard[s].append(down[s].process(row) if down[s] is not None else row)
    out = []
    for s in range(n):
        k = s % sc["members"]
        x = np.concatenate(heard[s]) if heard[s] else np.zeros(0, np.int16)
        if rate == 8000 and k not in sc["pcm_pins"]:
            codes = oracle.g711_encode(0 if law_of(k) == "a" else 1, x)
            packet = 80 * (sc["ptime"] // 10 if sc["ptime"] >= 10 else 2)   # alaw.c:56-90: whole packets only
            out.append(codes[:len(codes) // packet * packet])
        else:
            out.append(x)
    return out


RESAMPLED = ("g711_endpoints_in_a_16k_conference", "g711_packets_of_20ms_into_a_48k_conference", "wideband_endpoints_in_a_48k_conference")


@pytest.mark.parametrize("name", [n for n in NAMES if n not in ("all_but_one_fall_silent", "a_lone_contributor_is_heard_even_muted")])
@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_server_conference_is_the_oracle_chains(host, runs, oracle, name, form):
    """Every scenario but the two with a LONE contributor (the reference forwards that pin's blocks, this plugin mixes them -- the stated
    exception -- and the oracle mixer models the plugin's choice, so it would prove nothing) against the chain of oracle objects, bit for
    bit -- the AGC switches included (the oracle's MSVolume re-frames to 10 ms chunks from the walk the call preceded, msvolume.c:480-503).
    Endpoints at another rate than their conference pass two float resamplers (held to the library's order within 1 LSB): there the
    PCM in front of the encoders -- the same call with PCM endpoints -- is held to 1 LSB and 1e-4 RMS of full scale, and the packets to
    exactly the G.711 of that PCM."""
    if name in RESAMPLED:
        sc = sg.SCENARIOS[name]
        members = dict(nconf=2, members=4)
        members.update(sc)
        tap_sc = dict(sc, pcm_pins=tuple(range(members["members"])))
        tap = sg.run(PKG, form == "fused", tap_sc, host)["out"]
        want = oracle_server(oracle, tap_sc)
        got = both(host, runs, name)[0 if form == "fused" else 1]
        er = sc.get("endpoint_rate") or sc.get("rate", 8000)
        for s, (x, y) in enumerate(zip(tap, want)):
            assert 0 <= len(y) - len(x) <= 4 * er // 100 and len(x) > 1000, (name, s, len(x), len(y))
            e = np.asarray(x).astype(np.int64) - y[:len(x)].astype(np.int64)
            assert np.abs(e).max() <= 1 and np.sqrt(np.mean((e / 32768.0) ** 2)) <= 1e-4, (name, s, int(np.abs(e).max()))
            z = got["out"][s]
            if got["laws"][s] and np.asarray(z).dtype == np.uint8:   # an encoded pin: exactly the G.711 of the tapped PCM
                L = 0 if got["laws"][s] == "a" else 1
                assert len(z) > 500 and np.array_equal(z, oracle.g711_encode(L, np.asarray(x))[:len(z)]), (name, s)
        return
    got = both(host, runs, name)[0 if form == "fused" else 1]["out"]
    want = oracle_server(oracle, sg.SCENARIOS[name])
    for s, (x, y) in enumerate(zip(got, want)):
        x = np.asarray(x).view(np.uint8) if y.dtype == np.uint8 else x
        # the plugin's last tick is in flight when the test drains (one block / at most one packet short of the oracle's)
        assert 0 <= len(y) - len(x) <= max(160, 2 * sg.SCENARIOS[name].get("rate", 8000) // 100), (name, s, len(x), len(y))
        assert len(x) > 1000 and np.array_equal(x, y[:len(x)]), (name, s, int(np.argmax(x != y[:len(x)])))


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_g711_endpoints_in_a_wideband_conference_against_the_oracle_chain(host, oracle, form):
    """DIRECT: G.711 endpoints in a 16 kHz conference (audioconference.c:209-257: the endpoints' resamplers work) against the chain of
    oracle objects -- oracle.Volume at 8 kHz -> oracle.Resampler 8k -> 16k -> OracleMixer at 16 kHz -> oracle.Resampler 16k -> 8k ->
    oracle.g711_encode.  FIRST the PCM in front of the encoders (the same conference with PCM endpoints: what the out_resamplers hand
    on): within 1 LSB and 1e-4 RMS of full scale of the oracle chain -- north_star's bar for the float resampler (the library's order
    is held within 1 LSB).  THEN the packets: exactly the G.711 of THAT PCM (the encoder is integer work, pinned against the
    reference's own g711.c) -- and, because a sample that moves by one can cross a G.711 decision level, the oracle's packets except for
    rare neighbouring code words: fewer than 2 % of the bytes."""
    sc = {"rate": 16000, "endpoint_rate": 8000, "law": "mixed", "nconf": 1, "members": 3, "nticks": 120}
    res = sg.run(PKG, form == "fused", sc, host)
    tap = sg.run(PKG, form == "fused", dict(sc, pcm_pins=(0, 1, 2)), host)   # the same call, every endpoint a PCM one: the mixes as the out_resamplers leave them
    assert (res["stats"]["legs"] > 0) == (form == "fused") and (tap["stats"]["legs"] > 0) == (form == "fused")
    n, nt = 3, sc["nticks"]
    pcm = sg.signals(n, nt, 8000, seed=5)
    law_of = lambda k: "a" if k % 2 else "u"
    vols = [oracle.Volume(8000) for _ in range(n)]
    up = [oracle.Resampler(8000, 16000) for _ in range(n)]
    down = [oracle.Resampler(16000, 8000) for _ in range(n)]
    mixer = cg.OracleMixer(oracle, 160)
    for k in range(n):
        mixer.link(k)
    heard = [[] for _ in range(n)]
    for t in range(nt):
        arrived = {k: up[k].process(vols[k].chunk(pcm[k, t * 80:(t + 1) * 80])) for k in range(n)}
        for pin, row in mixer.tick(10 * t, arrived).items():
            heard[pin].append(down[pin].process(row))
    for k in range(n):
        L = 0 if law_of(k) == "a" else 1
        want_pcm = np.concatenate(heard[k])
        got_pcm = np.asarray(tap["out"][k])
        assert 0 <= len(want_pcm) - len(got_pcm) <= 320 and len(got_pcm) > 8000, (k, len(got_pcm), len(want_pcm))
        e = got_pcm.astype(np.int64) - want_pcm[:len(got_pcm)].astype(np.int64)
        assert np.abs(e).max() <= 1, (k, int(np.abs(e).max()))
        assert np.sqrt(np.mean((e / 32768.0) ** 2)) <= 1e-4
        want = oracle.g711_encode(L, want_pcm)
        got = np.asarray(res["out"][k]).view(np.uint8)
        assert 0 <= len(want) - len(got) <= 320 and len(got) > 8000, (k, len(got), len(want))
        np.testing.assert_array_equal(got, oracle.g711_encode(L, got_pcm)[:len(got)])   # the packets ARE the G.711 of the tapped PCM
        assert np.mean(got != want[:len(got)]) < 0.02, (k, float(np.mean(got != want[:len(got)])))
