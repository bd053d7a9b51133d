"""The plugin's fused call-leg chain with the real kernels: conferences of  MSResample -> MSSpeexEC -> MSVolume (AGC) ->
MSAudioMixer  built from the facades in the test runtime (tests/fused_graph.py), run fused (one device-resident batch per
hub: the canceller's tick kernel with the resampler folded in + volume-and-mix, mediastreamer2_amd/host/filters/
leg_chain.inl) and with MSMI355X_NO_FUSE=1 (every facade on its own bank -- the path tests/test_gpu_plugin.py and the
tester scenarios hold to the oracle).  Equal bit for bit: every leg's mix, every speaker pin, the meters."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fused_graph as fg  # noqa: E402

pytestmark = pytest.mark.gpu
PKG = os.path.join(fg.ROOT, "mediastreamer2_amd")


@pytest.fixture(scope="module")
def host():
    import torch  # noqa: F401  (one HIP runtime per process, see mediastreamer2_amd/_lib.py)
    return fg.Host(PKG)


@pytest.mark.parametrize("name", list(fg.SCENARIOS))
def test_fused_conference_equals_the_facades_one_by_one(host, name):
    fused = fg.run(PKG, True, fg.SCENARIOS[name], host)
    plain = fg.run(PKG, False, fg.SCENARIOS[name], host)
    sc = fg.SCENARIOS[name]
    assert (fused["stats"]["legs"] == 0 if sc.get("expect_unfused") else fused["stats"]["legs"] > 0) and plain["stats"]["legs"] == 0
    if "eleven_times" in name:   # every conference is back in its batch after the eleventh re-plumbing (they come back with five chunks and more)
        assert fused["stats"]["legs"] == 8 and fused["stats"]["conferences"] == 2, fused["stats"]
    if name.startswith("audiostream_8k"):   # the receiving side lives in a fused batch too (recv_leg.inl) -- but for a local_mixer with two linked inputs in front of the PLC
        assert fused["stats"]["recv_streams"] == (0 if "local_player_linked" in name else fused["stats"]["legs"]) and plain["stats"]["recv_streams"] == 0
    if name == "audiostream_8k_g711":
        # with lost packets the fused receiving side conceals in the tick the packet is missing in, as the reference does; the facades one by one
        # conceal a tick later (their PLC sees a walk's blocks with the next flush), so the far end meets the canceller a tick apart around every loss
        # and the two forms are no longer sample for sample the same: BOTH are held to the oracle chain in test_audiostream_endpoint_is_the_oracle_chain,
        # the lossless call (audiostream_8k_g711_lossless) is held equal bit for bit here
        assert fused["late"] == 0 and plain["late"] == 0 and fused["after"] == (0, 0, 0) and plain["after"] == (0, 0, 0)
        assert all(len(x) == len(y) for x, y in zip(fused["spk"], plain["spk"]))
        return
    assert fg.compare(fused, plain, sc.get("tail_blocks", 0), sc.get("rate", 48000) // 100, sc.get("compare_ticks")) == []
    assert any(x.any() for x in fused["out"]) and sum(len(x) for x in fused["out"]) > 0
    assert fused["late"] == 0 and plain["late"] == 0, "a device queue differed from the host's framing, or a launch failed"
    assert fused["after"] == (0, 0, 0) and plain["after"] == (0, 0, 0)
    if "ptime20" not in name and not sc.get("tail_blocks") and not sc.get("compare_ticks"):
        np.testing.assert_array_equal(fused["levels"], plain["levels"])


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_echo_limiter_leg_is_the_oracles_two_volume_chain(host, oracle, form):
    """DIRECT: a default AudioStream's sending side with the echo limiter on (audiostream.c:1798-1826,2236-2240) against the chain of
    oracle objects: volrecv (oracle.Volume, block by block, msvolume.c:505-513) meters the far end upstream of the canceller; the
    microphone goes Resampler -> MSSpeexEC's framing -> Echo + Preproc -> volsend (oracle.Volume on 10 ms chunks, :480-503) whose echo
    limiter (:201-238) reads volrecv's energy of the SAME tick.  What the leg sends, within north_star's 1e-4 RMS of full scale, and
    the limiter really works in it (the far end's loud period pulls the sent level down)."""
    sc = dict(fg.SCENARIOS["echo_limiter_no_mixer"], members=3, nticks=150, far_gaps=False)
    res = fg.run(PKG, form == "fused", sc, host)
    assert (res["stats"]["legs"] > 0) == (form == "fused")
    F, rate, in_rate, ns, ni, nt = 256, 48000, 16000, 480, 160, sc["nticks"]
    mic, far = fg.scene(3, nt, in_rate, rate, seed=sc.get("seed", 7))
    worst = 0.0
    for s in range(3):
        rs, ec = oracle.Resampler(in_rate, rate), oracle.Echo(F, 128 * rate // 1000, rate)
        pp = oracle.Preproc(F, rate, ec)
        vr, vs = oracle.Volume(rate), oracle.Volume(rate)
        vs.v.has_peer, vs.v.ea_thres, vs.v.force = 1, 0.002, 20.0
        q_mic, q_ref, q_vol = (np.zeros(0, np.int16) for _ in range(3))
        started, sent, gains = False, [], []
        for t in range(nt):
            fb = vr.chunk(far[s, t * ns:(t + 1) * ns])            # volrecv: a meter, the block goes on as it came
            assert np.array_equal(fb, far[s, t * ns:(t + 1) * ns])
            if started:                                          # speexec.c:240-247
                q_ref = np.concatenate([q_ref, fb])
            q_mic = np.concatenate([q_mic, rs.process(mic[s, t * ni:(t + 1) * ni])])
            while len(q_mic) >= F:                               # :256
                fr, q_mic, started = q_mic[:F], q_mic[F:], True
                if len(q_ref) < F:                               # :262-275
                    q_ref = np.concatenate([q_ref, np.zeros(F, np.int16)])
                r, q_ref = q_ref[:F], q_ref[F:]
                q_vol = np.concatenate([q_vol, pp.run(ec.cancel(fr, r))])
            while len(q_vol) >= ns:                              # msvolume.c:480-503
                ch, q_vol = q_vol[:ns], q_vol[ns:]
                sent.append(vs.chunk(ch, peer_energy=vr.v.energy))
                gains.append(vs.v.gain)
        want = np.concatenate(sent)
        got = res["out"][s]
        assert 0 <= len(want) - len(got) <= 2 * ns and len(got) > 60000, (s, len(got), len(want))
        d = (got.astype(np.float64) - want[:len(got)].astype(np.float64)) / 32768.0
        worst = max(worst, float(np.sqrt(np.mean(d * d))))
        assert min(gains) < 0.7 and np.abs(want.astype(np.int64)).max() > 100, (s, min(gains))   # (the limiter pulled the gain down; not a comparison of silences)
    assert worst <= 1e-4, worst


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_mic_equalizer_leg_is_the_oracle_chain(host, oracle, form):
    """DIRECT: a sending leg with a mic_equalizer (audiostream.c:1798-1810: read_resampler -> mic_equalizer -> ec -> volsend) against the
    chain of oracle objects -- Resampler -> Equalizer (one FIR block per microphone block, its memory carried, equalizer.c:256-288) ->
    MSSpeexEC's framing -> Echo + Preproc -> MSVolume without AGC (a meter: the frames leave as they came); a gain set in mid-call meets
    the next walk's block, and so does MS_FILTER_SET_SAMPLE_RATE at the filter's own rate, which makes the response flat again while the
    leg stays in its batch.  What the leg sends, within north_star's 1e-4 RMS of full scale."""
    sc = {"mic_equalizer": True, "no_mixer": True, "no_agc": True, "nconf": 1, "members": 3, "delay_ms": 10, "nticks": 140,
          "events": [(60, "eq_gain", 1, 3.0), (100, "eq_rate", 2, 48000)]}   # (leg 2: MS_FILTER_SET_SAMPLE_RATE at the rate it has -- flat again, equalizer.c:305-309)
    res = fg.run(PKG, form == "fused", sc, host)
    assert (res["stats"]["legs"] > 0) == (form == "fused")
    F, rate, in_rate, ns, ni, nt = 256, 48000, 16000, 480, 160, sc["nticks"]
    mic, far = fg.scene(3, nt, in_rate, rate, seed=sc.get("seed", 7))
    worst = 0.0
    for s in range(3):
        rs, eq, ec = oracle.Resampler(in_rate, rate), oracle.Equalizer(rate), oracle.Echo(F, 128 * rate // 1000, rate)
        pp = oracle.Preproc(F, rate, ec)
        eq.set_gain(1000.0 + 300.0 * s, 2.5, 600.0)
        eq.set_gain(4000.0, 0.4, 1500.0)
        D = 10 * rate // 1000
        q_mic, q_ref = np.zeros(0, np.int16), np.zeros(D, np.int16)   # (the delay line starts with `delay` of zeros, speexec.c:205-208)
        started, sent = False, []
        for t in range(nt):
            if t == 60 and s == 1:
                eq.set_gain(2000.0, 3.0, 800.0)
            if t == 100 and s == 2:
                eq.set_rate(rate)
            if started:
                q_ref = np.concatenate([q_ref, far[s, t * ns:(t + 1) * ns]])
            q_mic = np.concatenate([q_mic, eq.run(rs.process(mic[s, t * ni:(t + 1) * ni]))])
            while len(q_mic) >= F:
                fr, q_mic, started = q_mic[:F], q_mic[F:], True
                if len(q_ref) < D + F:                           # speexec.c:262-275: less than the delay + a frame queued: a frame of silence goes in
                    q_ref = np.concatenate([q_ref, np.zeros(F, np.int16)])
                r, q_ref = q_ref[:F], q_ref[F:]
                sent.append(pp.run(ec.cancel(fr, r)))
        want = np.concatenate(sent)
        got = res["out"][s]
        assert 0 <= len(want) - len(got) <= 3 * ns and len(got) > 60000, (s, len(got), len(want))
        d = (got.astype(np.float64) - want[:len(got)].astype(np.float64)) / 32768.0
        worst = max(worst, float(np.sqrt(np.mean(d * d))))
        assert np.abs(want.astype(np.int64)).max() > 100
    assert worst <= 1e-4, worst


def _audiostream_oracle(oracle, sc, nstreams, form, law=1):
    """A full-duplex narrow-band AudioStream (audiostream.c:1798-1832) as a chain of oracle objects, per stream:
      packets -> g711_decode -> GenericPlcFilter (msgenericplc.c:59-167, the concealer's clock on the ticker's) -> FlowCtl (flowcontrol.c:107-152)
              -> [one tick: the receiving batch's latency] -> dtmfgen -> volrecv (a meter) -> recv_tee -> MSSpeexEC's far end + speaker frames
      microphone -> MSSpeexEC's framing (speexec.c:223-305) -> Echo + Preproc -> volsend (no AGC: the frames as they are) -> what is sent (PCM).
    form: "fused" -- a lost packet is concealed in the tick it is missing in (the reference's timing); "one_by_one" -- the facades' PLC sees a walk's
    blocks with the next flush, so a concealment is decided and delivered a tick later than a received block of the same walk would have been.
    Returns (sent PCM per stream, speaker audio per stream)."""
    F, rate, ns, nt = 64, 8000, 80, sc["nticks"]
    mic, far = fg.scene(nstreams, nt, rate, rate, seed=sc.get("seed", 7))
    drops = {(ev[0], ev[2]): ev[3] for ev in sc.get("events", []) if ev[1] == "flow_drop"}
    sent_all, spk_all = [], []
    for s in range(nstreams):
        codes = oracle.g711_encode(law, far[s])
        plc, fc = oracle.GenericPlcFilter(rate), oracle.FlowCtl()
        ec = oracle.Echo(F, 128 * rate // 1000, rate)
        pp = oracle.Preproc(F, rate, ec)
        arrive = {}   # tick -> blocks that reach the canceller's far end in that walk
        q_mic, q_ref, q_spk = (np.zeros(0, np.int16) for _ in range(3))
        started, sent, spk = False, [], []
        for t in range(nt):
            lost = not sc.get("lossless") and (t + 2 * s) % 19 == 7
            blocks = [] if lost else [oracle.g711_decode(law, codes[t * ns:(t + 1) * ns])]
            if (t, s) in drops:   # MS_AUDIO_FLOW_CONTROL_DROP before walk t: drop_ms out of the next second (flowcontrol.c:199-211), met by the blocks of that walk
                fc.set_target(drops[(t, s)] * rate // 1000, 1000 * rate // 1000)
            made = plc.tick(10 * t, blocks)
            for i, b in enumerate(made):
                concealed = i >= len(blocks)
                late = 2 if (concealed and form == "one_by_one") else 1
                b = fc.process(b) if sc.get("flowcontrol") else b
                if b.size:
                    arrive.setdefault(t + late, []).append(b)
            for b in arrive.pop(t, []):   # speexec.c:239-250: dropped until the microphone has started, then kept twice
                if started:
                    q_ref, q_spk = np.concatenate([q_ref, b]), np.concatenate([q_spk, b])
            q_mic = np.concatenate([q_mic, mic[s, t * ns:(t + 1) * ns]])
            while len(q_mic) >= F:   # :256
                fr, q_mic, started = q_mic[:F], q_mic[F:], True
                if len(q_ref) < F:   # :261-272 (no configured delay): a frame of silence to the speaker and into the delay line
                    q_ref = np.concatenate([q_ref, np.zeros(F, np.int16)])
                    spk.append(np.zeros(F, np.int16))
                else:
                    spk.append(q_spk[:F])
                    q_spk = q_spk[F:]
                r, q_ref = q_ref[:F], q_ref[F:]
                sent.append(pp.run(ec.cancel(fr, r)))
        sent_all.append(np.concatenate(sent))
        spk_all.append(np.concatenate(spk))
    return sent_all, spk_all


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_audiostream_endpoint_is_the_oracle_chain(host, oracle, form):
    """DIRECT: the reference's full-duplex G.711 AudioStream with the application's CPU filters in between (audiostream.c:1798-1832), packets lost
    (one in 19), MSAudioFlowControl asked to drop in mid-call -- against the chain of oracle objects.  Speaker audio bit for bit (decoder, PLC and flow
    control are exact), what is sent within north_star's 1e-4 RMS of full scale as PCM (tapped in front of the encoder), and the packets exactly the G.711
    of that PCM (oracle.g711_encode is pinned against the reference's own g711.c).  Per direction the plugin adds ONE tick: the far end reaches the
    canceller a tick after its packet, the cleaned frames leave a tick after the microphone block (asserted by the chain's own timing: a far end
    that arrived a tick earlier or later would cancel differently)."""
    base = {"volrecv": True, "cpu_filters": True, "g711": True, "flowcontrol": True, "no_mixer": True, "no_agc": True, "no_resampler": True, "in_rate": 8000, "rate": 8000,
            "nconf": 1, "members": 4, "nticks": 150, "events": [(50, "flow_drop", 1, 20), (90, "flow_drop", 2, 30)]}
    pcm_run = fg.run(PKG, form == "fused", dict(base, encoder=False), host)
    pkt_run = fg.run(PKG, form == "fused", base, host)
    assert (pcm_run["stats"]["legs"] > 0) == (form == "fused") and pcm_run["stats"]["recv_streams"] == (4 if form == "fused" else 0)
    assert pcm_run["late"] == 0 and pkt_run["late"] == 0
    sent, spk = _audiostream_oracle(oracle, base, 4, form)
    worst = 0.0
    for s in range(4):
        got_spk, want_spk = pcm_run["spk"][s], spk[s]
        assert 0 <= len(want_spk) - len(got_spk) <= 2 * 80 and len(got_spk) > 11000, (s, len(got_spk), len(want_spk))
        np.testing.assert_array_equal(got_spk, want_spk[:len(got_spk)], err_msg=f"speaker audio of stream {s}")
        assert np.abs(want_spk.astype(np.int64)).max() > 1000
        got, want = pcm_run["out"][s], sent[s]
        assert 0 <= len(want) - len(got) <= 3 * 80 and len(got) > 11000, (s, len(got), len(want))
        d = (got.astype(np.float64) - want[:len(got)].astype(np.float64)) / 32768.0
        worst = max(worst, float(np.sqrt(np.mean(d * d))))
        # the packets: MSUlawEnc's default 20 ms -- exactly the G.711 of the PCM the other run tapped
        codes = pkt_run["out"][s].view(np.uint8)
        np.testing.assert_array_equal(codes, oracle.g711_encode(1, got)[:len(codes)], err_msg=f"packets of stream {s}")
        assert len(codes) >= len(got) - 2 * 160
    assert worst <= 1e-4, worst


def test_the_default_audiostream_adds_one_tick_per_direction(host, oracle):
    """AUDIO_STREAM_FEATURE_ALL's mixers (outbound_mixer in front of the encoder, local_mixer behind the decoder, one linked input each:
    audiostream.c:1585-1588,1770-1772,1807,1815) forward in the walk as the reference's bypass does (audiomixer.c:219-286), the encoder sits in the
    leg's batch: a packet's audio reaches the speaker pin ONE tick after the packet, a microphone block's packet leaves ONE tick after the block
    (+ the canceller's own framing: 64-sample frames out of 80-sample blocks).  Stated in INTEGRATION.md; asserted here on a lossless call."""
    sc = dict(fg.SCENARIOS["audiostream_8k_default_features"], members=3, nticks=60, events=[])
    res = fg.run(PKG, True, sc, host)
    assert res["stats"]["legs"] == 3 and res["stats"]["recv_streams"] == 3   # (one flush round per tick: tests/test_plugin_fused_cpu.py holds the count)
    sent, spk = _audiostream_oracle(oracle, dict(sc, lossless=True), 3, "fused")
    for s in range(3):
        np.testing.assert_array_equal(res["spk"][s], spk[s][:len(res["spk"][s])])
        codes = res["out"][s].view(np.uint8)
        # 60 ticks of 80 samples: 59 reach the canceller's output (one tick of latency), packed to 20 ms packets
        assert 60 * 80 - len(codes) <= 80 + 160 + 64, len(codes)
        want = oracle.g711_decode(1, oracle.g711_encode(1, sent[s]))[:len(codes)].astype(np.float64)
        d = (oracle.g711_decode(1, codes).astype(np.float64) - want) / 32768.0
        assert np.sqrt(np.mean(d * d)) <= 3e-3   # (G.711's own step is ~1e-2 of the sample: a neighbouring code word here and there; the PCM bar is the test above)


def test_the_fused_cancellers_cancel(host):
    """the microphone is the far end through a room: after two seconds a leg's mix of the OTHER legs' cleaned microphones is
    far below what the raw microphones would give (the cancellers converge inside the fused batch as anywhere else)"""
    sc = dict(fg.SCENARIOS["plain"], nconf=1, members=4, nticks=260)
    res = fg.run(PKG, True, sc, host)
    mic, _ = fg.scene(4, 260, 16000, 48000)
    tail = res["out"][0][-48000 // 2:].astype(np.float64)
    raw = mic[1:, -8000:].astype(np.float64).sum(axis=0)
    assert np.sqrt(np.mean(tail ** 2)) < 0.25 * np.sqrt(np.mean(raw ** 2))


def test_a_member_that_stops_qualifying_takes_the_conference_back_to_its_facades(host):
    """MS_ECHO_CANCELLER_SET_BYPASS_MODE on a fused leg: the conference leaves the batch at the next tick and the facades
    carry on one by one (audio keeps flowing; nothing is left behind afterwards)"""
    sc = dict(fg.SCENARIOS["plain"], nconf=1, nticks=80, events=[(30, "bypass", 2, 1)])
    res = fg.run(PKG, True, sc, host)
    assert res["stats"]["conferences"] == 0          # (read at tick 40: un-fused by then)
    n = len(res["out"][0])
    assert n >= 70 * 480 and res["out"][0][-4800:].any() and res["late"] == 0 and res["after"] == (0, 0, 0)


def test_a_forwarding_in_resampler_that_is_told_to_resample_sends_the_conference_back_to_its_facades(host):
    """MSAudioConference's in_resampler in front of every pin (audioconference.c:209-257) forwards while the endpoint runs at the
    conference's rate, and the fused batch looks through it; given another input rate it must resample again: the conference
    leaves the batch at the next tick and the facades carry on."""
    sc = dict(fg.SCENARIOS["endpoint_resamplers"], nconf=1, nticks=80, events=[(30, "in_rs_rate", 2, 44100)])
    res = fg.run(PKG, True, sc, host)
    assert res["stats"]["conferences"] == 0          # (read at tick 40: un-fused by then)
    assert len(res["out"][0]) >= 70 * 480 and res["out"][0][-4800:].any() and res["late"] == 0 and res["after"] == (0, 0, 0)


def test_detach_and_reattach_fuses_again(host):
    sc = dict(fg.SCENARIOS["plain"], nconf=2, nticks=90, events=[(35, "reattach", 0, 0)])
    res = fg.run(PKG, True, sc, host)
    assert res["stats"]["conferences"] == 2          # fused again after the re-attach
    assert res["out"][0][-4800:].any() and res["late"] == 0 and res["after"] == (0, 0, 0)


def test_a_volume_method_on_a_fused_leg_meets_the_next_walks_chunk_in_both_forms(host):
    """MS_VOLUME_SET_GAIN between two ticks meets the NEXT walk's chunk in the reference (msvolume.c:270-276: the filter's next
    process()).  Both forms now do the same: what a method sets waits for the coming flush while the last walk's blocks are still
    waiting for it (Pool::work_waiting / flushed), and in a conference with AGC it also passes by the chunks MSVolume had already
    handed to the mixer's channel (LegBank::v_delay) -- so fused == one by one sample for sample, with the bank's work leaving at the
    end of the walk or with the flush (round 4: up to two ticks apart)."""
    fused = fg.run(PKG, True, fg.SCENARIOS["gain_method_early"], host)
    plain = fg.run(PKG, False, fg.SCENARIOS["gain_method_early"], host)
    ns = 480
    # conference 1 (legs 4..7) has its gain event at tick 70 on leg 5; conference 0 at tick 40 on leg 1
    for s in range(8):
        x, y = fused["out"][s], plain["out"][s]
        assert len(x) == len(y)
        diff = np.flatnonzero(x != y)
        assert diff.size == 0, (s, diff.min() // ns)
        first_event = 40 if s < 4 else 70
        if s not in (1, 5):   # (a leg does not hear itself) the change is heard from the walk the call preceded, not before
            ref = fg.run  # noqa: F841
            assert np.array_equal(x[:(first_event - 1) * ns], y[:(first_event - 1) * ns])
    assert fused["late"] == 0 and plain["late"] == 0


@pytest.mark.parametrize("shape", ["", "nors", "noagc", "nors noagc nomixer", "eprs", "eq", "el nomixer noagc", "astream", "astream default"])
def test_config3_sized_fused_run_equals_the_facades_one_by_one_by_checksum(shape):
    """4 096 full legs (128 conferences of 32, four tickers) for 190 ticks through tests/host/plugin_bench, fused and with the
    facades one by one (MSMI355X_NO_FUSE=1): every leg's mix and every leg's speaker audio, byte for byte and in order, folded
    into one number per run (PLUGIN_BENCH_CHECKSUM) -- the numbers must agree; so must a run that stages through device
    buffers with copy launches (MSMI355X_ZERO_COPY=0).  The size-independent form of test_fused_conference_equals_...
    shape: the leg without MSResample (a 48 kHz microphone), with MSVolume's AGC off (the reference's default), without a
    conference mixer -- "nors noagc nomixer" is the sending side of a default AudioStream; "eq": a mic_equalizer in every leg;
    "el": the echo limiter on, volrecv metered beside the leg; "astream": full-duplex narrow-band G.711 AudioStreams with the
    application's own filters in between (audiostream.c:1798-1832) -- the receiving side (decoder -> PLC) one batch, the sending leg
    another, the encoder behind dtmfgen_rtp on its facade; "astream default": the reference's default features -- local_mixer, flow
    control, outbound_mixer, no dtmfgen_rtp: the encoder inside the sending leg's batch.  (The sources deliver every packet: no
    losses, so the two forms' timing is the same.)"""
    import json
    import subprocess
    host_dir = os.path.join(fg.ROOT, "tests", "host")
    r = subprocess.run(["make", "-C", host_dir, "plugin_bench"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

    def run(**extra):
        env = dict(os.environ, PLUGIN_BENCH_CHECKSUM="1", PLUGIN_BENCH_SHAPE=shape, **extra)
        if "MSMI355X_NO_FUSE" not in extra:
            env.pop("MSMI355X_NO_FUSE", None)
        p = subprocess.run([os.path.join(host_dir, "plugin_bench"), os.path.join(PKG, "libmsmi355xfilters.so"), "4096", "4", "150", "40"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    fused, plain, staged = run(), run(MSMI355X_NO_FUSE="1"), run(MSMI355X_ZERO_COPY="0")
    assert fused["fused_legs"] == 4096 and plain["fused_legs"] == 0 and staged["fused_legs"] == 4096
    assert fused["mix_bytes"] == plain["mix_bytes"] > 4096 * 150 * (70 if shape.startswith("astream") else 900)   # (PCMU packets there: 80 B per leg and tick)
    assert fused["mix_checksum"] == plain["mix_checksum"] == staged["mix_checksum"], (fused["mix_checksum"], plain["mix_checksum"], staged["mix_checksum"])
    assert fused["speaker_checksum"] == plain["speaker_checksum"] == staged["speaker_checksum"]
    assert fused["late_events"] == 0 and plain["late_events"] == 0


@pytest.mark.parametrize("shape", ["", "server dec", "astream default"])
def test_graphs_replumbed_by_an_application_thread_on_the_device(shape):
    """PLUGIN_BENCH_CHURN with the real kernels: an application thread detaches and attaches one conference (or stream) graph after the other
    while four tickers carry 4 096 legs paced at 10 ms (msticker.c:153-221: the reference's threading model) -- the device queues must hold
    what the host's framing says through every re-plumbing (MSMI355X_CHECK_LEVELS), nothing may be dropped or fail, every leg must be back in
    its batch at the end (but for the graph that is in the application's hands when the count is read)."""
    import json
    import subprocess
    host_dir = os.path.join(fg.ROOT, "tests", "host")
    r = subprocess.run(["make", "-C", host_dir, "plugin_bench"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, PLUGIN_BENCH_CHURN="50", PLUGIN_BENCH_PACED="1", PLUGIN_BENCH_SHAPE=shape, MSMI355X_CHECK_LEVELS="1")
    env.pop("MSMI355X_NO_FUSE", None)
    p = subprocess.run([os.path.join(host_dir, "plugin_bench"), os.path.join(PKG, "libmsmi355xfilters.so"), "4096", "4", "200", "20"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["churn"]["thread"] == "the application's" and d["churn"]["replumbings"] >= 100, d["churn"]   # (~400 on a quiet host: 50 a second and ticker for two seconds)
    assert d["fused_legs"] >= d["legs"] - (32 if shape != "astream default" else 1), d
    assert d["late_events"] == 0, d
