"""The plugin's fused call-leg chain (mediastreamer2_amd/host/filters/leg_chain.inl) on a box without a GPU: the SAME host
code -- fusing, the framing state machines run on counts, staging, the slab emit, un-fusing -- against the host-memory
double of the kernel library (tests/host/mi_double.cpp, TEST INFRASTRUCTURE: its canceller passes the microphone through,
its resampler repeats samples; queues, counts and the mix are exact).  Every scenario is run fused and with
MSMI355X_NO_FUSE=1 (the facades one by one) on the same inputs: every leg's mix and speaker audio must be equal bit for bit,
and the device queues must hold what the reference's bufferizers would (MSMI355X_CHECK_LEVELS).  tests/test_gpu_plugin_fused.py
does the same with the real kernels."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


@pytest.fixture(scope="module")
def verdict():
    r = subprocess.run(["make", "-C", HOST, "all"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fused_graph.py"), "--double"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["plain", "delay_and_far_gaps", "ptime20", "odd_pins", "gain_method", "gain_method_early", "wideband_8k_16k", "no_mixer", "no_mixer_ptime20",
                                  "no_resampler", "no_resampler_16k_ptime20", "no_resampler_no_mixer",
                                  "no_agc", "no_agc_ptime20_16k", "no_agc_no_resampler_no_mixer",
                                  "endpoint_resamplers", "endpoint_resamplers_no_agc_no_resampler",
                                  "echo_limiter_no_mixer", "echo_limiter_agc_20ms", "echo_limiter_replumbed", "echo_limiter_peer_reconfigured",
                                  "echo_limiter_conference_keeps_its_facades",
                                  "far_end_through_volrecv", "far_end_through_volrecv_no_mixer", "volrecv_with_a_gain_from_the_start", "spk_equalizer_keeps_the_leg_on_its_facades", "audiostream_16k_with_the_applications_filters", "audiostream_8k_g711",
                                  "audiostream_8k_g711_lossless", "audiostream_8k_pcma_flowcontrol_encoder_in_the_leg", "audiostream_8k_default_features",
                                  "audiostream_8k_default_features_local_player_linked", "audiostream_8k_all_features_idle_equalizers", "audiostream_8k_idle_equalizers_without_recv_tee",
                                  "mic_equalizer", "mic_equalizer_no_mixer_8k_16k", "mic_equalizer_replumbed_then_leaves",
                                  "agc_switched_off_midcall", "bypass_switched_midcall", "agc_switched_on_midcall_no_mixer", "in_resampler_told_to_resample_midcall",
                                  "replumbed", "replumbed_eleven_times", "ptime20_replumbed_eleven_times", "ptime20_replumbed", "ptime20_replumbed_no_early_launch", "no_agc_replumbed", "no_agc_ptime20_16k_replumbed"])
def test_fused_conference_equals_the_facades_one_by_one(verdict, name):
    v = verdict[name]
    unfused = name in ("echo_limiter_conference_keeps_its_facades", "spk_equalizer_keeps_the_leg_on_its_facades")   # (a conference member with an echo limiter: stated in leg_chain.inl)
    assert (v["fused_stats"]["legs"] == 0 if unfused else v["fused_stats"]["legs"] > 0) and v["plain_stats"]["legs"] == 0, v   # the first run really was fused, the second not
    if name == "audiostream_8k_g711":   # with lost packets the fused receiving side conceals in the tick the packet is missing in, as the reference does (the
        # facades one by one a tick later): the far end meets the canceller a tick earlier around every loss.  What is SENT is equal here (the double's
        # canceller passes the microphone through); tests/test_gpu_plugin_fused.py holds this scenario to the oracle chain with the real kernels
        assert [b for b in v["bad"] if b[0] != "spk"] == [], v["bad"][:4]
    else:
        assert v["bad"] == [], v["bad"][:4]
    if name.startswith("audiostream_8k"):   # the receiving side lives in a fused batch too (recv_leg.inl) -- but for a local_mixer with two linked inputs in front of the PLC
        assert v["fused_stats"]["recv_streams"] == (0 if "local_player_linked" in name else v["fused_stats"]["legs"]) and v["plain_stats"]["recv_streams"] == 0
    if "eleven_times" in name:   # every conference is back in its batch after the eleventh re-plumbing (with 20 ms packets they come back with five chunks and more)
        assert v["fused_stats"]["legs"] == 8 and v["fused_stats"]["conferences"] == 2, v["fused_stats"]
    assert v["nonzero"] and v["samples"] > 0
    assert v["late"] == [0, 0], "a device queue differed from the host's framing, or a launch failed"
    assert v["after"] == [[0, 0, 0], [0, 0, 0]], "hubs / banks / slots left behind"
    if "ptime20" not in name and "replumbed" not in name and "reconfigured" not in name and "midcall" not in name and "leaves" not in name and "volrecv" not in name and "audiostream" not in name:  # (a run that ends a tick apart, fused_graph.compare; with 20 ms packets the meter of a fused leg sees its last chunk a tick later: stated in leg_chain.inl)
        assert v["levels_equal"]


def test_one_flush_round_per_tick_and_few_launches(verdict):
    """fused: one (enqueue, wait, emit) round per tick for the whole hub and ~4 launches (far-end push, canceller, its list
    turn-over, volume + mix) whatever the number of legs; one by one: four rounds per tick"""
    v = verdict["plain"]
    assert v["fused_stats"]["flush_rounds"] <= 61 and v["plain_stats"]["flush_rounds"] >= 200, v
    assert v["fused_stats"]["launches"] <= 4 * 62 + 8, v   # (the bank's work for tick t+1 leaves at the end of walk t; a conference that joins mid-walk costs a second batch once)
    # the reference's default AudioStream (both mixers, flow control, the encoder in the leg's batch): ONE round per tick for both directions
    # (stats read at tick 60 of 120, a re-plumbing at 61), where the facades one by one take six
    d = verdict["audiostream_8k_default_features"]
    assert d["fused_stats"]["flush_rounds"] <= 61 and d["plain_stats"]["flush_rounds"] >= 300, d


@pytest.mark.parametrize("paced", [False, True])
def test_plugin_bench_runs_full_legs_through_the_double(verdict, paced):
    """tests/host/plugin_bench (bench.py's plugin_path) against the double: every leg fused, four launches and one flush round per
    ticker and tick, a probe sink that received every tick's mix; paced: the tickers fire on the wall clock's 10 ms grid."""
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HOST, "double"))
    env.pop("MSMI355X_NO_FUSE", None)
    if paced:
        env["PLUGIN_BENCH_PACED"] = "1"
    r = subprocess.run([os.path.join(HOST, "plugin_bench"), os.path.join(HOST, "double", "libmsmi355xfilters.so"), "256", "2", "40", "10"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["paced"] is paced and d["legs"] == 256 and d["fused_legs"] == 256 and d["fused_conferences"] == 8
    assert abs(d["launches_per_tick_and_ticker"] - 4.0) < 0.3 and abs(d["flush_rounds_per_tick_and_ticker"] - 1.0) < 0.1, d
    assert d["probe_sink_blocks"] >= 40 and d["late_events"] == 0
    assert len(d["slow_ticks"]) == 5 and {"cpu_ms", "nvcsw", "nivcsw", "minflt"} <= set(d["slow_ticks"][0])
    if paced:
        assert 9.0 < d["wall_ms_per_tick"] < 12.0, d["wall_ms_per_tick"]   # 40 ticks on the 10 ms grid (+ the 20 ms lead)


@pytest.mark.parametrize("shape", ["", "server dec", "astream default"])
def test_graphs_replumbed_by_an_application_thread_find_their_way_back_into_their_batches(verdict, shape):
    """PLUGIN_BENCH_CHURN: an application thread detaches and attaches one conference (or stream) graph after the other while the tickers run,
    as the reference's callers do (msticker.c:153-221; audioconference.c:322-374) -- with MSMI355X_CHECK_LEVELS the device queues must hold
    what the host's framing says through every re-plumbing, nothing may be dropped, and at the end EVERY leg lives in its batch again
    (a conference that came back with three or more chunks in flight used to be refused at the attach and stayed on its facades for good)."""
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HOST, "double"), PLUGIN_BENCH_CHURN="100", PLUGIN_BENCH_PACED="1", PLUGIN_BENCH_SHAPE=shape,
               MSMI355X_CHECK_LEVELS="1")
    env.pop("MSMI355X_NO_FUSE", None)
    r = subprocess.run([os.path.join(HOST, "plugin_bench"), os.path.join(HOST, "double", "libmsmi355xfilters.so"), "512", "2", "150", "10"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["churn"]["thread"] == "the application's" and d["churn"]["replumbings"] >= 100, d["churn"]
    assert d["fused_legs"] >= d["legs"] - (32 if shape != "astream default" else 1), d   # (at most the graph that is in the application's hands right now)
    assert d["late_events"] == 0, d


@pytest.mark.parametrize("shape", ["", "nors", "noagc", "nomixer", "nors noagc nomixer", "eprs", "server", "server dec", "eq", "eq nomixer noagc", "el nomixer", "astream", "astream default", "server wb", "server dec wb"])
def test_plugin_bench_shapes_fused_equal_one_by_one_by_checksum(verdict, shape):
    """every leg shape the fused chain takes (PLUGIN_BENCH_SHAPE: without MSResample / without AGC / without a conference mixer):
    256 legs x 70 ticks against the double, every leg's mix and speaker audio folded into one number per run -- fused ==
    the facades one by one == staged through device buffers; the device queues hold what the host's framing says throughout"""
    def run(**extra):
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(HOST, "double"), PLUGIN_BENCH_CHECKSUM="1", PLUGIN_BENCH_SHAPE=shape,
                   MSMI355X_CHECK_LEVELS="1", **extra)
        if "MSMI355X_NO_FUSE" not in extra:
            env.pop("MSMI355X_NO_FUSE", None)
        r = subprocess.run([os.path.join(HOST, "plugin_bench"), os.path.join(HOST, "double", "libmsmi355xfilters.so"), "256", "2", "60", "10"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    fused, plain, staged = run(), run(MSMI355X_NO_FUSE="1"), run(MSMI355X_ZERO_COPY="0")
    # ("server": a conference server's remote members -- volrecv -> in_resampler -> pin -> out_resampler -> MSUlawEnc, filters/server_leg.inl)
    assert fused["fused_legs"] == 256 and plain["fused_legs"] == 0 and staged["fused_legs"] == 256
    assert fused["mix_bytes"] == plain["mix_bytes"] > 0
    assert fused["mix_checksum"] == plain["mix_checksum"] == staged["mix_checksum"]
    assert fused["speaker_checksum"] == plain["speaker_checksum"] == staged["speaker_checksum"]
    assert fused["late_events"] == 0 and staged["late_events"] == 0


@pytest.fixture(scope="module")
def glue(verdict):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "conference_glue.py"), "--double"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_msaudioconference_glue_over_fused_legs(glue):
    """tests/conference_glue.py: two conferences driven the way src/voip/audioconference.c drives its mixer -- members plumbed to the
    lowest free pin with the graph detached and attached around it (:198-257,322-345), one leaving from the middle and the next
    joiner taking its pin (:366-374), the loudest member muted and un-muted (:376-388), the active-speaker election over
    MS_VOLUME_GET_MAX every 50 ms (:419-464; bookkeeping and election by oracle/conference.c) -- fused against the facades one
    by one (host-memory double)."""
    g = glue
    assert g["plain_fused_legs_seen"] == 0 and min(g["fused_legs_seen"]) >= 7 and max(g["fused_legs_seen"]) == 8   # fused again after every re-plumbing
    assert g["late"] == 0 and g["plain_late"] == 0 and g["after"] == [0, 0, 0] and g["plain_after"] == [0, 0, 0]
    assert g["pins"] == {"a0": 0, "a1": 1, "a2": -1, "a3": 3, "b0": 0, "b1": -1, "b2": 2, "b3": 3, "b4": 1} and g["sizes"][-1] == [3, 4]
    # the same samples until the graph is first re-plumbed ...
    assert g["differ_before_replumb"] == []
    # ... and afterwards too: BOTH forms deliver the tick in flight at a detach (the first postprocess of the graph flushes the graph:
    # filters.cpp facade_detached), as the reference's synchronous filters have nothing in flight there (msticker.c:197-218)
    assert g["differ_after_replumb"] == []
    for name, (lag, left) in g["lag_after_leave"].items():
        assert lag == 0 and left == 0.0, (name, lag, left)
    # the election: never another winner; the 1 s maxima within 1 dB (their windows open a tick apart)
    assert g["winner_differs"] == [] and g["worst_db_gap"] < 1.0
    assert len(g["polls_differ_before_replumb"]) <= 2, g["polls_differ_before_replumb"]   # (the poll at which a window rolls over)
    a, b = g["winners"]["a"], g["winners"]["b"]
    assert a[10] == 1 and a[20] == 2 and a[30] == 0 and a[50] == 0 and a[63] == 2 and a[-1] == 1      # a1 | muted: a2 | a0's loud period, whose 1 s maximum outlasts it | a2's | a2 gone: a1
    assert a[66] == -1 and g["speakers"]["a"][66] == 2    # right after the re-plumbing the maxima start over: nobody is elected, the speaker stays (:460-464)
    assert b[10] == 1 and b[35] == 3 and b[60] == 1 and b[-1] == 0      # b1 | b3 joined (pin 3) | b3 quiet, b1 gone: b4 on pin 1 | b0's loud period
    # a muted member is not heard, and is reported at the lowest level (:403)
    r = g["a0_mix_rms"]
    assert r["a1_muted"] < 0.5 * r["a1_talking"] and r["a1_back"] > 0.8 * r["a1_talking"] and g["volume_of_muted"] == -120
    # MSVolume's meter reads on across the re-plumbing (struct Volume outlives the detach, msvolume.c:88-118)
    for k in ("a1_meter_across_leave", "a1_meter_across_leave_plain"):
        assert abs(g[k][0] - g[k][1]) < 1.0 and g[k][1] > -30, g[k]
