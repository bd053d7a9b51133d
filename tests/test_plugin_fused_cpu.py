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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fused_graph.py"), "--double"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["plain", "delay_and_far_gaps", "ptime20", "odd_pins", "gain_method", "wideband_8k_16k", "no_mixer", "no_mixer_ptime20",
                                  "no_resampler", "no_resampler_16k_ptime20", "no_resampler_no_mixer",
                                  "no_agc", "no_agc_ptime20_16k", "no_agc_no_resampler_no_mixer"])
def test_fused_conference_equals_the_facades_one_by_one(verdict, name):
    v = verdict[name]
    assert v["fused_stats"]["legs"] > 0 and v["plain_stats"]["legs"] == 0, v   # the first run really was fused, the second not
    assert v["bad"] == [], v["bad"][:4]
    assert v["nonzero"] and v["samples"] > 0
    assert v["late"] == [0, 0], "a device queue differed from the host's framing, or a launch failed"
    assert v["after"] == [[0, 0, 0], [0, 0, 0]], "hubs / banks / slots left behind"
    if "ptime20" not in name:  # (with 20 ms packets the meter of a fused leg sees its last chunk a tick later: stated in leg_chain.inl)
        assert v["levels_equal"]


def test_one_flush_round_per_tick_and_few_launches(verdict):
    """fused: one (enqueue, wait, emit) round per tick for the whole hub and ~4 launches (far-end push, canceller, its list
    turn-over, volume + mix) whatever the number of legs; one by one: four rounds per tick"""
    v = verdict["plain"]
    assert v["fused_stats"]["flush_rounds"] <= 61 and v["plain_stats"]["flush_rounds"] >= 200, v
    assert v["fused_stats"]["launches"] <= 4 * 62 + 8, v   # (the bank's work for tick t+1 leaves at the end of walk t; a conference that joins mid-walk costs a second batch once)


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


@pytest.mark.parametrize("shape", ["", "nors", "noagc", "nomixer", "nors noagc nomixer"])
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
    assert fused["fused_legs"] == 256 and plain["fused_legs"] == 0 and staged["fused_legs"] == 256
    assert fused["mix_bytes"] == plain["mix_bytes"] > 0
    assert fused["mix_checksum"] == plain["mix_checksum"] == staged["mix_checksum"]
    assert fused["speaker_checksum"] == plain["speaker_checksum"] == staged["speaker_checksum"]
    assert fused["late_events"] == 0 and staged["late_events"] == 0
