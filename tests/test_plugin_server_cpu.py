"""A conference SERVER's members as one device-resident batch (mediastreamer2_amd/host/filters/server_leg.inl) on a box without a
GPU: the same host code -- recognition, staging, the census and the channels' queues on counts, the encoded return legs, the
packing to the encoder's ptime, un-fusing, the tick in flight at a detach -- against the host-memory double of the kernel library
(tests/host/mi_double.cpp, TEST INFRASTRUCTURE).  Every scenario of tests/server_graph.py runs fused and with MSMI355X_NO_FUSE=1
(MSVolume, MSAudioMixer and MSUlawEnc / MSAlawEnc one by one): every member's G.711 packets (or PCM), every listener's tap and every
member's meter must be the same, byte for byte.  tests/test_gpu_plugin_server.py does the same with the real kernels and holds both
to the chain of oracle objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["ulaw_ptime20", "alaw_ptime10_direct", "mixed_laws_and_a_pcm_pin", "packets_of_20ms_in", "late_packets", "a_member_falls_silent",
         "all_but_one_fall_silent", "mute_and_gain", "mute_and_gain_early", "reattach", "agc_switched_on", "wideband_pcm_48k",
         "g711_bridge_packets_of_20ms", "g711_bridge_some_members_pcm", "late_packets_replumbed", "a_lone_contributor_is_heard_even_muted", "g711_endpoints_in_a_16k_conference", "g711_packets_of_20ms_into_a_48k_conference", "wideband_endpoints_in_a_48k_conference", "late_packets_replumbed_no_early_launch", "late_packets_agc_switched_on"]


@pytest.fixture(scope="module")
def verdict():
    r = subprocess.run(["make", "-C", HOST, "all"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "server_graph.py"), "--double"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_scenario_is_listed(verdict):
    assert sorted(verdict) == sorted(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_fused_server_conference_equals_the_facades_one_by_one(verdict, name):
    v = verdict[name]
    assert v["fused_stats"]["legs"] > 0 and v["plain_stats"]["legs"] == 0, v   # the first run really was fused (at mid-run), the second not
    assert v["bad"] == [], v["bad"][:4]
    assert v["nonzero"] and v["bytes"] > 0
    assert v["late"] == [0, 0], "the device queue differed from the host's framing, or a launch failed"
    assert v["after"] == [[0, 0, 0], [0, 0, 0]], "hubs / banks / slots left behind"
    assert v["levels_equal"] and v["meters_equal"]


def test_a_server_conference_is_one_round_and_a_handful_of_launches_per_tick(verdict):
    """fused: one (enqueue, wait, emit) round per tick for the whole hub -- level + queue per block round, mix, one encode per law --
    whatever the number of members; one by one: three rounds per tick (MSVolume, the mixer, the encoders)"""
    v = verdict["ulaw_ptime20"]
    assert v["fused_stats"]["flush_rounds"] <= 62 and v["plain_stats"]["flush_rounds"] >= 150, v
    assert v["fused_stats"]["launches"] <= 5 * 62, v
