"""MSAudioConference's glue (src/voip/audioconference.c, mixer mode) over the plugin's filters with the real kernels:
tests/conference_glue.py plays a scripted call -- members joining and leaving two conferences (the conference graph detached and
attached again around every one of them), the loudest member muted, the active speaker elected every 50 ms from
MS_VOLUME_GET_MAX -- through legs of MSResample -> MSSpeexEC -> MSVolume(AGC) -> mixer pin, (1) fused, (2) the facades one by one,
and (3) as the chain of ORACLE objects predicts it (oracle Resampler -> Echo + Preproc with MSSpeexEC's framing -> Volume with its
1 s OrtpExtremum, bookkeeping and election by oracle/conference.c)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conference_glue as cg  # noqa: E402
import fused_graph as fg  # noqa: E402

pytestmark = pytest.mark.gpu
PKG = os.path.join(fg.ROOT, "mediastreamer2_amd")


@pytest.fixture(scope="module")
def runs(oracle):
    import torch  # noqa: F401  (one HIP runtime per process, see mediastreamer2_amd/_lib.py)
    host = fg.Host(PKG)
    fused = cg.run(PKG, True, oracle, host)
    plain = cg.run(PKG, False, oracle, host)
    return fused, plain, cg.verdict(fused, plain)


def test_conference_glue_fused_equals_the_facades_one_by_one(runs):
    fused, plain, g = runs
    assert g["plain_fused_legs_seen"] == 0 and min(g["fused_legs_seen"]) >= 7 and max(g["fused_legs_seen"]) == 8
    assert g["late"] == 0 and g["plain_late"] == 0 and tuple(g["after"]) == (0, 0, 0) and tuple(g["plain_after"]) == (0, 0, 0)
    assert g["pins"] == {"a0": 0, "a1": 1, "a2": -1, "a3": 3, "b0": 0, "b1": -1, "b2": 2, "b3": 3, "b4": 1}
    assert g["differ_before_replumb"] == []            # bit for bit until the graph is first re-plumbed ...
    # ... and after it: both forms deliver the tick in flight at a detach (filters.cpp facade_detached: the first postprocess of a
    # detaching graph flushes that graph through its chain), as the reference's synchronous filters hold nothing at that point
    # (msticker.c:197-218, audioconference.c:322-374) -- the whole call sample for sample
    assert g["differ_after_replumb"] == [], g["differ_after_replumb"][:4]
    for name, (f, p) in g["level_after"].items():
        assert f == p, (name, f, p)
    # the same election throughout; the 1 s maxima apart by at most the step between two chunks' energies at a talker's onset (the fused
    # form meters a chunk when the mixer takes it, MSVolume's facade when it is complete: up to a tick earlier)
    assert g["winner_differs"] == [] and g["worst_db_gap"] < 1.5 and g["worst_db_gap_settling"] < 3.0, (g["worst_db_gap"], g["worst_db_gap_settling"])
    r = g["a0_mix_rms"]
    assert r["a1_muted"] < 0.5 * r["a1_talking"] and r["a1_back"] > 0.8 * r["a1_talking"] and g["volume_of_muted"] == -120
    for k in ("a1_meter_across_leave", "a1_meter_across_leave_plain"):
        assert abs(g[k][0] - g[k][1]) < 1.0 and g[k][1] > -30, g[k]


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_conference_mixes_are_the_oracle_chains(runs, oracle, form):
    """DIRECT: what every member hears over the scripted call -- joins, a leave from the middle, muting, four re-plumbings -- against
    the chain of oracle objects followed by the oracle's mixer with MSAudioMixer's bookkeeping restated (conference_glue.OracleMixer:
    census, per-pin queues, ALWAYS_STREAMOUT, what a detach drops): block for block the same stream (the plugin's comes a tick later;
    a member's very last block is in flight when the test drains, or sits on the link that its leave un-plumbs), within north_star's
    1e-4 RMS of full scale over the whole call and over every second of it."""
    got = (runs[0] if form == "fused" else runs[1])["out"]
    want = {}
    cg.oracle_polls(oracle, mixes=want)
    worst = 0.0
    for name in cg.LEGS:
        x, y = got[name], want[name]
        assert len(y) - len(x) in (0, 480) and len(x) > 70000, (name, len(x), len(y))
        d = (x.astype(np.float64) - y[:len(x)].astype(np.float64)) / 32768.0
        assert np.sqrt(np.mean(d * d)) <= 1e-4, (name, float(np.sqrt(np.mean(d * d))))
        per_s = np.sqrt(np.mean(d[:len(d) // 48000 * 48000].reshape(-1, 48000) ** 2, axis=1))
        worst = max(worst, float(per_s.max()))
        assert np.abs(y[:len(x)].astype(np.int64)).max() > 2000   # (not a comparison of silences)
    assert worst <= 1e-4, worst


@pytest.mark.parametrize("form", ["fused", "one_by_one"])
def test_conference_glue_elects_what_the_oracle_chain_elects(runs, oracle, form):
    """every poll of the call against the oracle's.  The plugin's batches come back a tick after they leave, and the fused form
    meters a chunk when the mixer takes it (one per tick) where MSVolume meters it when it is complete (none, one or two per
    tick): a poll after tick t reads the oracle's meters as of tick t - 1 or t - 2.  Same winner (polls in which the oracle's two
    loudest are within 1 dB of each other or of the -30 dB threshold, or in which the two readings elect differently, left out);
    every audible member's 1 s maximum within 0.5 dB of one of the two readings -- 1.5 dB in the window that opens at a
    re-plumbing, which holds the restarted cancellers' first chunks."""
    got = (runs[0] if form == "fused" else runs[1])["polls"]
    want1, want2 = cg.oracle_polls(oracle, latency=1), cg.oracle_polls(oracle, latency=2)
    assert len(got) == len(want1) == len(want2) == 2 * (cg.NTICKS // cg.POLL_EVERY)
    skipped, checked, worst, worst_window = 0, 0, 0.0, 0.0
    for (t, c, a), (t1, c1, b), (_, _, b2) in zip(got, want1, want2):
        assert (t, c) == (t1, c1) and a["db"].keys() == b["db"].keys(), (t, c)
        settling = any(0 <= t - e < 12 for e in cg.REPLUMBED[c]) or t < 12   # the maxima start over: a chunk's difference in timing decides
        window = any(0 <= t - e < 105 for e in cg.REPLUMBED[c])
        for k, v in b["db"].items():
            if v > -60 and a["db"][k] > -60 and not settling:
                lo, hi = min(v, b2["db"][k]), max(v, b2["db"][k])
                gap = 0.0 if lo <= a["db"][k] <= hi else min(abs(v - a["db"][k]), abs(b2["db"][k] - a["db"][k]))   # (between the two: a chunk in between)
                if window:
                    worst_window = max(worst_window, gap)
                else:
                    worst = max(worst, gap)
        top = sorted(b["db"].values(), reverse=True)
        tight = (len(top) > 1 and top[0] - top[1] < 1.0) or any(abs(v + 30.0) < 1.0 for v in top[:2])
        if settling or tight or b["winner"] != b2["winner"]:
            skipped += 1
            continue
        checked += 1
        assert a["winner"] == b["winner"], (form, t, c, a, b)
    assert checked > 0.75 * len(got), (checked, skipped)
    assert worst < 0.5 and worst_window < 1.5, (worst, worst_window)
    # the speakers over the call: conference a: a1 | a1 muted: a2 | a0's loud period | a2's | a2 gone: a1;  b: b1 | b3 joined on pin 3 |
    # b1 gone and b4 on ITS pin 1 | b0's loud period
    sp = [p["speaker"] for _, c, p in got if c == "a"]
    assert sp[10] == 1 and sp[20] == 2 and sp[30] == 0 and sp[63] == 2 and sp[-1] == 1
    sb = [p["speaker"] for _, c, p in got if c == "b"]
    assert sb[10] == 1 and sb[35] == 3 and sb[60] == 1 and sb[-1] == 0
