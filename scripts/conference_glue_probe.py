"""Dev tool: tests/conference_glue.py's call through the plugin (fused, one by one) against the oracle chain's polls: where the 1 s
maxima differ and by how much.  python scripts/conference_glue_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import oracle
import conference_glue as cg
import fused_graph as fg

PKG = os.path.join(ROOT, "mediastreamer2_amd")
h = fg.Host(PKG)
tf, tp, to = [], [], []
fused, plain = cg.run(PKG, True, oracle, h, trace=tf), cg.run(PKG, False, oracle, h, trace=tp)
want = cg.oracle_polls(oracle, trace=to)
g = cg.verdict(fused, plain)
print("verdict:", {k: g[k] for k in ("worst_db_gap", "worst_db_gap_settling", "winner_differs", "lag_after_leave", "level_after", "polls_differ_before_replumb",
                                     "a1_meter_across_leave", "a1_meter_across_leave_plain", "a0_mix_rms")})
for form, got in (("fused", fused["polls"]), ("one_by_one", plain["polls"])):
    rows = []
    for (t, c, a), (_, _, b) in zip(got, want):
        for k, v in b["db"].items():
            rows.append((abs(v - a["db"][k]), t, c, k, round(a["db"][k], 2), round(v, 2), a["winner"], b["winner"]))
    rows.sort(reverse=True)
    print(form, "largest gaps (|gap|, tick, conf, leg, plugin dB, oracle dB, winners):")
    for r in rows[:25]:
        print("   ", r)
    print(form, "winner differs at:", [(t, c, a["winner"], b["winner"]) for (t, c, a), (_, _, b) in zip(got, want) if a["winner"] != b["winner"]])

print("per-tick meters (MS_VOLUME_GET, dB) around the events: tick, leg, fused, one by one, oracle (same tick), oracle (a tick before)")
for name, ticks in (("b2", range(146, 172)), ("b0", range(146, 160)), ("a0", range(138, 150)), ("a1", range(330, 345))):
    for t in ticks:
        g = lambda tr, tt: round(tr[tt].get(name, float("nan")), 2) if 0 <= tt < len(tr) else None
        print("   ", t, name, g(tf, t), g(tp, t), g(to, t), g(to, t - 1))
