"""Dev probe: the chain's single ticks one after the other at N legs, from reset or seeded with converged state, with the
cancellers' event counters (foreground updates, background resets, frames) of 64 sampled legs after every tick.
  python scripts/tick_series_probe.py 118784 [ticks] [steady]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 160
steady = len(sys.argv) > 3
ctx = ms.Context(0)
conv = bench.Converged(ms, torch, ctx) if steady else None
rig = bench.ChainRig(ms, torch, ctx, n)
if conv:
    conv.seed(rig)
g1 = [rig.capture([t]) for t in range(rig.RING)]
ids = np.unique(np.linspace(0, rig.n - 1, 64).astype(int))
prev = np.zeros((len(ids), 4))
rows = []
for t in range(nt):
    ctx.timer_start()
    g1[t % rig.RING].launch()
    ms_ = ctx.timer_stop()
    c = np.array([rig.aec.get(int(i), "counters", 4) for i in ids])
    ad = np.mean([rig.aec.get(int(i), "scalars", 16)[8] for i in ids])
    d = c - prev
    prev = c
    rows.append((t, round(ms_, 3), int(d[:, 3].sum()), int(d[:, 0].sum()), int(d[:, 1].sum()), round(float(ad), 2)))
print("tick ms frames fg_updates bg_resets adapted   (64 legs sampled)")
for r in rows:
    print(*r)
v = np.array([r[1] for r in rows])
print(json.dumps({"streams": rig.n, "mean": float(v.mean()), "max": float(v.max()), "argmax": int(v.argmax())}))
