"""Dev probe: N consecutive single ticks of the steady-state chain; prints the largest ones with their position."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 122880
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = ms.Context(0)
conv = bench.Converged(ms, torch, ctx)
head = bench.Headline(ms, torch, ctx, n, 1, 0, None, 0)
head.prepare(16, conv)
import time
for rep in range(3):
    if rep == 1:
        head.rig.canceller_stats()   # host read-backs: the GPU idles for a while
    if rep == 2:
        time.sleep(0.5)
    v = head.tick_series(nt)
    top = np.argsort(v)[-8:][::-1]
    print(json.dumps({"rep": rep, **bench.series_stats(v), "top": [(int(i), round(float(v[i]), 3)) for i in top]}), flush=True)

# The same series with the host kept two ticks ahead of the GPU (events from torch on the library's stream): a tick's
# interval then starts when the GPU finishes the tick before it, so a host stall between "record" and "launch" -- which
# the series above charges to the tick -- cannot show.  Outliers in both = the GPU's; only above = the host's.
stream = torch.cuda.ExternalStream(ctx.stream)
for rep in range(2):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(nt)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(nt)]
    for t in range(nt):
        e0[t].record(stream)
        head.g1[t % len(head.g1)].launch()
        e1[t].record(stream)
        if t >= 2:
            e1[t - 2].synchronize()
    ctx.sync()
    v = np.array([e0[t].elapsed_time(e1[t]) for t in range(nt)])
    top = np.argsort(v)[-8:][::-1]
    print(json.dumps({"rep": f"host two ticks ahead {rep}", **bench.series_stats(v), "top": [(int(i), round(float(v[i]), 3)) for i in top]}), flush=True)
    v = head.tick_series(nt)
    top = np.argsort(v)[-8:][::-1]
    print(json.dumps({"rep": f"as the bench {rep}", **bench.series_stats(v), "top": [(int(i), round(float(v[i]), 3)) for i in top]}), flush=True)

# Where is the host when a long tick happens?  The bench's loop again with wall-clock stamps: for the longest ticks, the
# GPU time, the wall time of the same iteration and the host's gap before it (the GPU idles for that long); then the
# same with Python's garbage collector off.
import gc
for label in ("gc on", "gc off"):
    if label == "gc off":
        gc.collect()
        gc.freeze()
        gc.disable()
    v = np.empty(nt)
    ta = np.empty(nt)
    tb = np.empty(nt)
    for t in range(nt):
        ta[t] = time.perf_counter()
        ctx.timer_start()
        head.g1[t % len(head.g1)].launch()
        v[t] = ctx.timer_stop()
        tb[t] = time.perf_counter()
    gap = np.concatenate([[0.0], (ta[1:] - tb[:-1]) * 1e3])
    wall = (tb - ta) * 1e3
    top = np.argsort(v)[-6:][::-1]
    print(json.dumps({"rep": label, **bench.series_stats(v), "gap_ms_p50": round(float(np.median(gap)), 4), "gap_ms_max": round(float(gap.max()), 3),
                      "gap_argmax": int(gap.argmax()), "wall_minus_gpu_p50": round(float(np.median(wall - v)), 4),
                      "top": [{"i": int(i), "gpu": round(float(v[i]), 3), "wall": round(float(wall[i]), 3), "gap_before": round(float(gap[i]), 3),
                               "gap_before_prev": round(float(gap[i - 1]), 3) if i else None} for i in top]}), flush=True)
