"""Dev tool: are the long ticks of a PACED series the device's or the host's?  Every tick's graph is bracketed by two stamps of the
device's constant-rate clock (mi_debug_stamp, inside the captured graph: the device's own time line from the first launch's start to
the last one's end; a third stamp behind the canceller's launch splits it), and timed with HIP events around the graph launch as bench.py does (which also sees how long the launch took to
reach the device).  python scripts/paced_events_probe.py [legs] [ticks]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mediastreamer2_amd as ms
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 118784
nticks = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = ms.Context(0)
L = ctx.L
L.mi_debug_stamp.argtypes = [C.c_void_p, C.c_void_p]
L.mi_debug_wall_clock_khz.argtypes = [C.c_void_p]
khz = L.mi_debug_wall_clock_khz(ctx.h) or 100000
conv = bench.Converged(ms, torch, ctx)
rig = bench.ChainRig(ms, torch, ctx, n)
conv.seed(rig)
P = rig.RING
ts = torch.zeros(4 * P, dtype=torch.int64, device="cuda")  # per tick: graph start, canceller end, graph end
torch.cuda.synchronize()
graphs = []
for t in range(P):
    ctx.capture_begin()
    L.mi_debug_stamp(ctx.h, ts.data_ptr() + 32 * t)
    rig.tick(t, parts=lambda stage, t=t: stage == "aec_end" and L.mi_debug_stamp(ctx.h, ts.data_ptr() + 32 * t + 8))
    L.mi_debug_stamp(ctx.h, ts.data_ptr() + 32 * t + 16)
    graphs.append(ctx.capture_end())
for t in range(4 * P):
    graphs[t % P].launch()
ctx.sync()
res = {}
for mode in ("back_to_back", "paced"):
    ev, dev, aec = np.empty(nticks), np.empty(nticks), np.empty(nticks)
    nxt = time.perf_counter()
    for t in range(nticks):
        if mode == "paced":
            while time.perf_counter() < nxt:
                pass
            nxt = max(nxt + 0.010, time.perf_counter() - 0.050)
        ctx.timer_start()
        graphs[t % P].launch()
        ev[t] = ctx.timer_stop()
        a, m, b = ts[4 * (t % P):4 * (t % P) + 3].tolist()
        dev[t] = (b - a) / khz
        aec[t] = (m - a) / khz
    top = np.argsort(ev)[-6:][::-1]
    res[mode] = {"events_p50_ms": round(float(np.median(ev)), 4), "events_max_ms": round(float(ev.max()), 4),
                 "device_p50_ms": round(float(np.median(dev)), 4), "device_max_ms": round(float(dev.max()), 4),
                 "submit_p50_ms": round(float(np.median(ev - dev)), 4), "submit_max_ms": round(float((ev - dev).max()), 4),
                 "canceller_p50_ms": round(float(np.median(aec)), 4), "rest_p50_ms": round(float(np.median(dev - aec)), 4),
                 "slowest_by_events [tick, events, device, canceller, rest]":
                 [[int(i), round(float(ev[i]), 3), round(float(dev[i]), 3), round(float(aec[i]), 3), round(float(dev[i] - aec[i]), 3)]
                  for i in top]}
print(json.dumps({"legs": n, "ticks": nticks, "wall_clock_khz": khz, **res}))
