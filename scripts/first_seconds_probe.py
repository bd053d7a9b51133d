"""Dev tool: is the device held once, some seconds into a process's first load?  A small chain (16 384 legs, ticks of ~1.1 ms) is
ticked and timed with HIP events from the moment the process has a context, for `seconds`; every tick that takes more than 3x the
median is printed with the time since the process started and since the first launch.  python scripts/first_seconds_probe.py [seconds]"""
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mediastreamer2_amd as ms
import bench

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ctx = ms.Context(0)
rig = bench.ChainRig(ms, torch, ctx, 16384)
rig.warm()
g1 = [rig.capture([t]) for t in range(rig.RING)]
t_first = time.perf_counter()
v, at = [], []
t = 0
while time.perf_counter() - t_first < seconds:
    ctx.timer_start()
    g1[t % len(g1)].launch()
    v.append(ctx.timer_stop())
    at.append(time.perf_counter())
    t += 1
v = np.array(v)
med = float(np.median(v))
print(f"{len(v)} ticks, median {med:.3f} ms, max {v.max():.3f} ms; imports + context took {t_first - T0:.1f} s")
for i in np.flatnonzero(v > 3 * med):
    print(f"  tick {i}: {v[i]:.2f} ms at {at[i] - T0:.2f} s after the process started, {at[i] - t_first:.2f} s after the first launch")
