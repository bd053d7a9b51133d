"""The headline's chain at a FIXED leg count in steady state, for profiling: every launch of every kernel in this process
runs at that count (what `profiles/r03_headline_fixed_kernel_stats.csv` and the counter passes are collected from).

  python3 scripts/headline_probe.py N --state /tmp/conv.npy            # first call: converge the base legs, save their state, exit
  rocprofv3 --kernel-trace --stats ... -- python3 scripts/headline_probe.py N --state /tmp/conv.npy --ticks 64
  rocprofv3 --pmc FETCH_SIZE ...      -- python3 scripts/headline_probe.py N --state /tmp/conv.npy --ticks 16 --settle 16

The converged state of the SCENE_BASE base legs travels through a file (mi_aec_export_state blobs) so that the profiled
process launches nothing at any other leg count: it imports the blobs into a holding batch (copies only) and seeds the
N legs from it on the device (mi_aec_copy_state), settles, then runs --ticks eager ticks."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("streams", type=int)
ap.add_argument("--state", default="/tmp/msmi355x_converged.npy")
ap.add_argument("--ticks", type=int, default=64)
ap.add_argument("--settle", type=int, default=bench.SETTLE_TICKS)
ap.add_argument("--from-reset", action="store_true")
a = ap.parse_args()
ctx = ms.Context(0)
if not a.from_reset and not os.path.exists(a.state):
    conv = bench.Converged(ms, torch, ctx)
    blobs = np.stack([np.frombuffer(conv.base.aec.export_state(s), np.uint8) for s in range(conv.base.n)])
    np.save(a.state, blobs)
    print(json.dumps({"saved": a.state, "legs": int(conv.base.n), "adapted_fraction": conv.adapted_fraction}))
    sys.exit(0)
rig = bench.ChainRig(ms, torch, ctx, a.streams)
if not a.from_reset:
    blobs = np.load(a.state, mmap_mode="r")
    hold = ms.AecBatch(ctx, blobs.shape[0], rig.RATE, frame_size=rig.F, filter_length=128 * rig.RATE // 1000)
    for s in range(blobs.shape[0]):
        hold.import_state(s, np.asarray(blobs[s]).tobytes())
    for first in range(0, rig.n, hold.nstreams):
        rig.aec.copy_state_from(hold, 0, first, min(hold.nstreams, rig.n - first))
    ctx.sync()
    hold.close()
rig.warm(max(rig.RING, a.settle // rig.RING * rig.RING))
per = []
for t in range(a.ticks):
    ctx.timer_start()
    rig.tick(t)
    per.append(ctx.timer_stop())
ad, fg, frames, nleg = rig.canceller_stats()
print(json.dumps({"streams": rig.n, "ticks": a.ticks, "tick_ms_mean": round(float(np.mean(per)), 4), "tick_ms_max": round(float(np.max(per)), 4),
                  "adapted_fraction": ad, "state": "from reset" if a.from_reset else "steady"}))
