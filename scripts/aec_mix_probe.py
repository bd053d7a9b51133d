"""Dev probe: launch time of the canceller's tick kernel at N legs for different per-leg frame counts in one launch:
all two frames, all one, and one leg in eight with one frame placed in different patterns.
  python scripts/aec_mix_probe.py 65536"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
FLAGS = ms.MI_AEC_POSTFILTER if os.environ.get("AEC_PROBE_POST", "1") != "0" else 0  # 0: the canceller without its post-filter tail
ctx = ms.Context(0)
leg = bench.make_aec_leg(ms, torch, ctx, n)
aec, mics, refs, out, two, one = leg.keep
ids = np.arange(n)
rng = np.random.default_rng(1)
pats = {
    "all two": np.full(n, 2),
    "all one": np.full(n, 1),
    "one in 8, s%8==7": np.where(ids % 8 == 7, 1, 2),
    "one in 8, (s+s/8)%8==7": np.where((ids + ids // 8) % 8 == 7, 1, 2),
    "one in 8, random": np.where(rng.random(n) < 0.125, 1, 2),
    "first eighth": np.where(ids < n // 8, 1, 2),
    "last eighth": np.where(ids >= n - n // 8, 1, 2),
    "half one (s%2)": np.where(ids % 2 == 1, 1, 2),
    "half one (first half)": np.where(ids < n // 2, 1, 2),
    "all zero": np.full(n, 0),
}
cnts = {k: torch.from_numpy(c.astype(np.uint8)).cuda() for k, c in pats.items()}
torch.cuda.synchronize()
ts = {k: [] for k in pats}
for rep in range(6):  # patterns interleaved, several passes: clock / thermal drift hits them all alike
    for name, cnt in cnts.items():
        for i in range(2):
            aec.process_frames(mics[i % 4], refs[i % 4], out, cnt, max_frames=2, flags=FLAGS)
        ctx.sync()
        for i in range(4):
            ctx.timer_start()
            aec.process_frames(mics[i % 4], refs[i % 4], out, cnt, max_frames=2, flags=FLAGS)
            ts[name].append(ctx.timer_stop())
for name, c in pats.items():
    v = ts[name]
    print(f"{name:28s} frames/leg {c.mean():.3f}  launch {np.median(v):7.3f} ms  (min {min(v):.3f}, max {max(v):.3f})", flush=True)
