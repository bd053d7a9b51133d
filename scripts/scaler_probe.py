"""Dev probe: the scaler's launch (1080p I420 -> 720p RGB24 / I420) at several batch sizes, against a device copy of the
same bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

ctx = ms.Context(0)
sizes = [int(v) for v in sys.argv[1:]] or [64, 128, 256]
for fmt in (ms.MI_PIX_RGB24, ms.MI_PIX_I420):
    for n in sizes:
        leg = bench.make_scaler_leg(ms, torch, ctx, nframes=n, fmt=fmt)
        steps = 16
        g = leg.run(steps, 4)
        best = min(leg.timed(steps, g) for _ in range(5)) / steps
        nb = leg.alg_bytes
        a = torch.empty(nb // 2, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cp = []
        for _ in range(5):
            e0.record()
            for _ in range(8):
                b.copy_(a)
            e1.record()
            e1.synchronize()
            cp.append(e0.elapsed_time(e1) / 8)
        print(f"{leg.name} {n:4d} frames: {best * 1e3:8.1f} us  {nb / best / 1e9:7.1f} GB/s ({nb / best / 8e9:.3f} of peak)   "
              f"copy of the same bytes: {min(cp) * 1e3:8.1f} us {nb / min(cp) / 1e9:7.1f} GB/s", flush=True)
        del leg, a, b, g
        torch.cuda.empty_cache()
