"""GPU-side ablation timing of the AEC kernel (not a test): where does the frame time go?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mediastreamer2_amd as ms

ctx = ms.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rate, F = 48000, 256
rng = np.random.default_rng(0)
mic = torch.from_numpy(rng.normal(0, 3000, (n, F)).astype(np.int16)).cuda()
ref = torch.from_numpy(rng.normal(0, 3000, (n, F)).astype(np.int16)).cuda()
out = torch.zeros_like(mic)
torch.cuda.synchronize()
for tail_ms, flags in ((128, 1), (128, 0), (128, 0x100), (6, 0), (6, 1), (32, 0), (64, 0)):
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=tail_ms * rate // 1000)
    for _ in range(5):
        aec.process(mic, ref, out=out, flags=flags)
    ctx.sync()
    ctx.timer_start()
    K = 30
    for _ in range(K):
        aec.process(mic, ref, out=out, flags=flags)
    ms_ = ctx.timer_stop()
    M = (tail_ms * rate // 1000 + F - 1) // F
    print(f"streams={n} tail={tail_ms}ms M={M} postfilter={flags}: {ms_ / K * 1000:.1f} us/frame-batch", flush=True)
    aec.close()
