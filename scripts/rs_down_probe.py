"""Dev tool: launch time of the other resampler ratios (generic kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
n = 4096
for (ir, orate) in ((48000, 16000), (48000, 8000), (16000, 8000), (44100, 48000), (8000, 48000), (16000, 48000), (8000, 16000), (32000, 48000), (48000, 32000), (24000, 16000)):
    in_len = ir // 100
    rs = ms.ResamplerBatch(ctx, n, ir, orate)
    x = torch.from_numpy(bench.synth_pcm_batch(n, in_len, ir)).cuda()
    cap = rs.out_capacity(in_len); ostride = (cap + 7) & ~7
    out = torch.zeros((n, ostride), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3): rs.process(x, out=out)
    ctx.sync()
    K = 100
    ctx.capture_begin()
    for _ in range(K): rs.process(x, out=out)
    g = ctx.capture_end(); g.launch(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start(); g.launch(); best = min(best, ctx.timer_stop())
    print(f"{ir}->{orate}: {best / K * 1e3:.2f} us per {n}-stream tick", flush=True)
