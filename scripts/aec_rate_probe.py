"""Dev tool: canceller launch time at the three frame sizes.  python scripts/aec_rate_probe.py [streams]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for rate, F in ((8000, 64), (16000, 128), (48000, 256)):
    M = (128 * rate // 1000 + F - 1) // F
    N = 2 * F
    aec = ms.AecBatch(ctx, n, rate, frame_size=F, filter_length=128 * rate // 1000)
    mic = torch.from_numpy(bench.synth_pcm_batch(n, F, rate)).cuda()
    ref = torch.from_numpy(bench.synth_pcm_batch(n, F, rate, sigma=2000.0)).cuda()
    out = torch.zeros_like(mic)
    per_frame = 3 * F * 2 + (3 * M * N + (M + 1) * N + N) * 4  # io + W r/w + foreground + X history + newest block
    leg = bench.Leg(ctx, f"aec F={F}", lambda i: aec.process(mic, ref, out=out, flags=ms.MI_AEC_POSTFILTER), 1, n * per_frame, n, "frames")
    K = 10
    g = leg.run(K, 2)
    ctx.sync()
    best = min(leg.timed(K, g) for _ in range(3))
    us = best * 1e3 / K
    print(json.dumps({"rate": rate, "F": F, "M": M, "streams": n, "us_per_launch": round(us, 1), "ns_per_frame": round(us * 1e3 / n, 2),
                      "algorithmic_GBps": round(n * per_frame / us / 1e3, 1), "state_bytes_per_stream": aec.state_bytes(),
                      "legs_per_10ms_tick": int(n * 10000 / us / (rate / 100 / F))}), flush=True)
    del aec, leg, g
    torch.cuda.empty_cache()
