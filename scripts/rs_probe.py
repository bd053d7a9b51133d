"""Dev tool: resampler launch time vs batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
for n in [int(a) for a in sys.argv[1:]] or (4096, 65536, 262144):
    lg = bench.make_resample_leg(ms, torch, ctx, n)
    K = 50
    g = lg.run(K, 3)
    ctx.sync()
    best = min(lg.timed(K, g) for _ in range(3))
    r = bench.roofline(best, K, lg.alg_bytes)
    print("resample", os.environ.get("MSMI355X_ABLATE", "-"), n, r["avg_launch_us"], "us", r["achieved"], "GB/s", flush=True)
    del lg, g
    torch.cuda.empty_cache()
