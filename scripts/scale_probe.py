"""Dev tool: throughput of the per-tick kernels as the batch grows (are they bandwidth-bound at scale?)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
for n in (4096, 16384, 65536, 262144):
    lg = bench.make_resample_leg(ms, torch, ctx, n)
    K = 50
    g = lg.run(K, 3)
    ctx.sync()
    best = min(lg.timed(K, g) for _ in range(3))
    r = bench.roofline(best, K, lg.alg_bytes)
    print("resample", n, r["avg_launch_us"], "us", r["achieved"], "GB/s", flush=True)
    del lg, g
    torch.cuda.empty_cache()
for nconf in (128, 1024, 4096):
    lg = bench.make_mixer_leg(ms, torch, ctx, nconf=nconf)
    K = 50
    g = lg.run(K, 3)
    ctx.sync()
    best = min(lg.timed(K, g) for _ in range(3))
    r = bench.roofline(best, K, lg.alg_bytes)
    print("mixer", nconf, r["avg_launch_us"], "us", r["achieved"], "GB/s", flush=True)
    del lg, g
    torch.cuda.empty_cache()
for n in (4096, 65536):
    lg = bench.make_volume_leg(ms, torch, ctx, nstreams=n)
    K = 50
    g = lg.run(K, 3)
    ctx.sync()
    best = min(lg.timed(K, g) for _ in range(3))
    r = bench.roofline(best, K, lg.alg_bytes)
    print("volume", n, r["avg_launch_us"], "us", r["achieved"], "GB/s", flush=True)
    del lg, g
    torch.cuda.empty_cache()
