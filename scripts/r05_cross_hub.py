"""Cross-hub batching, the device's side (round 5): the chain's tick for N legs as ONE batch on one stream -- what a
shared device batch that the last hub of a tick launches would hand the GPU -- against T batches of N / T legs on T contexts (T HIP
streams, launched back to back from one thread and waited for together) -- what T ticker hubs hand it today.  Graph-replayed ticks,
wall time per tick over whole scene periods; rocprofv3 --kernel-trace --stats around this script gives the kernels' own time.
   python scripts/r05_cross_hub.py <legs> <T> [ticks]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

legs, T = int(sys.argv[1]), int(sys.argv[2])
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 320


def measure(parts):
    ctxs = [ms.Context(0) for _ in range(parts)]
    rigs = [bench.ChainRig(ms, torch, c, legs // parts) for c in ctxs]
    for r in rigs:
        r.warm(32)
    graphs = [[r.capture([t]) for t in range(r.RING)] for r in rigs]   # one graph per tick and rig: a hub launches its own tick
    for c in ctxs:
        c.sync()
    for t in range(32):
        for g in graphs:
            g[t % 16].launch()
    for c in ctxs:
        c.sync()
    per = []
    for t in range(nt):
        t0 = time.perf_counter()
        for g in graphs:
            g[t % 16].launch()
        for c in ctxs:
            c.sync()
        per.append((time.perf_counter() - t0) * 1e3)
    per.sort()
    out = {"legs": sum(r.n for r in rigs), "batches": parts, "legs_per_batch": rigs[0].n, "tick_ms_p50": round(per[len(per) // 2], 4),
           "tick_ms_p99": round(per[int(len(per) * 0.99)], 4), "tick_ms_max": round(per[-1], 4)}
    for g in graphs:
        for x in g:
            x.close()
    for r in rigs:
        r.close()
    for c in ctxs:
        c.close()
    torch.cuda.empty_cache()
    return out


print(json.dumps({"one_batch": measure(1), "per_hub": measure(T)}))
