"""Dev/measurement tool (GPU box): what keeps the device warm across the idle gap between two PACED ticks?
An MSTicker fires every 10 ms (src/base/msticker.c:419-443,496-515); at the headline count a tick takes ~8.6 ms back to back
and ~9.0 ms paced: the device idles for a millisecond, and the next tick starts slower.  This runs 1000 paced ticks of the
headline chain at a fixed count with different things happening in the gap (on a second stream, started right after a tick's
results are in, ended or outrun by the next tick):
   nothing | one sleeping wavefront | one busy wavefront | a trickle of tiny kernels | a device copy of a few MB every 100 us
and prints p50 / p99 / max per variant, next to the back-to-back series.

  python3 scripts/paced_probe.py 122880"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 122880
nticks = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = ms.Context(0)
conv = bench.Converged(ms, torch, ctx)
head = bench.Headline(ms, torch, ctx, n, 1, 0, None, 0)
head.prepare(16, conv)
head.tick_series(bench.rig_period(head))
side = torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
small = torch.zeros(64, device="cuda")


def paced(fill):
    v = np.empty(nticks)
    nxt = time.perf_counter()
    for t in range(nticks):
        while time.perf_counter() < nxt:
            if fill:
                fill()
        nxt = max(nxt + 0.010, time.perf_counter() - 0.050)
        ctx.timer_start()
        head.g1[t % len(head.g1)].launch()
        v[t] = ctx.timer_stop()
    return v


def sleeping_wave():  # torch.cuda._sleep spins one block for N cycles: ~1.2 ms at a time, re-armed while the host waits
    if side.query():
        with torch.cuda.stream(side):
            torch.cuda._sleep(2_000_000)


def tiny_kernels():
    with torch.cuda.stream(side):
        small.add_(1.0)


last = [0.0]


def trickle_copy(mb):
    def f():
        now = time.perf_counter()
        if now - last[0] >= 100e-6 and side.query():
            last[0] = now
            with torch.cuda.stream(side):
                b[:mb << 20].copy_(a[:mb << 20])
    return f


def copy_continuous():  # the memory system kept as busy as a tick keeps it: 64 MB device copies one behind the other until the tick starts
    if side.query():
        with torch.cuda.stream(side):
            b.copy_(a)
            b.copy_(a)


out = {"streams": head.rig.n, "back_to_back": bench.series_stats(head.tick_series(nticks))}
for name, fill in (("idle_gap", None), ("spinning_wave", sleeping_wave), ("tiny_kernels", tiny_kernels), ("copy_4MB_per_100us", trickle_copy(4)),
                   ("copy_32MB_per_100us", trickle_copy(32)), ("copy_continuous_full_rate", copy_continuous), ("idle_gap_again", None)):
    s = bench.series_stats(paced(fill))
    torch.cuda.synchronize()
    out[name] = {k: s[k] for k in ("p50_ms", "p99_ms", "max_ms", "late")}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
