"""Dev/measurement tool: end-to-end rate of the PLUGIN path (host buffers in, host buffers out) for N MSResample
filters spread over T ticker threads: per 10 ms tick every filter's process() staging, then per ticker one upload,
one launch per bank, one download and the emit into the output queues.
  python scripts/plugin_rate.py 50000 4"""
import os
import sys
import ctypes as C
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import numpy as np  # noqa: E402
from test_gpu_plugin import Host, MS_RESAMPLE_ID, SET_SAMPLE_RATE, SET_OUTPUT_SAMPLE_RATE  # noqa: E402

h = Host()
S = h.S
tickers = [S.ms_ticker_new() for _ in range(nthreads)]
chains = [[] for _ in range(nthreads)]
for k in range(n):
    src, rs, snk = h.source(), h.create(MS_RESAMPLE_ID), h.sink()
    h.call_int(rs, SET_SAMPLE_RATE, 16000)
    h.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000)
    h.link(src, 0, rs, 0)
    h.link(rs, 0, snk, 0)
    S.ms2shim_sink_set_discard(C.c_void_p(snk), 1)
    S.ms_ticker_attach(tickers[k % nthreads], src)
    chains[k % nthreads].append((src, rs, snk))
x = (np.random.default_rng(0).normal(0, 3000, 160)).astype(np.int16)
nt = 30
worst = [0.0] * nthreads
series = [[] for _ in range(nthreads)]
total = [0.0] * nthreads
bar = threading.Barrier(nthreads)


def run(i):
    # each ticker thread queues its own sources' blocks (allocated by the thread that will free them, as an RTP receiver
    # on that thread would) and warms its banks up
    for t in range(nt + 5):
        for src, _, _ in chains[i]:
            S.ms2shim_source_push(src, x.ctypes.data, x.nbytes)
    for _ in range(5):
        S.ms_ticker_step(tickers[i])
    bar.wait()
    for _ in range(nt):
        t0 = time.perf_counter()
        S.ms_ticker_step(tickers[i])  # ctypes drops the GIL: the ticker threads really run side by side
        dt = time.perf_counter() - t0
        worst[i] = max(worst[i], dt)
        total[i] += dt
        series[i].append(dt)


th = [threading.Thread(target=run, args=(i,)) for i in range(nthreads)]
for t in th:
    t.start()
for t in th:
    t.join()
wall = max(total)
got = S.ms2shim_sink_blocks(chains[0][0][2])
print(f"plugin path: {n} MSResample filters on {nthreads} ticker thread(s), {nt} ticks: wall {wall / nt * 1e3:.3f} ms per tick, "
      f"per-thread tick mean {max(total) / nt * 1e3:.3f} ms / worst {max(worst) * 1e3:.3f} ms, "
      f"{wall / nt / n * 1e6:.3f} us per stream-tick (host staging + PCIe + kernels + emit), sink0 blocks {got}")
print("thread 0 ticks (ms):", " ".join(f"{v * 1e3:.1f}" for v in series[0]))
med = sorted(v for ser in series for v in ser)[len(series) * nt // 2]
print(f"median tick over all threads: {med * 1e3:.3f} ms")
