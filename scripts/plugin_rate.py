"""Dev/measurement tool: end-to-end rate of the PLUGIN path (host buffers in, host buffers out) for N MSResample
filters on one ticker thread: process() staging + one H2D + one launch + one D2H + emit, per 10 ms tick.
  MSMI355X_SLOTS=4096 python scripts/plugin_rate.py 4096"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
os.environ.setdefault("MSMI355X_SLOTS", str(max(256, n)))
import numpy as np  # noqa: E402
from test_gpu_plugin import Host, MS_RESAMPLE_ID, SET_SAMPLE_RATE, SET_OUTPUT_SAMPLE_RATE  # noqa: E402

h = Host()
chains = []
for k in range(n):
    src, rs, snk = h.source(), h.create(MS_RESAMPLE_ID), h.sink()
    h.call_int(rs, SET_SAMPLE_RATE, 16000)
    h.call_int(rs, SET_OUTPUT_SAMPLE_RATE, 48000)
    h.link(src, 0, rs, 0)
    h.link(rs, 0, snk, 0)
    h.S.ms_ticker_attach(h.ticker, src)
    chains.append((src, rs, snk))
x = (np.random.default_rng(0).normal(0, 3000, 160)).astype(np.int16)
nt = 30
for t in range(nt + 5):
    for src, _, _ in chains:
        h.S.ms2shim_source_push(src, x.ctypes.data, x.nbytes)
h.step(5)  # warm-up: pools created, first launches
t0 = time.perf_counter()
h.step(nt)
dt = time.perf_counter() - t0
got = h.S.ms2shim_sink_size(chains[0][2])
print(f"plugin path: {n} MSResample filters, {nt} ticks: {dt / nt * 1e3:.3f} ms per tick, "
      f"{dt / nt / n * 1e6:.3f} us per stream-tick (host staging + PCIe + kernel + emit), sink0 bytes {got}")
