"""Measurement tool: mi_session at 16 kHz wideband (no resampler: mic and far end both 16 kHz, 128-sample canceller
frames, 128 ms tail), end to end with PCIe.  python scripts/session_rate16k.py 131072 262144"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
for n in [int(a) for a in sys.argv[1:]] or (65536, 262144):
    se = ms.Session(ctx, n, in_rate=16000, rate=16000, use_graphs=False)
    mic = bench.synth_pcm_batch(n, 160, 16000)
    for _ in range(3):
        m, r = se.acquire()
        m[:], r[:] = mic, mic[::-1]
        se.submit()
    for _ in range(3):
        se.collect()
    K = 40
    t0 = time.perf_counter()
    for t in range(K):
        if se.in_flight() == 3:
            se.collect()
        se.acquire()
        se.submit()
    while se.in_flight():
        se.collect()
    dt = (time.perf_counter() - t0) / K
    print(f"16 kHz session, {n} streams: {dt * 1e3:.3f} ms per tick end to end ({n * 960 / 1e6:.1f} MB PCIe per tick), "
          f"{'fits' if dt < 0.010 else 'EXCEEDS'} the 10 ms tick", flush=True)
    se.close()
