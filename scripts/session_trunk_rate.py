"""Measurement tool: end-to-end tick rate of mi_session in trunk mode -- G.711 at 8 kHz in and out, loop-back reference:
160 bytes per leg and tick over PCIe instead of 2240.  python scripts/session_trunk_rate.py 65536 81920"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
for n in [int(a) for a in sys.argv[1:]] or (4096, 65536):
    se = ms.Session(ctx, n, in_rate=8000, rate=48000, use_graphs=False, mic_codec=ms.MI_SESSION_PCMA, out_rate=8000,
                    out_codec=ms.MI_SESSION_PCMA, ref_loopback=True, ref_delay_ms=40)
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 256, (n, 80), dtype=np.uint8)
    for _ in range(3):
        m, r = se.acquire()
        m[:] = codes
        se.submit()
    for _ in range(3):
        se.collect()
    K = 60
    worst = 0.0
    t0 = time.perf_counter()
    last = t0
    for t in range(K):
        if se.in_flight() == 3:
            se.collect()
            now = time.perf_counter()
            worst = max(worst, now - last)
            last = now
        se.acquire()
        se.submit()
    while se.in_flight():
        se.collect()
    dt = (time.perf_counter() - t0) / K
    mb = sum(se.tick_bytes()) * n / 1e6
    print(f"trunk session {n} legs: {dt * 1e3:.3f} ms per tick end to end (worst gap between collects {worst * 1e3:.3f} ms), "
          f"{mb:.1f} MB over PCIe per tick, {'fits' if dt < 0.010 else 'EXCEEDS'} the 10 ms tick", flush=True)
    se.close()
