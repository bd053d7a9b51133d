"""Measurement tool: mi_session as a narrow-band G.711 bridge end to end -- everything at 8 kHz (64-sample canceller frames,
128 ms tail = 16 blocks), PCMA in and out, loop-back reference, PLC.  python scripts/session_nb_rate.py 262144 524288"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mediastreamer2_amd as ms
ctx = ms.Context(0)
for n in [int(a) for a in sys.argv[1:]] or (65536, 262144):
    se = ms.Session(ctx, n, in_rate=8000, rate=8000, use_graphs=False, mic_codec=ms.MI_SESSION_PCMA,
                    out_codec=ms.MI_SESSION_PCMA, ref_loopback=True, ref_delay_ms=40, plc=True)
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 256, (n, 80), dtype=np.uint8)
    lost = rng.random(n) < 0.03
    for _ in range(3):
        m, r = se.acquire()
        m[:] = codes
        se.submit()
    for _ in range(3):
        se.collect()
    K = 40
    t0 = time.perf_counter()
    for t in range(K):
        if se.in_flight() == 3:
            se.collect()
        se.acquire()
        if t % 2:
            se.events()[lost] = ms.MI_PLC_CONCEAL
        se.submit()
    while se.in_flight():
        se.collect()
    dt = (time.perf_counter() - t0) / K
    mb = sum(se.tick_bytes()) * n / 1e6
    print(f"narrow-band bridge {n} legs: {dt * 1e3:.3f} ms per tick end to end, {mb:.1f} MB over PCIe per tick, "
          f"{'fits' if dt < 0.010 else 'EXCEEDS'} the 10 ms tick", flush=True)
    se.close()
