"""Dev/measurement tool: PCIe cost of feeding the chained path from host buffers, per 10 ms tick:
H2D mic (16 kHz, 320 B) + far-end reference (48 kHz, 960 B) per stream, D2H the mixed output (960 B per stream)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import mediastreamer2_amd as ms
ctx = ms.Context(0)
L = ctx.L
for n in [int(a) for a in sys.argv[1:]] or (4096, 65536):
    up, down = n * (320 + 960), n * 960
    h_up, h_dn = L.mi_host_alloc(ctx.h, up), L.mi_host_alloc(ctx.h, down)
    d_up, d_dn = L.mi_dev_alloc(ctx.h, up), L.mi_dev_alloc(ctx.h, down)
    C.memset(h_up, 1, up)
    for _ in range(3):
        L.mi_copy_h2d(ctx.h, d_up, h_up, up); L.mi_copy_d2h(ctx.h, h_dn, d_dn, down)
    ctx.sync()
    K = 20
    ctx.timer_start()
    for _ in range(K):
        L.mi_copy_h2d(ctx.h, d_up, h_up, up)
        L.mi_copy_d2h(ctx.h, h_dn, d_dn, down)
    ms_ = ctx.timer_stop() / K
    print(f"{n} streams: H2D {up/1e6:.1f} MB + D2H {down/1e6:.1f} MB per tick: {ms_:.3f} ms  ({(up+down)/ms_/1e6:.1f} GB/s)")
    L.mi_host_free(ctx.h, h_up); L.mi_host_free(ctx.h, h_dn); L.mi_dev_free(ctx.h, d_up); L.mi_dev_free(ctx.h, d_dn)
