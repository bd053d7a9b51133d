"""Dev probe: does the canceller's launch time depend on HOW the legs' re-framing phases are arranged over the slots?
Workgroup b runs on XCD b % 8, so with the legs served in slot order a regular arrangement (phase = s % 8: every tick's light
legs on one XCD) used to cost as much as if every leg were heavy.  The FIFO entry now serves the legs from per-class lists
sorted by the frames they have (aec_tick.hpp: TickOrder), so every arrangement should cost the same as the product's hashed one.
  python scripts/aec_mix_probe.py 65536"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = ms.Context(0)
ids = np.arange(n)
hashed = np.array([ctx.L.mi_fifo_phase_of(int(s), 8) for s in range(min(n, 1 << 16))])
pats = {
    "all legs in one phase (no stagger)": np.zeros(n, int),
    "phase = s % 8": ids % 8,
    "phase = (s + s / 8) % 8": (ids + ids // 8) % 8,
    "phase = s * 8 / n (eight blocks of slots)": ids * 8 // n,
    "phase = hash(s) (the product's)": np.resize(hashed, n),
}
mic16, ref48 = bench.echo_scene(0)
z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
for name, phase in pats.items():
    rs = ms.ResamplerBatch(ctx, n, 16000, 48000)
    aec = ms.AecBatch(ctx, n, 48000, frame_size=256, filter_length=128 * 48)
    fm, fr = (ms.FifoBatch(ctx, n, 1024) for _ in range(2))
    fo = ms.FifoBatch(ctx, n, 4096)
    lead = torch.from_numpy((32 * phase).astype(np.int32)).cuda()
    zeros = z(n, 224)
    torch.cuda.synchronize()
    fm.push(zeros, count=lead)
    fr.push(zeros, count=lead)
    reps = -(-n // bench.SCENE_BASE)
    mic = [torch.from_numpy(np.ascontiguousarray(mic16[:, r * 160:(r + 1) * 160])).cuda().repeat(reps, 1)[:n].contiguous() for r in range(8)]
    ref = [torch.from_numpy(np.ascontiguousarray(ref48[:, r * 480:(r + 1) * 480])).cuda().repeat(reps, 1)[:n].contiguous() for r in range(8)]
    sink = z(n, 480)
    torch.cuda.synchronize()
    ts = []
    for t in range(40):
        ctx.timer_start()
        aec.process_fifos_resampled(rs, mic[t % 8], fm, fr, ref[t % 8], fo, max_frames=2)
        dt = ctx.timer_stop()
        fo.pop(480, sink, zero_fill=True)
        if t >= 16:
            ts.append(dt)
    ctx.sync()
    v = np.array(ts)
    print(f"{name:44s} launch mean {v.mean():7.3f} ms  min {v.min():7.3f}  max {v.max():7.3f}   (24 ticks, {n} legs)", flush=True)
    for o in (rs, aec, fm, fr, fo):
        o.close()
    del mic, ref
    torch.cuda.empty_cache()
