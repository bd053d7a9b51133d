"""Dev probe (needs the library built with -DAEC_PROF_TIMING, see scripts/aec_timing_probe.sh): where a wave of the
canceller's tick kernel spends its time.  One two-frame launch at N legs; per phase the median over the legs of the
constant-clock stamps' differences (wall_clock64: 100 MHz)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = ms.Context(0)
leg = bench.make_aec_leg(ms, torch, ctx, n)
aec, mics, refs, out, two, one = leg.keep
L = ctx.L
L.mi_aec_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
assert L.mi_aec_debug_profile(aec.h, None) == 0  # arms the collection
for i in range(6):
    aec.process_frames(mics[i % 4], refs[i % 4], out, two, max_frames=2)
ctx.sync()
ctx.timer_start()
aec.process_frames(mics[2], refs[2], out, two, max_frames=2)
ms_launch = ctx.timer_stop()
buf = np.zeros((n, 16), np.uint64)
assert L.mi_aec_debug_profile(aec.h, buf.ctypes.data) == 0
t = buf.astype(np.int64)
tick_ns = 10.0  # wall_clock64(): the 100 MHz constant counter (s_memrealtime)
if os.environ.get("AEC_PROBE_TAIL"):
    names = ["state loads issued -> tail frame start (0-1; last frame: previous frame's end)", "residual-echo transform (1-2)", "echo estimate, windowing (2-3)",
             "band sum 1 (3-4)", "analysis transform (4-5)", "power spectrum + band sum 2 (5-6)", "noise update (6-7)", "band sum 3 (7-8)",
             "SNRs, zeta (8-9)", "band gains (9-10)", "per-bin gains (10-11)", "synthesis transform, output (11-12)", "state stores (12-13)"]
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11), (11, 12), (12, 13)]
    print("stamps inside the post-filter tail, LAST frame of the tick")
    for nm, (a_, b_) in zip(names, pairs):
        print(f"  {nm:80s} {np.median(t[:, b_] - t[:, a_]) * tick_ns / 1e3:7.2f} us")
    sys.exit(0)
names = ["fifo/prologue -> state+far-end ready (0-1)", "f1: notch, ring store, prop step (2-3)", "f1: streaming pass (3-4)",
         "f1: responses, two-path control (4-5)", "f1: output, spectra, adaptation (5-6)", "f2: notch, prop step (7-8)",
         "f2: streaming pass (8-9)", "f2: responses, control (9-10)", "f2: output, spectra, adaptation (10-11)",
         "post-filter tail + state stores (12-13)"]
pairs = [(0, 1), (2, 3), (3, 4), (4, 5), (5, 6), (7, 8), (8, 9), (9, 10), (10, 11), (12, 13)]
print(f"{n} legs, two frames each: launch {ms_launch:.3f} ms; counter tick = {tick_ns:.2f} ns; a wave lives "
      f"{np.median(t[:, 13] - t[:, 0]) * tick_ns / 1e3:.1f} us (median)")
tot = 0.0
for nm, (a, b) in zip(names, pairs):
    d = np.median(t[:, b] - t[:, a]) * tick_ns / 1e3
    tot += d
    print(f"  {nm:48s} {d:7.2f} us")
print(f"  {'sum of the phases':48s} {tot:7.2f} us")
