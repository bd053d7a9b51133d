"""The W-in-LDS experiment's parity probe (round 5): the chained path (mi_session: resampler folded into the canceller's launch,
FIFOs, volume + mix) over legs with spread re-framing phases on an echo scene, long enough for the cancellers to adapt and for
foreground updates / background resets to fall on both frames of a tick.  Prints one line: sha256 of every tick's mixes + the
cancellers' event counters.  Run once with MSMI355X_AEC_W_IN_LDS=1 and once without: the lines must be the same."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import mediastreamer2_amd as ms  # noqa: E402

n, nt = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = ms.Context(0)
se = ms.Session(ctx, n, members=32, in_rate=16000, rate=48000, tail_ms=128, agc=True, use_graphs=False, stagger=True)
rng = np.random.default_rng(77)
room = rng.normal(0, 1, (n, 64)) * np.exp(-np.arange(64) / 12.0)
room *= 0.5 / np.abs(room).sum(axis=1, keepdims=True)
h = hashlib.sha256()
hist = np.zeros((n, 480 * 3))
for t in range(nt):
    far = (rng.normal(0, 3000, (n, 480)) + 2000 * np.sin(2 * np.pi * 700 * (np.arange(480) + 480 * t) / 48000)).round().clip(-32767, 32767)
    hist = np.concatenate([hist[:, 480:], far], axis=1)
    late = hist[:, 480:960]                                        # 10 ms late
    echo = np.stack([np.convolve(late[s], room[s])[:480] for s in range(n)])
    loud = 6000.0 if (t // 40) % 3 == 1 else 150.0                 # double talk comes and goes: foreground updates and background resets
    mic48 = echo + rng.normal(0, loud, (n, 480))
    mic = mic48.reshape(n, 160, 3).mean(axis=2).round().clip(-32767, 32767).astype(np.int16)
    hm, hr = se.acquire()
    hm[:] = mic
    hr[:] = far.astype(np.int16)
    se.submit()
    h.update(se.collect().tobytes())
while se.in_flight():
    h.update(se.collect().tobytes())
print("wlds-check", os.environ.get("MSMI355X_AEC_W_IN_LDS", "0"), nt, h.hexdigest())
se.close()
ctx.close()
