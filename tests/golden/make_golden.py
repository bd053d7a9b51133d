#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference has NO numeric golden vectors for this path and cannot
be built here, so these fixtures freeze (a) analytic known answers and (b) the CPU
oracle's outputs on seeded inputs, as a regression net for the oracle itself and as the
vectors the GPU tests replay.  They are data only (inputs + expected outputs); run
`python tests/golden/make_golden.py` from the repo root to regenerate.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pcm(seed, n, sigma=3000.0, rate=48000):
    rng = np.random.default_rng(0x5EED + seed)
    t = np.arange(n) / rate
    return np.clip(np.round(rng.normal(0, sigma, n) + 3276.7 * np.sin(2 * np.pi * 1000 * t)), -32767, 32767).astype(np.int16)


def main():
    oracle.build()
    # --- mixer: 3 conferences x 8 members x 160 samples, with every control exercised
    rng = np.random.default_rng(42)
    x = np.stack([[pcm(c * 8 + m, 160, 9000.0) for m in range(8)] for c in range(3)])
    x[0, 0, :6] = [-32768, 32767, -32768, 32767, 1, -1]
    has = (rng.random((3, 8)) > 0.15).astype(np.uint8)
    act = (rng.random((3, 8)) > 0.2).astype(np.uint8)
    oen = (rng.random((3, 8)) > 0.2).astype(np.uint8)
    gain = np.where(rng.random((3, 8)) < 0.4, rng.uniform(0, 2.5, (3, 8)), 1.0).astype(np.float32)
    outs, sums, flat = [], [], []
    for c in range(3):
        o, s = oracle.mixer_tick(x[c], has[c], gain[c], act[c], oen[c], 1)
        outs.append(o)
        sums.append(s)
        flat.append(oracle.mixer_tick(x[c], has[c], gain[c], act[c], oen[c], 0)[0])
    np.savez_compressed(os.path.join(HERE, "mixer.npz"), x=x, has=has, act=act, oen=oen, gain=gain,
                        out=np.stack(outs), sum=np.stack(sums), flat=np.stack(flat))

    # --- volume: AGC + noise gate stream, 40 ticks of 160 samples @16k
    v = oracle.Volume(16000)
    v.v.agc_enabled = 1
    v.v.noise_gate_enabled = 1
    v.v.gain = v.v.target_gain = v.v.ng_floorgain
    sig = pcm(7, 160 * 40, 2500.0, 16000)
    env = (np.arange(len(sig)) // 1600) % 2
    sig = (sig.astype(np.int32) * (1 + 9 * env) // 4).clip(-32767, 32767).astype(np.int16)
    out, energy, gains = [], [], []
    for t in range(40):
        out.append(v.chunk(sig[t * 160:(t + 1) * 160]))
        energy.append(v.v.energy)
        gains.append(v.v.gain)
    np.savez_compressed(os.path.join(HERE, "volume.npz"), x=sig, out=np.concatenate(out),
                        energy=np.array(energy, np.float32), gain=np.array(gains, np.float32))

    # --- resampler: 16k->48k and 48k->16k, 10 blocks
    for a, b, n in ((16000, 48000, 160), (48000, 16000, 480), (44100, 48000, 441)):
        r = oracle.Resampler(a, b)
        xin = pcm(a // 1000, n * 10, 3000.0, a)
        y = np.concatenate([r.process(xin[i * n:(i + 1) * n]) for i in range(10)])
        np.savez_compressed(os.path.join(HERE, f"resample_{a}_{b}.npz"), x=xin, y=y, table=r.table())

    # --- equalizer taps + output
    e = oracle.Equalizer(16000)
    e.set_gain(1000, 2.0, 500)
    e.set_gain(300, 0.3, 100)
    xin = pcm(3, 160 * 6, 2500.0, 16000)
    taps = e.taps()
    y = np.concatenate([e.run(xin[i * 160:(i + 1) * 160]) for i in range(6)])
    np.savez_compressed(os.path.join(HERE, "equalizer.npz"), x=xin, y=y, taps=taps, spectrum=e.spectrum())

    # --- scaler: 64x48 -> 40x30 RGB and I420
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, oracle.i420_size(64, 48), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "scaler.npz"), src=src,
                        rgb=oracle.i420_scale_to_rgb24(src, 64, 48, 40, 30), i420=oracle.i420_scale(src, 64, 48, 40, 30))

    # --- AEC: 30 frames at 16 kHz, canceller + post-filter
    rate, F, flen = 16000, 128, 2048
    rng = np.random.default_rng(11)
    far = rng.normal(0, 3000, F * 30)
    ir = rng.normal(0, 1, 64) * np.exp(-np.arange(64) / 12.0)
    mic = 0.5 * np.convolve(np.concatenate([np.zeros(320), far]), ir)[:len(far)] + rng.normal(0, 100, len(far))
    far16 = np.clip(np.round(far), -32767, 32767).astype(np.int16)
    mic16 = np.clip(np.round(mic), -32767, 32767).astype(np.int16)
    ec = oracle.Echo(F, flen, rate)
    pp = oracle.Preproc(F, rate, ec)
    o1, o2 = [], []
    for f in range(30):
        sl = slice(f * F, (f + 1) * F)
        o = ec.cancel(mic16[sl], far16[sl])
        o1.append(o)
        o2.append(pp.run(o))
    np.savez_compressed(os.path.join(HERE, "aec.npz"), mic=mic16, far=far16, out=np.concatenate(o1),
                        post=np.concatenate(o2), W=ec.get("W", 16 * 256))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
