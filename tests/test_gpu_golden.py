"""The frozen fixtures of tests/golden/*.npz replayed through the HIP kernels (C ABI): inputs and expected outputs are
data committed to the repository (generated once by tests/golden/make_golden.py from the oracle), so a joint drift of
oracle and kernel -- both changed the same wrong way -- shows up here even though the live-oracle tests stay green.
One test per filter; the bars are the filters' own (bit-exact for integer work, <= 1 LSB / 1e-4 RMS for the float ones)."""
import os

import numpy as np
import pytest

import mediastreamer2_amd as ms

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def test_golden_mixer(ctx):
    d = load("mixer")
    x, has, act, oen, gain = d["x"], d["has"], d["act"], d["oen"], d["gain"]
    nconf, mm, ns = x.shape
    mx = ms.MixerBatch(ctx, nconf, mm, ns)
    flags = (ms.MI_MIX_LINKED | (act * ms.MI_MIX_ACTIVE) | (oen * ms.MI_MIX_OUTPUT)).astype(np.uint8)
    mx.set_controls(flags, gain)
    got = mx.process(x, has_data=has, conf_mode=1, out=np.zeros_like(x))
    want = d["out"]
    for c in range(nconf):
        for m in range(mm):
            if oen[c, m]:  # rows of disabled outputs are not written (audiomixer.c:118-127)
                np.testing.assert_array_equal(got[c, m], want[c, m])
    flat = mx.process(x, has_data=has, conf_mode=0)
    np.testing.assert_array_equal(flat, d["flat"])
    mx.close()


def test_golden_volume(ctx):
    d = load("volume")
    x = d["x"]
    vb = ms.VolumeBatch(ctx, 1, 16000)
    p = vb.default_params()
    p.agc_enabled = 1
    p.noise_gate_enabled = 1
    vb.set_params([p])
    st = vb.get_state()[0]
    st.gain = st.target_gain = p.ng_floorgain
    vb.set_state([st])
    out = []
    for t in range(40):
        blk = x[None, t * 160:(t + 1) * 160].copy()
        out.append(vb.process(blk)[0])
        s = vb.get_state()[0]
        assert np.float32(s.energy) == d["energy"][t] and np.float32(s.gain) == d["gain"][t], f"tick {t}"
    np.testing.assert_array_equal(np.concatenate(out), d["out"])
    vb.close()


@pytest.mark.parametrize("a,b,n", [(16000, 48000, 160), (48000, 16000, 480), (44100, 48000, 441)])
def test_golden_resampler(ctx, a, b, n):
    d = load(f"resample_{a}_{b}")
    rs = ms.ResamplerBatch(ctx, 1, a, b)
    tab = rs.table()
    if rs.info()["direct"]:
        np.testing.assert_array_equal(np.asarray(tab, np.float32).ravel()[:len(d["table"])].view(np.uint32), d["table"].view(np.uint32))
    ys = []
    for i in range(10):
        out, olen = rs.process(d["x"][None, i * n:(i + 1) * n].copy())
        ys.append(out[0, :int(olen[0])])
    y = np.concatenate(ys)
    assert len(y) == len(d["y"])
    assert np.abs(y.astype(int) - d["y"].astype(int)).max() <= 1  # FMA vs separate multiply-add
    dd = (y.astype(np.float64) - d["y"]) / 32768.0
    assert np.sqrt(np.mean(dd * dd)) <= 1e-4
    rs.close()


def test_golden_equalizer(ctx):
    d = load("equalizer")
    eq = ms.EqualizerBatch(ctx, 1, 16000)
    eq.set_gain(0, 1000, 2.0, 500)
    eq.set_gain(0, 300, 0.3, 100)
    np.testing.assert_array_equal(np.asarray(eq.taps(0), np.float32).view(np.uint32), d["taps"].view(np.uint32))
    y = np.concatenate([eq.process(d["x"][None, i * 160:(i + 1) * 160].copy())[0] for i in range(6)])
    np.testing.assert_array_equal(y, d["y"])
    eq.close()


def test_golden_scaler(ctx):
    d = load("scaler")
    rgb = ms.ScalerBatch(ctx, 64, 48, 40, 30, ms.MI_PIX_RGB24).process(d["src"][None, :])
    np.testing.assert_array_equal(np.asarray(rgb).ravel(), d["rgb"].ravel())
    i420 = ms.ScalerBatch(ctx, 64, 48, 40, 30, ms.MI_PIX_I420).process(d["src"][None, :])
    np.testing.assert_array_equal(np.asarray(i420).ravel(), d["i420"].ravel())


def test_golden_echo_canceller(ctx):
    d = load("aec")
    F, flen, rate = 128, 2048, 16000
    mic, far = d["mic"].reshape(-1, F), d["far"].reshape(-1, F)
    for post, key in ((0, "out"), (ms.MI_AEC_POSTFILTER, "post")):
        aec = ms.AecBatch(ctx, 1, rate, frame_size=F, filter_length=flen)
        got = np.concatenate([aec.process(mic[k:k + 1], far[k:k + 1], flags=post)[0] for k in range(len(mic))])
        if not post:  # the canceller is bit-exact until the filter adapts (30 frames: it has not)
            np.testing.assert_array_equal(got, d[key])
            W = aec.get(0, "W", 16 * 256)
            np.testing.assert_array_equal(W.view(np.uint32), d["W"].view(np.uint32))
        else:
            dd = (got.astype(np.float64) - d[key]) / 32768.0
            assert np.sqrt(np.mean(dd * dd)) <= 1e-4 and np.abs(got.astype(int) - d[key]).max() <= 1
        aec.close()
