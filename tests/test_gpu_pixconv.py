"""MSPixConv conversions (packed -> I420) through the C ABI vs the oracle: bit-exact (integer work)."""
import numpy as np
import pytest

import mediastreamer2_amd as ms

pytestmark = pytest.mark.gpu

FMTS = [ms.MI_PIX_YUY2, ms.MI_PIX_UYVY, ms.MI_PIX_BGR24, ms.MI_PIX_RGB24_RAW, ms.MI_PIX_BGRA32]
BPP = {2: 2, 3: 2, 4: 3, 5: 3, 6: 4}


def frames(fmt, w, h, n, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (n, w * h * BPP[fmt]), dtype=np.uint8)
    # a smooth frame and the extremes as well
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = ((xx * 255 // max(1, w - 1)) ^ (yy * 3)).astype(np.uint8)
    f[0] = np.repeat(ramp.ravel(), BPP[fmt])
    if n > 1:
        f[1] = 255
    if n > 2:
        f[2] = 0
    return f


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("w,h", [(352, 288), (640, 480), (176, 144), (50, 37), (18, 2), (1280, 721)])
def test_pixconv_bit_exact(ctx, oracle, fmt, w, h):
    pc = ms.PixConvBatch(ctx, w, h, fmt)
    src = frames(fmt, w, h, 4, fmt * 1000 + w)
    got = pc.process(src)
    for i in range(src.shape[0]):
        want = oracle.pixconv_to_i420(fmt, src[i], w, h)
        hp = h + (h & 1)
        # the pad row of an odd-height luma plane is never written by either side
        np.testing.assert_array_equal(got[i][: w * h], want[: w * h])
        np.testing.assert_array_equal(got[i][w * hp:], want[w * hp:])


def test_pixconv_flip_is_negative_stride(ctx, oracle):
    """pixconv.c:78-81: MS_RGB24_REV walks the bitmap bottom-up."""
    w, h = 352, 288
    pc = ms.PixConvBatch(ctx, w, h, ms.MI_PIX_RGB24_RAW, flip=True)
    src = frames(ms.MI_PIX_RGB24_RAW, w, h, 2, 7)
    got = pc.process(src)
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.pixconv_to_i420(ms.MI_PIX_RGB24_RAW, src[i], w, h, flip=True))
        flipped = src[i].reshape(h, w * 3)[::-1].ravel()
        np.testing.assert_array_equal(got[i], oracle.pixconv_to_i420(ms.MI_PIX_RGB24_RAW, flipped, w, h))


def test_pixconv_device_resident_batch(ctx, oracle):
    import torch
    w, h, n = 1280, 720, 16
    pc = ms.PixConvBatch(ctx, w, h, ms.MI_PIX_YUY2)
    src = frames(ms.MI_PIX_YUY2, w, h, n, 3)
    out = pc.process(torch.from_numpy(src).cuda())
    ctx.sync()
    got = out.cpu().numpy()
    for i in (0, 5, n - 1):
        np.testing.assert_array_equal(got[i], oracle.pixconv_to_i420(ms.MI_PIX_YUY2, src[i], w, h))


def test_pixconv_rejects_what_the_reference_cannot_convert(ctx):
    with pytest.raises(Exception):
        ms.PixConvBatch(ctx, 352, 288, ms.MI_PIX_I420)   # same format is a pass-through in the filter
    with pytest.raises(Exception):
        ms.PixConvBatch(ctx, 351, 288, ms.MI_PIX_YUY2)   # odd width


def test_scaler_planes_host_matches_packed(ctx, oracle):
    """MSScalerDesc.context_process shape: strided planes in, strided planes out."""
    sw, sh, dw, dh = 640, 480, 352, 288
    rng = np.random.default_rng(5)
    packed = rng.integers(0, 256, oracle.i420_size(sw, sh), dtype=np.uint8)
    want = oracle.i420_scale(packed, sw, sh, dw, dh)
    ys, cs = sw + 32, sw // 2 + 16  # padded strides
    Y = np.zeros((sh, ys), np.uint8); U = np.zeros((sh // 2, cs), np.uint8); V = np.zeros((sh // 2, cs), np.uint8)
    Y[:, :sw] = packed[: sw * sh].reshape(sh, sw)
    U[:, : sw // 2] = packed[sw * sh: sw * sh + sw * sh // 4].reshape(sh // 2, sw // 2)
    V[:, : sw // 2] = packed[sw * sh + sw * sh // 4:].reshape(sh // 2, sw // 2)
    dys, dcs = dw + 8, dw // 2 + 8
    oY = np.zeros((dh, dys), np.uint8); oU = np.zeros((dh // 2, dcs), np.uint8); oV = np.zeros((dh // 2, dcs), np.uint8)
    sc = ms.ScalerBatch(ctx, sw, sh, dw, dh, ms.MI_PIX_I420)
    sc.process_planes([Y, U, V], [ys, cs, cs], [oY, oU, oV], [dys, dcs, dcs])
    np.testing.assert_array_equal(oY[:, :dw].ravel(), want[: dw * dh])
    np.testing.assert_array_equal(oU[:, : dw // 2].ravel(), want[dw * dh: dw * dh + dw * dh // 4])
    np.testing.assert_array_equal(oV[:, : dw // 2].ravel(), want[dw * dh + dw * dh // 4:])
    assert not oY[:, dw:].any() and not oU[:, dw // 2:].any()
