"""Video side of the drop-in boundary on the GPU: MSSizeConv / MSPixConv facades registered under the
reference ids (src/videofilters/sizeconv.c, pixconv.c) and the MSScalerDesc the plugin installs with
ms_video_set_scaler_impl (src/voip/msvideo.c:719-721).  Graphs are source -> filter -> sink, ticked like
the reference's testers; frames carry the mblk video header of msvideo.c:79-83."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_plugin import Host, mid, MS_FILTER_BASE_ID, MS_FILTER_IS_HW_ACCELERATED

pytestmark = pytest.mark.gpu

MS_PIX_CONV_ID, MS_SIZE_CONV_ID = 29, 31
SET_VIDEO_SIZE = mid(MS_FILTER_BASE_ID, 100, 8)
GET_VIDEO_SIZE = mid(MS_FILTER_BASE_ID, 101, 8)
SET_PIX_FMT = mid(MS_FILTER_BASE_ID, 102, 4)
SET_FPS = mid(MS_FILTER_BASE_ID, 104, 4)
OUTPUT_FMT_CHANGED = mid(MS_FILTER_BASE_ID, 0, 0)
MS_YUV420P, MS_YUYV, MS_RGB24, MS_RGB24_REV, MS_UYVY, MS_YUY2, MS_RGBA32_REV = 1, 2, 3, 4, 6, 7, 11


class VSize(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int)]


@pytest.fixture(scope="module")
def vhost():
    h = Host()
    S, vp = h.S, C.c_void_p
    S.ms2shim_source_push_yuv.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32]
    S.ms2shim_source_push_ts.argtypes = [vp, vp, C.c_size_t, C.c_uint32]
    S.ms2shim_sink_last_ts.restype = C.c_uint32
    S.ms2shim_sink_last_ts.argtypes = [vp]
    S.ms2shim_watch.argtypes = [vp]
    S.ms2shim_notify_last.restype = C.c_uint
    S.ms_scaler_create_context.restype = vp
    S.ms_scaler_create_context.argtypes = [C.c_int] * 7
    S.ms_scaler_process.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(C.c_int)]
    S.ms_scaler_context_free.argtypes = [vp]
    S.ms_video_get_scaler_impl.restype = vp
    return h


def i420(oracle, w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    y = (16 + 200 * (xx + yy) / (w + h) + rng.normal(0, 10, (h, w))).clip(0, 255).astype(np.uint8)
    u = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
    return np.concatenate([y.ravel(), u.ravel(), v.ravel()])


def drain_bytes(host, sink):
    n = host.S.ms2shim_sink_size(sink)
    buf = np.zeros(n, np.uint8)
    if n:
        host.S.ms2shim_sink_read(sink, buf.ctypes.data, n)
    return buf


def test_video_descs_override_reference_ids(vhost):
    for fid, name in ((MS_SIZE_CONV_ID, b"MSSizeConv"), (MS_PIX_CONV_ID, b"MSPixConv")):
        f = vhost.create(fid)
        assert vhost.S.ms2shim_filter_name(f) == name
        assert vhost.S.ms2shim_filter_flags(f) & MS_FILTER_IS_HW_ACCELERATED
        vhost.S.ms_filter_destroy(f)
    assert vhost.S.ms_video_get_scaler_impl()  # installed by libmsmi355xfilters_init


def test_sizeconv_graph_batches_frames_of_many_filters(vhost, oracle):
    """Several MSSizeConv instances with one geometry share one launch per tick; output == oracle, +1 tick."""
    sw, sh, dw, dh = 640, 480, 352, 264  # 4:3 -> the aspect fix-up leaves the target alone
    n = 5
    chains = []
    for k in range(n):
        src, f, sink = vhost.source(), vhost.create(MS_SIZE_CONV_ID), vhost.sink()
        assert vhost.call(f, SET_VIDEO_SIZE, VSize(dw, dh)) == 0
        vhost.link(src, 0, f, 0)
        vhost.link(f, 0, sink, 0)
        assert vhost.S.ms_ticker_attach(vhost.ticker, src) == 0
        chains.append((src, f, sink))
    frames = [[i420(oracle, sw, sh, 100 * k + t) for t in range(3)] for k in range(n)]
    for k, (src, f, sink) in enumerate(chains):
        for t in range(3):
            vhost.S.ms2shim_source_push_yuv(src, frames[k][t].ctypes.data, sw, sh, 9000 * (t + 1) + k)
    vhost.step(5)
    for k, (src, f, sink) in enumerate(chains):
        assert vhost.S.ms2shim_sink_blocks(sink) == 3
        assert vhost.S.ms2shim_sink_last_ts(sink) == 9000 * 3 + k      # timestamps travel with the frames
        got = drain_bytes(vhost, sink)
        want = np.concatenate([oracle.i420_scale(frames[k][t], sw, sh, dw, dh) for t in range(3)])
        np.testing.assert_array_equal(got, want)
        got_sz = VSize()
        assert vhost.call(f, GET_VIDEO_SIZE, got_sz) == 0 and (got_sz.width, got_sz.height) == (dw, dh)
        vhost.S.ms_ticker_detach(vhost.ticker, src)


def test_sizeconv_passthrough_aspect_fixup_and_fps(vhost, oracle):
    src, f, sink = vhost.source(), vhost.create(MS_SIZE_CONV_ID), vhost.sink()
    vhost.link(src, 0, f, 0)
    vhost.link(f, 0, sink, 0)
    vhost.S.ms2shim_watch(f)
    vhost.S.ms_ticker_attach(vhost.ticker, src)
    # (1) same size: forwarded untouched, same tick (sizeconv.c:135-136)
    a = i420(oracle, 352, 288, 1)
    vhost.S.ms2shim_source_push_yuv(src, a.ctypes.data, 352, 288, 1)
    vhost.step(1)
    np.testing.assert_array_equal(drain_bytes(vhost, sink), a)
    # (2) 16:9 input into the 4:3 CIF target: the filter changes its own target to keep the aspect, notifies
    #     MS_FILTER_OUTPUT_FMT_CHANGED and emits nothing until the application re-sets the size (:147-157,:175)
    before = vhost.S.ms2shim_notify_count()
    b = i420(oracle, 1280, 720, 2)
    vhost.S.ms2shim_source_push_yuv(src, b.ctypes.data, 1280, 720, 2)
    vhost.step(3)
    assert vhost.S.ms2shim_notify_count() == before + 1 and vhost.S.ms2shim_notify_last() == OUTPUT_FMT_CHANGED
    sz = VSize()
    vhost.call(f, GET_VIDEO_SIZE, sz)
    assert (sz.width, sz.height) == (352, 720 * 352 // 1280)
    assert vhost.S.ms2shim_sink_size(sink) == 0
    # (3) portrait input: orientation swap first (:141-146)
    vhost.call(f, SET_VIDEO_SIZE, VSize(352, 288))
    c = i420(oracle, 480, 640, 3)
    vhost.S.ms2shim_source_push_yuv(src, c.ctypes.data, 480, 640, 3)
    vhost.step(2)
    vhost.call(f, GET_VIDEO_SIZE, sz)
    assert sz.height > sz.width
    vhost.S.ms_ticker_detach(vhost.ticker, src)
    vhost.S.ms_filter_destroy(f)


def test_sizeconv_fps_limit_keeps_newest(vhost, oracle):
    """sizeconv.c:113-132: with fps set, at most one frame per frame period, the most recent one."""
    sw, sh, dw, dh = 640, 480, 320, 240
    src, f, sink = vhost.source(), vhost.create(MS_SIZE_CONV_ID), vhost.sink()
    vhost.call(f, SET_VIDEO_SIZE, VSize(dw, dh))
    fps = C.c_float(10.0)  # one frame per 100 ms = 10 ticks
    assert vhost.S.ms_filter_call_method(f, SET_FPS, C.byref(fps)) == 0
    vhost.link(src, 0, f, 0)
    vhost.link(f, 0, sink, 0)
    vhost.S.ms_ticker_attach(vhost.ticker, src)
    fr = [i420(oracle, sw, sh, 50 + t) for t in range(30)]
    for t in range(30):  # a frame every tick
        vhost.S.ms2shim_source_push_yuv(src, fr[t].ctypes.data, sw, sh, t)
    vhost.step(34)
    nb = vhost.S.ms2shim_sink_blocks(sink)
    assert 2 <= nb <= 4, nb   # ~3 frame periods elapsed
    got = drain_bytes(vhost, sink).reshape(nb, -1)
    # each emitted frame is the scaled version of SOME input frame, in increasing order
    idx = []
    for g in got:
        hits = [t for t in range(30) if np.array_equal(g, oracle.i420_scale(fr[t], sw, sh, dw, dh))]
        assert hits
        idx.append(hits[0])
    assert idx == sorted(idx) and len(set(idx)) == len(idx)
    vhost.S.ms_ticker_detach(vhost.ticker, src)


@pytest.mark.parametrize("msfmt,ofmt,bpp,flip", [(MS_YUY2, 2, 2, False), (MS_UYVY, 3, 2, False), (MS_RGB24, 4, 3, False),
                                                (MS_RGB24_REV, 5, 3, True), (MS_RGBA32_REV, 6, 4, False)])
def test_pixconv_graph(vhost, oracle, msfmt, ofmt, bpp, flip):
    w, h = 352, 288
    src, f, sink = vhost.source(), vhost.create(MS_PIX_CONV_ID), vhost.sink()
    assert vhost.call(f, SET_VIDEO_SIZE, VSize(w, h)) == 0
    assert vhost.call_int(f, SET_PIX_FMT, msfmt) == 0
    vhost.link(src, 0, f, 0)
    vhost.link(f, 0, sink, 0)
    vhost.S.ms_ticker_attach(vhost.ticker, src)
    rng = np.random.default_rng(msfmt)
    frames = [rng.integers(0, 256, w * h * bpp, dtype=np.uint8) for _ in range(3)]
    for t, fr in enumerate(frames):
        vhost.S.ms2shim_source_push_ts(src, fr.ctypes.data, fr.nbytes, 777 + t)
    vhost.step(5)
    assert vhost.S.ms2shim_sink_blocks(sink) == 3 and vhost.S.ms2shim_sink_last_ts(sink) == 779
    got = drain_bytes(vhost, sink)
    want = np.concatenate([oracle.pixconv_to_i420(ofmt, fr, w, h, flip=flip) for fr in frames])
    np.testing.assert_array_equal(got, want)
    vhost.S.ms_ticker_detach(vhost.ticker, src)


def test_pixconv_same_format_is_passthrough(vhost, oracle):
    src, f, sink = vhost.source(), vhost.create(MS_PIX_CONV_ID), vhost.sink()  # default in_fmt == out_fmt == YUV420P
    vhost.link(src, 0, f, 0)
    vhost.link(f, 0, sink, 0)
    vhost.S.ms_ticker_attach(vhost.ticker, src)
    a = i420(oracle, 352, 288, 9)
    vhost.S.ms2shim_source_push_yuv(src, a.ctypes.data, 352, 288, 5)
    vhost.step(1)
    np.testing.assert_array_equal(drain_bytes(vhost, sink), a)
    vhost.S.ms_ticker_detach(vhost.ticker, src)


def test_scaler_desc_through_reference_entry_points(vhost, oracle):
    """ms_scaler_create_context / ms_scaler_process / ms_scaler_context_free (msvideo.c:702-717) reach the GPU."""
    S, vp = vhost.S, C.c_void_p
    sw, sh, dw, dh = 640, 480, 320, 240
    frame = i420(oracle, sw, sh, 4)
    # I420 -> I420
    ctx = S.ms_scaler_create_context(sw, sh, MS_YUV420P, dw, dh, MS_YUV420P, 2)
    assert ctx
    out = np.zeros(oracle.i420_size(dw, dh), np.uint8)
    sp = (vp * 3)(frame.ctypes.data, frame.ctypes.data + sw * sh, frame.ctypes.data + sw * sh * 5 // 4)
    ss = (C.c_int * 3)(sw, sw // 2, sw // 2)
    dp = (vp * 3)(out.ctypes.data, out.ctypes.data + dw * dh, out.ctypes.data + dw * dh * 5 // 4)
    ds = (C.c_int * 3)(dw, dw // 2, dw // 2)
    assert S.ms_scaler_process(ctx, sp, ss, dp, ds) == 0
    np.testing.assert_array_equal(out, oracle.i420_scale(frame, sw, sh, dw, dh))
    S.ms_scaler_context_free(ctx)
    # I420 -> RGB24 (what the display filters ask for)
    ctx = S.ms_scaler_create_context(sw, sh, MS_YUV420P, dw, dh, MS_RGB24, 2)
    rgb = np.zeros(dw * dh * 3, np.uint8)
    dp = (vp * 3)(rgb.ctypes.data, None, None)
    ds = (C.c_int * 3)(dw * 3, 0, 0)
    assert S.ms_scaler_process(ctx, sp, ss, dp, ds) == 0
    np.testing.assert_array_equal(rgb, oracle.i420_scale_to_rgb24(frame, sw, sh, dw, dh).ravel())
    S.ms_scaler_context_free(ctx)
    # packed source with a negative stride (bottom-up bitmap), as pixconv.c:78-81 passes it
    w, h = 176, 144
    bmp = np.random.default_rng(8).integers(0, 256, w * h * 3, dtype=np.uint8)
    ctx = S.ms_scaler_create_context(w, h, MS_RGB24_REV, w, h, MS_YUV420P, 2)
    out = np.zeros(oracle.i420_size(w, h), np.uint8)
    sp = (vp * 3)(bmp.ctypes.data + w * 3 * (h - 1), None, None)
    ss = (C.c_int * 3)(-w * 3, 0, 0)
    dp = (vp * 3)(out.ctypes.data, out.ctypes.data + w * h, out.ctypes.data + w * h * 5 // 4)
    ds = (C.c_int * 3)(w, w // 2, w // 2)
    assert S.ms_scaler_process(ctx, sp, ss, dp, ds) == 0
    np.testing.assert_array_equal(out, oracle.pixconv_to_i420(5, bmp, w, h, flip=True))
    S.ms_scaler_context_free(ctx)
    # a format the reference's scaler cannot take either
    assert not S.ms_scaler_create_context(w, h, 9, w, h, MS_YUV420P, 2)  # MS_RGB565
