"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- parity unpinned, see oracle/ms2_oracle.h.  Import
this from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() and
from nowhere else; the product package never touches it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
        for f in os.listdir(_HERE)
        if f.endswith((".c", ".h"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=sys.stderr)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _declare(_LIB)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OrcVolume(C.Structure):
    _fields_ = [
        ("energy", C.c_float), ("level_pk", C.c_float), ("instant_energy", C.c_float),
        ("lt_speaker_en", C.c_float), ("gain", C.c_float), ("static_gain", C.c_float),
        ("dc_offset", C.c_int), ("vol_upramp", C.c_float), ("vol_fast_upramp", C.c_float),
        ("vol_downramp", C.c_float), ("ea_thres", C.c_float), ("ea_transmit_thres", C.c_float),
        ("force", C.c_float), ("target_gain", C.c_float), ("sustain_time", C.c_int),
        ("sustain_dur", C.c_int), ("sample_rate", C.c_int), ("nsamples", C.c_int),
        ("ng_cut_time", C.c_int), ("ng_noise_dur", C.c_int), ("ng_threshold", C.c_float),
        ("ng_floorgain", C.c_float), ("ng_gain", C.c_float), ("agc_enabled", C.c_int),
        ("noise_gate_enabled", C.c_int), ("remove_dc", C.c_int), ("fast_upramp", C.c_int),
        ("has_peer", C.c_int),
    ]


def _declare(L):
    i16p, u8p, f32p, i32p = (C.POINTER(t) for t in (C.c_int16, C.c_uint8, C.c_float, C.c_int32))
    L.orc_mixer_tick.argtypes = [i16p, u8p, f32p, u8p, u8p, C.c_int, C.c_int, C.c_int, i16p, i32p]
    L.orc_mixer_tick.restype = None
    for n in ("orc_volume_init",):
        getattr(L, n).argtypes = [C.POINTER(OrcVolume)]
    L.orc_volume_set_rate.argtypes = [C.POINTER(OrcVolume), C.c_int]
    L.orc_volume_set_gain.argtypes = [C.POINTER(OrcVolume), C.c_float]
    L.orc_volume_set_db_gain.argtypes = [C.POINTER(OrcVolume), C.c_float]
    L.orc_volume_enable_noise_gate.argtypes = [C.POINTER(OrcVolume), C.c_int]
    L.orc_volume_chunk.argtypes = [C.POINTER(OrcVolume), i16p, C.c_int, C.c_float]
    L.orc_resampler_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.orc_resampler_new.restype = C.c_void_p
    L.orc_resampler_free.argtypes = [C.c_void_p]
    L.orc_resampler_process.argtypes = [C.c_void_p, i16p, C.POINTER(C.c_uint32), i16p, C.POINTER(C.c_uint32)]
    for n in ("orc_resampler_filt_len", "orc_resampler_den_rate", "orc_resampler_num_rate",
              "orc_resampler_is_direct"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int
    L.orc_resampler_table.argtypes = [C.c_void_p, f32p, C.c_int]
    L.orc_resampler_table.restype = C.c_int
    L.orc_msresample_outcap.argtypes = [C.c_uint32] * 3
    L.orc_msresample_outcap.restype = C.c_uint32
    L.orc_ms_fft.argtypes = [C.c_int, f32p, f32p]
    L.orc_ms_ifft.argtypes = [C.c_int, f32p, f32p]
    L.orc_equalizer_new.argtypes = [C.c_int]
    L.orc_equalizer_new.restype = C.c_void_p
    L.orc_equalizer_free.argtypes = [C.c_void_p]
    L.orc_equalizer_set_rate.argtypes = [C.c_void_p, C.c_int]
    L.orc_equalizer_set_gain.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
    L.orc_equalizer_design.argtypes = [C.c_void_p]
    L.orc_equalizer_run.argtypes = [C.c_void_p, i16p, C.c_int]
    L.orc_fir_mem16.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p]
    L.orc_scale_plane_bilinear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
    L.orc_i420_scale.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    L.orc_i420_to_rgb24.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int]
    L.orc_i420_scale_to_rgb24.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    L.orc_pixconv_to_i420.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, u8p]
    if hasattr(L, "orc_echo_new"):
        L.orc_echo_new.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_echo_new.restype = C.c_void_p
        L.orc_echo_free.argtypes = [C.c_void_p]
        L.orc_echo_cancel.argtypes = [C.c_void_p, i16p, i16p, i16p]
        L.orc_echo_get.argtypes = [C.c_void_p, C.c_char_p, f32p, C.c_int]
        L.orc_echo_get.restype = C.c_int
        L.adjust_framesize_8000.argtypes = [C.c_int, C.c_int]
        L.adjust_framesize_8000.restype = C.c_int
    if hasattr(L, "orc_preproc_new"):
        L.orc_preproc_new.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_preproc_new.restype = C.c_void_p
        L.orc_preproc_free.argtypes = [C.c_void_p]
        L.orc_preproc_run.argtypes = [C.c_void_p, i16p]


# ------------------------------------------------------------------ helpers
def mixer_tick(inp, has_data=None, gain=None, active=None, out_enabled=None, conf_mode=1):
    """inp [members][nsamples] int16 -> (out, sum). conf_mode 0 -> out is [nsamples]."""
    inp = np.ascontiguousarray(inp, dtype=np.int16)
    m, n = inp.shape
    has_data = np.ones(m, np.uint8) if has_data is None else np.ascontiguousarray(has_data, np.uint8)
    gain = np.ones(m, np.float32) if gain is None else np.ascontiguousarray(gain, np.float32)
    active = np.ones(m, np.uint8) if active is None else np.ascontiguousarray(active, np.uint8)
    out_enabled = np.ones(m, np.uint8) if out_enabled is None else np.ascontiguousarray(out_enabled, np.uint8)
    out = np.zeros((m, n) if conf_mode else (n,), np.int16)
    s = np.zeros(n, np.int32)
    lib().orc_mixer_tick(_p(inp, C.c_int16), _p(has_data, C.c_uint8), _p(gain, C.c_float),
                         _p(active, C.c_uint8), _p(out_enabled, C.c_uint8), m, n, int(conf_mode),
                         _p(out, C.c_int16), _p(s, C.c_int32))
    return out, s


class Volume:
    def __init__(self, rate=48000):
        self.v = OrcVolume()
        lib().orc_volume_init(C.byref(self.v))
        lib().orc_volume_set_rate(C.byref(self.v), rate)

    def chunk(self, samples, peer_energy=0.0):
        s = np.array(samples, dtype=np.int16, copy=True)
        lib().orc_volume_chunk(C.byref(self.v), _p(s, C.c_int16), len(s), float(peer_energy))
        return s


class Resampler:
    def __init__(self, in_rate, out_rate, quality=3):
        self.h = lib().orc_resampler_new(in_rate, out_rate, quality)
        if not self.h:
            raise ValueError("unsupported resampler config")
        self.in_rate, self.out_rate = in_rate, out_rate

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_resampler_free(self.h)
            self.h = None

    @property
    def filt_len(self):
        return lib().orc_resampler_filt_len(self.h)

    @property
    def den_rate(self):
        return lib().orc_resampler_den_rate(self.h)

    @property
    def num_rate(self):
        return lib().orc_resampler_num_rate(self.h)

    @property
    def direct(self):
        return bool(lib().orc_resampler_is_direct(self.h))

    def table(self):
        n = lib().orc_resampler_table(self.h, None, 0)
        t = np.zeros(n, np.float32)
        lib().orc_resampler_table(self.h, _p(t, C.c_float), n)
        return t

    def process(self, x):
        """One input block as msresample.c:150-177 would hand it over."""
        x = np.ascontiguousarray(x, dtype=np.int16)
        cap = lib().orc_msresample_outcap(len(x), self.in_rate, self.out_rate)
        out = np.zeros(cap, np.int16)
        il, ol = C.c_uint32(len(x)), C.c_uint32(cap)
        lib().orc_resampler_process(self.h, _p(x, C.c_int16), C.byref(il), _p(out, C.c_int16), C.byref(ol))
        assert il.value == len(x), "resampler did not consume its input (msresample.c:163)"
        return out[: ol.value]


def ms_fft(x):
    x = np.ascontiguousarray(x, np.float32)
    o = np.zeros_like(x)
    lib().orc_ms_fft(len(x), _p(x, C.c_float), _p(o, C.c_float))
    return o


def ms_ifft(x):
    x = np.ascontiguousarray(x, np.float32)
    o = np.zeros_like(x)
    lib().orc_ms_ifft(len(x), _p(x, C.c_float), _p(o, C.c_float))
    return o


class _EqState(C.Structure):
    _fields_ = [("rate", C.c_int), ("nfft", C.c_int), ("fir_len", C.c_int),
                ("fft_cpx", C.POINTER(C.c_float)), ("fir", C.POINTER(C.c_float)),
                ("mem", C.POINTER(C.c_float)), ("needs_update", C.c_int), ("active", C.c_int)]


class Equalizer:
    def __init__(self, rate):
        self.h = lib().orc_equalizer_new(rate)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_equalizer_free(self.h)
            self.h = None

    def _st(self):
        return C.cast(self.h, C.POINTER(_EqState)).contents

    def set_gain(self, freq, gain, width):
        lib().orc_equalizer_set_gain(self.h, int(freq), float(gain), int(width))

    def set_rate(self, rate):
        """MS_FILTER_SET_SAMPLE_RATE (equalizer.c:57-79,305-309): a flat spectrum and a cleared FIR memory, at the same rate too"""
        lib().orc_equalizer_set_rate(self.h, int(rate))

    def taps(self):
        lib().orc_equalizer_design(self.h)
        st = self._st()
        return np.ctypeslib.as_array(st.fir, (st.fir_len,)).copy()

    def spectrum(self):
        st = self._st()
        return np.ctypeslib.as_array(st.fft_cpx, (st.nfft,)).copy()

    def run(self, samples):
        s = np.array(samples, dtype=np.int16, copy=True)
        lib().orc_equalizer_run(self.h, _p(s, C.c_int16), len(s))
        return s


def i420_size(w, h):
    h2 = h + (h & 1)
    return w * h2 + 2 * (w // 2) * (h2 // 2)


def i420_scale(src, sw, sh, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros(i420_size(dw, dh), np.uint8)
    lib().orc_i420_scale(_p(src, C.c_uint8), sw, sh, _p(dst, C.c_uint8), dw, dh)
    return dst


def i420_to_rgb24(src, w, h):
    src = np.ascontiguousarray(src, np.uint8)
    rgb = np.zeros((h, w, 3), np.uint8)
    lib().orc_i420_to_rgb24(_p(src, C.c_uint8), w, h, _p(rgb, C.c_uint8), w * 3)
    return rgb


def i420_scale_to_rgb24(src, sw, sh, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    rgb = np.zeros((dh, dw, 3), np.uint8)
    lib().orc_i420_scale_to_rgb24(_p(src, C.c_uint8), sw, sh, _p(rgb, C.c_uint8), dw, dh)
    return rgb


class Echo:
    def __init__(self, frame_size, filter_length, rate):
        self.h = lib().orc_echo_new(frame_size, filter_length, rate)
        self.frame = frame_size

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_echo_free(self.h)
            self.h = None

    def cancel(self, rec, play):
        rec = np.ascontiguousarray(rec, np.int16)
        play = np.ascontiguousarray(play, np.int16)
        out = np.zeros(self.frame, np.int16)
        lib().orc_echo_cancel(self.h, _p(rec, C.c_int16), _p(play, C.c_int16), _p(out, C.c_int16))
        return out

    def get(self, what, n):
        a = np.zeros(n, np.float32)
        got = lib().orc_echo_get(self.h, what.encode(), _p(a, C.c_float), n)
        return a[:got]


class Preproc:
    def __init__(self, frame_size, rate, echo=None):
        self.echo = echo
        self.h = lib().orc_preproc_new(frame_size, rate, echo.h if echo is not None else None)
        self.frame = frame_size

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_preproc_free(self.h)
            self.h = None

    def run(self, x):
        x = np.array(x, dtype=np.int16, copy=True)
        lib().orc_preproc_run(self.h, _p(x, C.c_int16))
        return x


PIX_YUY2, PIX_UYVY, PIX_BGR24, PIX_RGB24_RAW, PIX_BGRA32 = 2, 3, 4, 5, 6
PIX_BPP = {PIX_YUY2: 2, PIX_UYVY: 2, PIX_BGR24: 3, PIX_RGB24_RAW: 3, PIX_BGRA32: 4}


def pixconv_to_i420(fmt, src, w, h, flip=False):
    """src: packed frame, w*h*bpp bytes; flip = walk it bottom-up (negative stride, pixconv.c:78-81)."""
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros(i420_size(w, h), np.uint8)
    stride = w * PIX_BPP[fmt]
    base = src.ctypes.data + (stride * (h - 1) if flip else 0)
    rc = lib().orc_pixconv_to_i420(fmt, C.c_void_p(base), -stride if flip else stride, w, h, _p(dst, C.c_uint8))
    if rc != 0:
        raise ValueError("orc_pixconv_to_i420 rejected the arguments")
    return dst


# ---------------------------------------------------------------- codecs / adapters / flow control
LAW_PCMA, LAW_PCMU = 0, 1


def _declare_codecs(L):
    i16p, u8p = C.POINTER(C.c_int16), C.POINTER(C.c_uint8)
    L.orc_g711_encode.argtypes = [C.c_int, i16p, C.c_size_t, u8p]
    L.orc_g711_decode.argtypes = [C.c_int, u8p, C.c_size_t, i16p]
    L.orc_l16_swap.argtypes = [i16p, C.c_size_t, i16p]
    L.orc_chan_adapt.argtypes = [C.c_int, i16p, i16p, C.c_size_t, i16p]
    L.orc_flowctl_init.argtypes = [C.POINTER(OrcFlowCtl)]
    L.orc_flowctl_set_target.argtypes = [C.POINTER(OrcFlowCtl), C.c_uint32, C.c_uint32]
    L.orc_flowctl_process.argtypes = [C.POINTER(OrcFlowCtl), i16p, C.c_size_t]
    L.orc_flowctl_process.restype = C.c_size_t
    L._codecs_declared = True


def _clib():
    L = lib()
    if not getattr(L, "_codecs_declared", False):
        _declare_codecs(L)
    return L


def g711_encode(law, pcm):
    pcm = np.ascontiguousarray(pcm, np.int16)
    out = np.empty(pcm.shape, np.uint8)
    _clib().orc_g711_encode(law, _p(pcm, C.c_int16), pcm.size, _p(out, C.c_uint8))
    return out


def g711_decode(law, codes):
    codes = np.ascontiguousarray(codes, np.uint8)
    out = np.empty(codes.shape, np.int16)
    _clib().orc_g711_decode(law, _p(codes, C.c_uint8), codes.size, _p(out, C.c_int16))
    return out


def l16_swap(x):
    x = np.ascontiguousarray(x, np.int16)
    out = np.empty_like(x)
    _clib().orc_l16_swap(_p(x, C.c_int16), x.size, _p(out, C.c_int16))
    return out


def chan_adapt(mode, a, b=None):
    """mode 0 mono->stereo, 1 stereo->mono (left kept), 2 two monos -> interleaved stereo (b None = silence)."""
    a = np.ascontiguousarray(a, np.int16)
    n = a.size // 2 if mode == 1 else a.size
    out = np.empty(n if mode == 1 else 2 * n, np.int16)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.int16)
        bp = _p(b, C.c_int16)
    _clib().orc_chan_adapt(mode, _p(a, C.c_int16), bp, n, _p(out, C.c_int16))
    return out


class OrcFlowCtl(C.Structure):
    _fields_ = [("strategy", C.c_int), ("silent_threshold", C.c_float), ("target_samples", C.c_uint32),
                ("total_samples", C.c_uint32), ("current_pos", C.c_uint32), ("current_dropped", C.c_uint32)]


class FlowCtl:
    """MSAudioFlowController (flowcontrol.c:37-152): set_target() then process() block by block."""

    def __init__(self, strategy=1, silent_threshold=0.02):
        self.c = OrcFlowCtl()
        _clib().orc_flowctl_init(C.byref(self.c))
        self.c.strategy = strategy
        self.c.silent_threshold = silent_threshold

    def set_target(self, samples_to_drop, total_samples):
        _clib().orc_flowctl_set_target(C.byref(self.c), samples_to_drop, total_samples)

    def process(self, block):
        x = np.array(block, dtype=np.int16, copy=True)
        n = _clib().orc_flowctl_process(C.byref(self.c), _p(x, C.c_int16), x.size)
        return x[:n]


def g711_ref():
    """The reference's own g711.c compiled by oracle/build_ref.sh, or None when it was never built."""
    so = os.path.join(_HERE, "_ref", "libg711_ref.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    R.Snack_Lin2Alaw.argtypes = R.Snack_Lin2Mulaw.argtypes = [C.c_short]
    R.Snack_Lin2Alaw.restype = R.Snack_Lin2Mulaw.restype = C.c_ubyte
    R.Snack_Alaw2Lin.argtypes = R.Snack_Mulaw2Lin.argtypes = [C.c_ubyte]
    R.Snack_Alaw2Lin.restype = R.Snack_Mulaw2Lin.restype = C.c_short
    return R


# ---------------------------------------------------------------- MSGenericPLC
class OrcConcealer(C.Structure):
    _fields_ = [("sample_time", C.c_int64), ("plc_start_time", C.c_int64), ("total_number_for_plc", C.c_ulong),
                ("max_plc_time", C.c_uint32)]


def _declare_plc(L):
    i16p = C.POINTER(C.c_int16)
    L.orc_plc_new.restype = C.c_void_p
    L.orc_plc_new.argtypes = [C.c_int]
    L.orc_plc_free.argtypes = [C.c_void_p]
    L.orc_plc_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_plc_received.argtypes = [C.c_void_p, i16p, C.c_size_t, C.c_int]
    L.orc_plc_conceal.argtypes = [C.c_void_p, i16p, C.c_uint16]
    L.orc_plc_transition_mix.argtypes = [i16p, i16p, C.c_uint16]
    L.orc_concealer_init.argtypes = [C.POINTER(OrcConcealer), C.c_uint32]
    L.orc_concealer_inc_sample_time.argtypes = [C.POINTER(OrcConcealer), C.c_uint64, C.c_uint32, C.c_int]
    L.orc_concealer_inc_sample_time.restype = C.c_uint32
    L.orc_concealer_required.argtypes = [C.POINTER(OrcConcealer), C.c_uint64]
    L._plc_declared = True


class Plc:
    """plc_context_t (genericplc.c) for one stream: received() for a block that arrived, conceal() for a missing tick."""

    def __init__(self, rate):
        L = lib()
        if not getattr(L, "_plc_declared", False):
            _declare_plc(L)
        self.L, self.rate = L, rate
        self.h = L.orc_plc_new(rate)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_plc_free(self.h)
            self.h = None

    def info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_plc_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(nb=a.value, index=b.value, used=c.value)

    def received(self, block, cng_resume=False):
        x = np.array(block, dtype=np.int16, copy=True)
        self.L.orc_plc_received(self.h, _p(x, C.c_int16), x.size, int(cng_resume))
        return x

    def conceal(self, n):
        x = np.zeros(n, np.int16)
        self.L.orc_plc_conceal(self.h, _p(x, C.c_int16), n)
        return x


class GenericPlcFilter:
    """generic_plc_process (msgenericplc.c:59-167, build without bcg729) for one mono stream: feed tick(now_ms,
    blocks) once per ticker interval, get the list of blocks the filter emits."""

    def __init__(self, rate, interval_ms=10):
        self.plc = Plc(rate)
        self.rate, self.interval = rate, interval_ms
        self.con = OrcConcealer()
        self.plc.L.orc_concealer_init(C.byref(self.con), 0xFFFFFFFF)  # MAX_PLC_COUNT = UINT32_MAX (:43)
        self.cng_set = self.cng_running = False

    def set_cn(self):  # MS_GENERIC_PLC_SET_CN (:196-201)
        self.cng_set = True

    def tick(self, now_ms, blocks):
        L, out = self.plc.L, []
        for b in blocks:
            ms = (1000 * b.size * 2) // (self.rate * 2)  # :66 for one channel
            L.orc_concealer_inc_sample_time(C.byref(self.con), now_ms, ms, 1)
            out.append(self.plc.received(b, cng_resume=self.cng_running))
            if self.cng_running:
                self.cng_running = self.cng_set = False
        if L.orc_concealer_required(C.byref(self.con), now_ms):
            n = self.rate * self.interval // 1000
            if self.cng_set:
                self.cng_set, self.cng_running = False, True
                out.append(np.zeros(n, np.int16))
            elif self.cng_running:
                out.append(np.zeros(n, np.int16))
            else:
                out.append(self.plc.conceal(n))
            L.orc_concealer_inc_sample_time(C.byref(self.con), now_ms, self.interval, 0)
        return out


# ------------------------------------------------------------------ conference glue (oracle/conference.c)
class OrcExtremum(C.Structure):
    _fields_ = [("current_extremum", C.c_float), ("last_stable", C.c_float), ("extremum_time", C.c_uint64), ("period", C.c_int)]


class OrcConference(C.Structure):
    _fields_ = [("nmembers", C.c_int), ("active_speaker", C.c_int), ("plumbed", C.c_uint8 * 50), ("muted", C.c_uint8 * 50)]


def _declare_conference(L):
    if getattr(L, "_conference_declared", False):
        return L
    ep, cp = C.POINTER(OrcExtremum), C.POINTER(OrcConference)
    L.orc_extremum_init.argtypes = [ep, C.c_int]
    L.orc_extremum_reset.argtypes = [ep]
    L.orc_extremum_record_min.argtypes = [ep, C.c_uint64, C.c_float]
    L.orc_extremum_record_max.argtypes = [ep, C.c_uint64, C.c_float]
    L.orc_extremum_get_current.argtypes = [ep]
    L.orc_extremum_get_current.restype = C.c_float
    L.orc_volume_linear_to_dbm0.argtypes = [C.c_float]
    L.orc_volume_linear_to_dbm0.restype = C.c_float
    L.orc_conference_init.argtypes = [cp]
    L.orc_conference_add_member.argtypes = [cp, C.c_int]
    L.orc_conference_remove_member.argtypes = [cp, C.c_int]
    L.orc_conference_mute_member.argtypes = [cp, C.c_int, C.c_int]
    L.orc_conference_get_size.argtypes = [cp]
    L.orc_conference_participant_volume.argtypes = [cp, C.c_int, C.c_float]
    L.orc_conference_process_events.argtypes = [cp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L._conference_declared = True
    return L


def linear_to_dbm0(x):
    return _declare_conference(lib()).orc_volume_linear_to_dbm0(float(x))


class Extremum:
    """OrtpExtremum as MSVolume feeds it (msvolume.c:115-116,405-406)"""

    def __init__(self, period_ms):
        self.L = _declare_conference(lib())
        self.e = OrcExtremum()
        self.L.orc_extremum_init(C.byref(self.e), int(period_ms))

    def reset(self):
        self.L.orc_extremum_reset(C.byref(self.e))

    def record_max(self, now_ms, v):
        return self.L.orc_extremum_record_max(C.byref(self.e), int(now_ms), float(v))

    def record_min(self, now_ms, v):
        return self.L.orc_extremum_record_min(C.byref(self.e), int(now_ms), float(v))

    @property
    def current(self):
        return self.L.orc_extremum_get_current(C.byref(self.e))


class Conference:
    """MSAudioConference's bookkeeping in mixer mode (src/voip/audioconference.c); members are named by their mixer pin"""

    def __init__(self):
        self.L = _declare_conference(lib())
        self.c = OrcConference()
        self.L.orc_conference_init(C.byref(self.c))
        self.order = []   # the conference's member list, in joining order

    def add_member(self, muted=False):
        pin = self.L.orc_conference_add_member(C.byref(self.c), int(muted))
        if pin >= 0:
            self.order.append(pin)
        return pin

    def remove_member(self, pin):
        self.L.orc_conference_remove_member(C.byref(self.c), int(pin))
        self.order.remove(pin)

    def mute_member(self, pin, muted):
        self.L.orc_conference_mute_member(C.byref(self.c), int(pin), int(muted))

    def muted(self, pin):
        return bool(self.c.muted[pin])

    @property
    def size(self):
        return self.L.orc_conference_get_size(C.byref(self.c))

    @property
    def active_speaker(self):
        return self.c.active_speaker

    def participant_volume(self, pin, volume_db):
        return self.L.orc_conference_participant_volume(C.byref(self.c), int(pin), float(volume_db))

    def process_events(self, max_db_by_pin):
        """max_db_by_pin: {pin: MS_VOLUME_GET_MAX of that member}.  -> (changed, winner pin or -1, its dB)"""
        order = (C.c_int * max(1, len(self.order)))(*self.order)
        db = (C.c_float * 50)(*([-120.0] * 50))
        for pin, v in max_db_by_pin.items():
            db[pin] = v
        wp, wd = C.c_int(), C.c_float()
        ch = self.L.orc_conference_process_events(C.byref(self.c), order, db, C.byref(wp), C.byref(wd))
        return bool(ch), wp.value, wd.value

