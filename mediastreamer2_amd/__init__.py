"""mediastreamer2_amd -- MI355X (gfx950) batched DSP backend for mediastreamer2's
audio/video filter hot path.

This package is a thin Python view of the C ABI in include/msmi355x.h
(libmsmi355x.so, hand-written HIP kernels in csrc/).  Python/PyTorch is used
by tests and bench.py only for device buffers, streams and torch.distributed;
the product is the shared library and the MSFilter facades in host/.

Numpy arrays go through the *_host entry points (H2D + kernel + D2H);
torch CUDA(HIP) tensors and raw device pointers go through the device-resident
entry points and stay asynchronous on the context stream.

Lifetime rule for device tensors: a Context created without `stream=` launches on its OWN HIP stream, which
torch's caching allocator does not know about -- keep every tensor handed to a process() call alive (and do
not write to it from torch) until ctx.sync(); temporaries such as `x[:, a:b].contiguous()` passed inline are
freed and reused by torch while the kernel may still be reading them.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MiError, VolumeParams, VolumeState, check, load  # noqa: F401

MI_MIX_LINKED, MI_MIX_ACTIVE, MI_MIX_OUTPUT = 1, 2, 4
MI_PIX_I420, MI_PIX_RGB24 = 0, 1
MI_AEC_POSTFILTER = 1


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr()
    raise TypeError(type(x))


def _on_device(x):
    return isinstance(x, int) or (_is_torch(x) and x.is_cuda)


class Context:
    """mi_ctx: one per GPU; launches go to `stream` (a hipStream_t handle, e.g.
    torch.cuda.current_stream().cuda_stream) or to a private stream."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        check(self.L.mi_ctx_create(int(device), stream, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def sync(self):
        check(self.L.mi_ctx_sync(self.h))

    @property
    def stream(self):
        return self.L.mi_ctx_stream(self.h)

    def props(self):
        cu, hbm, name = C.c_int(), C.c_size_t(), C.create_string_buffer(128)
        check(self.L.mi_ctx_props(self.h, C.byref(cu), C.byref(hbm), name, 128))
        return {"cu_count": cu.value, "hbm_bytes": hbm.value, "name": name.value.decode()}

    def capture_begin(self):
        check(self.L.mi_ctx_capture_begin(self.h))

    def capture_end(self):
        g = C.c_void_p()
        check(self.L.mi_ctx_capture_end(self.h, C.byref(g)))
        return Graph(self, g)

    def timer_start(self):
        check(self.L.mi_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(self.L.mi_timer_stop(self.h, C.byref(ms)))
        return ms.value


class Graph:
    """A captured sequence of mi_* launches (hipGraph)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def launch(self):
        check(self.ctx.L.mi_graph_launch(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.mi_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Batch:
    _destroy = None

    def close(self):
        if getattr(self, "h", None) and self._destroy:
            getattr(self.ctx.L, self._destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResamplerBatch(_Batch):
    """nstreams MSResample cores (msresample.c:102-177)."""
    _destroy = "mi_resampler_destroy"

    def __init__(self, ctx, nstreams, in_rate, out_rate, quality=3):
        self.ctx, self.nstreams, self.in_rate, self.out_rate = ctx, nstreams, in_rate, out_rate
        h = C.c_void_p()
        check(ctx.L.mi_resampler_create(ctx.h, nstreams, in_rate, out_rate, quality, C.byref(h)))
        self.h = h

    def info(self):
        v = [C.c_int() for _ in range(4)]
        check(self.ctx.L.mi_resampler_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("filt_len", "den_rate", "num_rate", "direct"), (x.value for x in v)))

    def table(self):
        n = self.ctx.L.mi_resampler_get_table(self.h, None, 0)
        t = np.zeros(n, np.float32)
        self.ctx.L.mi_resampler_get_table(self.h, t.ctypes.data, n)
        return t

    def out_capacity(self, in_len):
        return self.ctx.L.mi_resampler_out_capacity(self.h, in_len)

    def get_state(self, stream):
        """one stream's running state (position + history) as bytes: mi_resampler_get_state"""
        n = self.ctx.L.mi_resampler_state_bytes(self.h)
        buf = (C.c_uint8 * n)()
        check(self.ctx.L.mi_resampler_get_state(self.h, int(stream), buf, n))
        return bytes(buf)

    def set_state(self, stream, state):
        buf = (C.c_uint8 * len(state)).from_buffer_copy(state)
        check(self.ctx.L.mi_resampler_set_state(self.h, int(stream), buf, len(state)))

    def get_states(self, first, count):
        """the states of streams [first, first + count) back to back, one round trip: mi_resampler_get_states"""
        n = self.ctx.L.mi_resampler_state_bytes(self.h) * count
        buf = (C.c_uint8 * n)()
        check(self.ctx.L.mi_resampler_get_states(self.h, int(first), int(count), buf, n))
        return bytes(buf)

    def set_states(self, first, count, states):
        buf = (C.c_uint8 * len(states)).from_buffer_copy(states)
        check(self.ctx.L.mi_resampler_set_states(self.h, int(first), int(count), buf, len(states)))

    def reset(self, first=0, count=None):
        check(self.ctx.L.mi_resampler_reset(self.h, first, self.nstreams - first if count is None else count))

    def process(self, x, out=None, out_len=None):
        """x [nstreams, in_len] int16 (numpy -> host path, torch cuda -> device path).
        Returns (out [nstreams, out_stride], out_len [nstreams] or None)."""
        n, in_len = x.shape
        assert n == self.nstreams
        cap = self.out_capacity(in_len)
        ostride = (cap + 7) & ~7
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, np.int16)
            out = np.zeros((n, ostride), np.int16)
            ol = np.zeros(n, np.int32)
            check(self.ctx.L.mi_resampler_process_host(self.h, _ptr(x), in_len, x.shape[1], _ptr(out), ostride, _ptr(ol)))
            return out, ol
        import torch
        if out is None:
            out = torch.zeros((n, ostride), dtype=torch.int16, device=x.device)
            torch.cuda.current_stream().synchronize()  # torch filled it on ITS stream; the kernel below writes it on the context's
        check(self.ctx.L.mi_resampler_process(self.h, _ptr(x), in_len, x.stride(0), _ptr(out), out.stride(0), _ptr(out_len)))
        return out, out_len


class MixerBatch(_Batch):
    """nconf MSAudioMixer ticks (audiomixer.c:288-346)."""
    _destroy = "mi_mixer_destroy"

    def __init__(self, ctx, nconf, max_members, nsamples):
        self.ctx, self.nconf, self.mm, self.ns = ctx, nconf, max_members, nsamples
        h = C.c_void_p()
        check(ctx.L.mi_mixer_create(ctx.h, nconf, max_members, nsamples, C.byref(h)))
        self.h = h

    def set_controls(self, flags=None, gain=None):
        f = None if flags is None else np.ascontiguousarray(flags, np.uint8).reshape(self.nconf, self.mm)
        g = None if gain is None else np.ascontiguousarray(gain, np.float32).reshape(self.nconf, self.mm)
        check(self.ctx.L.mi_mixer_set_controls(self.h, _ptr(f), _ptr(g)))

    def process(self, x, has_data=None, conf_mode=1, out=None):
        """x [nconf, mm, ns] int16."""
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, np.int16)
            hd = None if has_data is None else np.ascontiguousarray(has_data, np.uint8)
            if out is None:
                out = np.zeros(x.shape if conf_mode else (self.nconf, self.ns), np.int16)
            check(self.ctx.L.mi_mixer_process_host(self.h, _ptr(x), _ptr(hd), int(conf_mode), _ptr(out)))
            return out
        import torch
        if out is None:
            out = torch.zeros(tuple(x.shape) if conf_mode else (self.nconf, self.ns), dtype=torch.int16, device=x.device)
            torch.cuda.current_stream().synchronize()  # torch filled it on ITS stream; the kernel below writes it on the context's
        check(self.ctx.L.mi_mixer_process(self.h, _ptr(x), _ptr(has_data), int(conf_mode), _ptr(out)))
        return out

    def process_volume_fifo(self, vol, fifo, out, first_stream=0, dry_skips=False, run=None):
        """MSVolume + the conference mix in one launch: every pin's chunk popped from `fifo`, levelled by `vol` (stream
        first_stream + conference * members + pin), mixed in conference mode into out [nconf, mm, ns]; dry_skips: a pin
        whose queue holds less than a tick is not metered (MI_VOLMIX_DRY_SKIPS)"""
        check(self.ctx.L.mi_mixer_process_volume_fifo_flags(self.h, vol.h, first_stream, fifo.h, _ptr(out), 1 if dry_skips else 0, _ptr(run)))
        return out

    def partial_sum(self, x, d_sum, has_data=None):
        check(self.ctx.L.mi_mixer_partial_sum(self.h, _ptr(x), _ptr(has_data), _ptr(d_sum)))
        return d_sum

    def finalize(self, x, d_sum, out, has_data=None, conf_mode=1):
        check(self.ctx.L.mi_mixer_finalize(self.h, _ptr(x), _ptr(has_data), _ptr(d_sum), int(conf_mode), _ptr(out)))
        return out


class Exchange(_Batch):
    """mi_exchange: the split conferences' int32 all-reduce on RCCL, enqueued on the context's stream."""
    _destroy = "mi_exchange_destroy"

    @staticmethod
    def unique_id(ctx):
        buf = np.zeros(128, np.uint8)
        check(ctx.L.mi_exchange_unique_id(buf.ctypes.data, buf.size))
        return buf.tobytes()

    def __init__(self, ctx, nranks, rank, unique_id):
        self.ctx, self.nranks, self.rank = ctx, nranks, rank
        buf = np.frombuffer(unique_id, np.uint8).copy()
        assert buf.size == 128
        h = C.c_void_p()
        check(ctx.L.mi_exchange_create(ctx.h, nranks, rank, buf.ctypes.data, C.byref(h)))
        self.h = h

    def allreduce(self, d_sum):
        """d_sum: contiguous int32 device tensor, summed in place over the ranks"""
        check(self.ctx.L.mi_exchange_allreduce_i32(self.h, _ptr(d_sum), d_sum.numel()))
        return d_sum

    __call__ = allreduce


class VolumeBatch(_Batch):
    """nstreams MSVolume chunks (msvolume.c:471-514)."""
    _destroy = "mi_volume_destroy"

    def __init__(self, ctx, nstreams, sample_rate):
        self.ctx, self.nstreams, self.rate = ctx, nstreams, sample_rate
        h = C.c_void_p()
        check(ctx.L.mi_volume_create(ctx.h, nstreams, sample_rate, C.byref(h)))
        self.h = h

    @staticmethod
    def default_params():
        p = VolumeParams()
        load().mi_volume_default_params(C.byref(p))
        return p

    def set_params(self, params, first=0):
        arr = (VolumeParams * len(params))(*params)
        check(self.ctx.L.mi_volume_set_params(self.h, first, len(params), arr))

    def set_peer_batch(self, peers):
        """streams whose params.peer is PEER_EXTERNAL (-2) read the energy of the same index in `peers` (another VolumeBatch, or None):
        msvolume.c:201-238 reads its peer FILTER's energy, whatever batch that filter lives in"""
        check(self.ctx.L.mi_volume_set_peer_batch(self.h, peers.h if peers is not None else None))

    def get_state(self, first=0, count=None):
        count = self.nstreams - first if count is None else count
        arr = (VolumeState * count)()
        check(self.ctx.L.mi_volume_get_state(self.h, first, count, arr))
        return list(arr)

    def set_state(self, states, first=0):
        arr = (VolumeState * len(states))(*states)
        check(self.ctx.L.mi_volume_set_state(self.h, first, len(states), arr))

    def get_max(self, first=0, count=None):
        """MS_VOLUME_GET_MAX, linear: maximum of the smoothed energy over the last second (device-side window)"""
        count = self.nstreams - first if count is None else count
        out = np.zeros(count, np.float32)
        check(self.ctx.L.mi_volume_get_max(self.h, first, count, out.ctypes.data))
        return out

    def reset_max(self, first=0, count=None):
        check(self.ctx.L.mi_volume_reset_max(self.h, first, self.nstreams - first if count is None else count))

    def process(self, x, nsamples=None, per_stream=None):
        """x [nstreams, stride] int16, modified in place (numpy: returns the array)."""
        n, stride = x.shape
        nsamples = stride if nsamples is None else nsamples
        if isinstance(x, np.ndarray):
            assert x.dtype == np.int16 and x.flags["C_CONTIGUOUS"]
            ps = None if per_stream is None else np.ascontiguousarray(per_stream, np.int32)
            check(self.ctx.L.mi_volume_process_host(self.h, _ptr(x), nsamples, stride, _ptr(ps)))
            return x
        check(self.ctx.L.mi_volume_process(self.h, _ptr(x), nsamples, x.stride(0), _ptr(per_stream)))
        return x


    def process_fifo(self, fifo, out, nsamples=None, first=None, count=None):
        """pop one chunk per stream from a FifoBatch (silence where it holds less), process, write to out's rows;
        first / count: only that range of streams"""
        n = out.shape[1] if nsamples is None else nsamples
        if first is None:
            check(self.ctx.L.mi_volume_process_fifo(self.h, fifo.h, _ptr(out), n, out.stride(0)))
        else:
            check(self.ctx.L.mi_volume_process_fifo_range(self.h, fifo.h, _ptr(out), n, out.stride(0), first,
                                                          self.nstreams - first if count is None else count))
        return out


class EqualizerBatch(_Batch):
    """nstreams MSEqualizer FIRs (equalizer.c:263-288, dsptools.c:253-268)."""
    _destroy = "mi_equalizer_destroy"

    def __init__(self, ctx, nstreams, sample_rate):
        self.ctx, self.nstreams, self.rate = ctx, nstreams, sample_rate
        h = C.c_void_p()
        check(ctx.L.mi_equalizer_create(ctx.h, nstreams, sample_rate, C.byref(h)))
        self.h = h
        self.fir_len = ctx.L.mi_equalizer_fir_len(h)

    def set_gain(self, stream, freq, gain, width):
        check(self.ctx.L.mi_equalizer_set_gain(self.h, stream, freq, gain, width))

    def flatten(self, stream):
        check(self.ctx.L.mi_equalizer_flatten(self.h, stream))

    def set_active(self, stream, active):
        check(self.ctx.L.mi_equalizer_set_active(self.h, stream, int(active)))

    def dump(self, stream):
        a = np.zeros(self.fir_len // 2, np.float32)
        check(self.ctx.L.mi_equalizer_dump(self.h, stream, _ptr(a), len(a)))
        return a

    def taps(self, stream):
        a = np.zeros(self.fir_len, np.float32)
        check(self.ctx.L.mi_equalizer_get_taps(self.h, stream, _ptr(a), len(a)))
        return a

    def set_taps(self, stream, taps):
        t = np.ascontiguousarray(taps, np.float32)
        check(self.ctx.L.mi_equalizer_set_taps(self.h, stream, _ptr(t), len(t)))

    def history(self, stream):
        """the FIR's memory (ms_fir_mem16's mem): fir_len int16"""
        a = np.zeros(self.fir_len, np.int16)
        check(self.ctx.L.mi_equalizer_get_history(self.h, stream, _ptr(a), len(a)))
        return a

    def set_history(self, stream, hist=None):
        if hist is None:
            check(self.ctx.L.mi_equalizer_set_history(self.h, stream, None, self.fir_len))
        else:
            a = np.ascontiguousarray(hist, np.int16)
            check(self.ctx.L.mi_equalizer_set_history(self.h, stream, _ptr(a), len(a)))

    def process(self, x, nsamples=None):
        n, stride = x.shape
        nsamples = stride if nsamples is None else nsamples
        if isinstance(x, np.ndarray):
            assert x.dtype == np.int16 and x.flags["C_CONTIGUOUS"]
            check(self.ctx.L.mi_equalizer_process_host(self.h, _ptr(x), nsamples, stride))
            return x
        check(self.ctx.L.mi_equalizer_process(self.h, _ptr(x), nsamples, x.stride(0)))
        return x


class AecBatch(_Batch):
    """nstreams MSSpeexEC cores (speexec.c:188-305): MDF canceller + post-filter."""
    _destroy = "mi_aec_destroy"

    def __init__(self, ctx, nstreams, sample_rate, frame_size=None, filter_length=None, tail_ms=128):
        self.ctx, self.nstreams, self.rate = ctx, nstreams, sample_rate
        if frame_size is None:
            frame_size = ctx.L.mi_aec_framesize(64, sample_rate)  # speexec.c:41,171-180
        if filter_length is None:
            filter_length = tail_ms * sample_rate // 1000  # speexec.c:194
        self.frame, self.filter_length = frame_size, filter_length
        h = C.c_void_p()
        check(ctx.L.mi_aec_create(ctx.h, nstreams, sample_rate, frame_size, filter_length, C.byref(h)))
        self.h = h

    def state_bytes(self):
        return self.ctx.L.mi_aec_state_bytes(self.h)

    def export_state(self, stream):
        """One stream's whole state as bytes (speexec.c:145-167 fetch_config)."""
        n = self.ctx.L.mi_aec_blob_bytes(self.h)
        buf = np.zeros(n, np.uint8)
        check(self.ctx.L.mi_aec_export_state(self.h, stream, buf.ctypes.data, n))
        return buf.tobytes()

    def stagger_info(self, tick_len):
        u, p = C.c_int(), C.c_int()
        check(self.ctx.L.mi_aec_stagger_info(self.h, tick_len, C.byref(u), C.byref(p)))
        return u.value, p.value

    def stagger_fifos(self, f_mic, f_ref, tick_len, first=0, count=None):
        """lead of unit * phase(stream) samples of silence in both queues of (freshly reset) legs: mi_aec_stagger_fifos"""
        count = self.nstreams - first if count is None else count
        check(self.ctx.L.mi_aec_stagger_fifos(self.h, f_mic.h, f_ref.h, tick_len, first, count))

    def copy_state_from(self, src, src_first=0, dst_first=0, count=None):
        """device-to-device copy of `count` streams' whole state from another batch of the same shape"""
        count = min(src.nstreams - src_first, self.nstreams - dst_first) if count is None else count
        check(self.ctx.L.mi_aec_copy_state(self.h, dst_first, src.h, src_first, count))

    def import_state(self, stream, blob):
        """speexec.c:121-143 apply_config: raises MiError for a blob of another shape."""
        buf = np.frombuffer(blob, np.uint8).copy()
        check(self.ctx.L.mi_aec_import_state(self.h, stream, buf.ctypes.data, buf.size))

    def reset(self, first=0, count=None):
        check(self.ctx.L.mi_aec_reset(self.h, first, self.nstreams - first if count is None else count))

    def get(self, stream, what, n):
        a = np.zeros(n, np.float32)
        got = self.ctx.L.mi_aec_get(self.h, stream, what.encode(), _ptr(a), n)
        if got < 0:
            check(got)
        return a[:got]

    def process(self, mic, ref, out=None, run=None, flags=MI_AEC_POSTFILTER):
        n, stride = mic.shape
        if isinstance(mic, np.ndarray):
            mic = np.ascontiguousarray(mic, np.int16)
            ref = np.ascontiguousarray(ref, np.int16)
            out = np.zeros_like(mic) if out is None else out
            r = None if run is None else np.ascontiguousarray(run, np.uint8)
            check(self.ctx.L.mi_aec_process_host(self.h, _ptr(mic), _ptr(ref), _ptr(out), stride, _ptr(r), flags))
            return out
        import torch
        if out is None:
            out = torch.zeros_like(mic)
            torch.cuda.current_stream().synchronize()  # torch filled it on ITS stream; the kernel below writes it on the context's
        check(self.ctx.L.mi_aec_process(self.h, _ptr(mic), _ptr(ref), _ptr(out), mic.stride(0), _ptr(run), flags))
        return out


    def process_fifos(self, f_mic, mic_tick, f_ref, ref_tick, f_out, tick_len=None, max_frames=2, flags=MI_AEC_POSTFILTER, count_out=None,
                      ref_len=None):
        """The tick with the FIFOs folded in: both new blocks queued, every whole frame cancelled, the results queued on
        f_out -- one launch (mi_aec_process_fifos).  FifoBatch objects whose capacities are multiples of the frame size."""
        n = mic_tick.shape[1] if tick_len is None else tick_len
        check(self.ctx.L.mi_aec_process_fifos(self.h, f_mic.h, _ptr(mic_tick), mic_tick.stride(0), f_ref.h, _ptr(ref_tick),
                                              ref_tick.stride(0), _ptr(ref_len), n, f_out.h, max_frames, flags, _ptr(count_out)))

    def process_fifos_resampled(self, rs, mic_in, f_mic, f_ref, ref_tick, f_out, in_len=None, max_frames=2, flags=MI_AEC_POSTFILTER,
                                count_out=None, ref_len=None, mic_gate=None):
        """process_fifos with the leg's up-sampler (a ResamplerBatch) folded into the same launch: mic_in holds the block at
        the resampler's input rate (mi_aec_process_fifos_resampled[_masked]: mic_gate [nstreams] uint8, 0 = no block for that leg)."""
        n = mic_in.shape[1] if in_len is None else in_len
        check(self.ctx.L.mi_aec_process_fifos_resampled_masked(self.h, rs.h, _ptr(mic_in), n, mic_in.stride(0), f_mic.h, f_ref.h, _ptr(ref_tick),
                                                               ref_tick.stride(0), _ptr(ref_len), f_out.h, max_frames, flags, _ptr(count_out), _ptr(mic_gate)))

    def process_frames(self, mic, ref, out, count, max_frames=2, flags=MI_AEC_POSTFILTER):
        """The frames of one tick in one launch: rows of mic / ref / out hold up to max_frames frames back to back,
        count [nstreams] uint8 (device) = frames ready per stream.  Device tensors only."""
        check(self.ctx.L.mi_aec_process_frames(self.h, _ptr(mic), _ptr(ref), _ptr(out), mic.stride(0), _ptr(count), max_frames, flags))
        return out


class ScalerBatch(_Batch):
    """MSScalerDesc context (msvideo.h:473-478) for a batch of I420 frames."""
    _destroy = "mi_scaler_destroy"

    def __init__(self, ctx, sw, sh, dw, dh, dst_fmt=MI_PIX_RGB24):
        self.ctx = ctx
        self.sw, self.sh, self.dw, self.dh, self.fmt = sw, sh, dw, dh, dst_fmt
        h = C.c_void_p()
        check(ctx.L.mi_scaler_create(ctx.h, sw, sh, dw, dh, dst_fmt, C.byref(h)))
        self.h = h
        self.src_bytes = ctx.L.mi_scaler_src_bytes(h)
        self.dst_bytes = ctx.L.mi_scaler_dst_bytes(h)

    def process(self, src, out=None):
        """src [nframes, src_bytes] uint8."""
        nf = src.shape[0]
        if isinstance(src, np.ndarray):
            src = np.ascontiguousarray(src, np.uint8)
            out = np.zeros((nf, self.dst_bytes), np.uint8) if out is None else out
            check(self.ctx.L.mi_scaler_process_host(self.h, nf, _ptr(src), src.shape[1], _ptr(out), out.shape[1]))
            return out
        import torch
        if out is None:
            out = torch.zeros((nf, self.dst_bytes), dtype=torch.uint8, device=src.device)
            torch.cuda.current_stream().synchronize()  # torch filled it on ITS stream; the kernel below writes it on the context's
        check(self.ctx.L.mi_scaler_process(self.h, nf, _ptr(src), src.stride(0), _ptr(out), out.stride(0)))
        return out

    def process_planes(self, src_planes, src_strides, dst_planes, dst_strides):
        """MSScalerDesc.context_process argument shape: numpy uint8 arrays per plane + strides (one frame)."""
        vp = C.c_void_p
        sp = (vp * 3)(*[vp(p.ctypes.data) for p in src_planes])
        dp = (vp * 3)(*[vp(p.ctypes.data) if p is not None else vp(0) for p in (list(dst_planes) + [None, None])[:3]])
        ss = (C.c_int32 * 3)(*src_strides)
        ds = (C.c_int32 * 3)(*(list(dst_strides) + [0, 0])[:3])
        check(self.ctx.L.mi_scaler_process_planes_host(self.h, sp, ss, dp, ds))


class ScalerPipe(_Batch):
    """mi_scaler_pipe: the scaler fed from host buffers, upload | kernel | download overlapped on three streams, up to
    `depth` batches in flight.  acquire() -> numpy view [batch, src_pitch] of the pinned staging to fill in place;
    submit(n); collect() -> numpy view [n, dst_pitch] of the pinned results of the OLDEST batch."""
    _destroy = "mi_scaler_pipe_destroy"

    def __init__(self, scaler, batch_frames, depth=3):
        self.ctx, self.scaler, self.batch, self.depth = scaler.ctx, scaler, batch_frames, depth
        h = C.c_void_p()
        check(self.ctx.L.mi_scaler_pipe_create(scaler.h, batch_frames, depth, C.byref(h)))
        self.h = h

    def acquire(self):
        p, pitch = C.c_void_p(), C.c_size_t()
        check(self.ctx.L.mi_scaler_pipe_acquire(self.h, C.byref(p), C.byref(pitch)))
        buf = (C.c_uint8 * (self.batch * pitch.value)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(self.batch, pitch.value)

    def submit(self, nframes):
        check(self.ctx.L.mi_scaler_pipe_submit(self.h, int(nframes)))

    def collect(self):
        p, pitch, n = C.c_void_p(), C.c_size_t(), C.c_int32()
        check(self.ctx.L.mi_scaler_pipe_collect(self.h, C.byref(p), C.byref(pitch), C.byref(n)))
        buf = (C.c_uint8 * (n.value * pitch.value)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(n.value, pitch.value)

    def in_flight(self):
        return self.ctx.L.mi_scaler_pipe_in_flight(self.h)


MI_PIX_YUY2, MI_PIX_UYVY, MI_PIX_BGR24, MI_PIX_RGB24_RAW, MI_PIX_BGRA32 = 2, 3, 4, 5, 6


class PixConvBatch(_Batch):
    """MSPixConv's conversion (pixconv.c:62-94 through yuv_scale msvideo.c:542-581) for a batch of packed frames."""
    _destroy = "mi_pixconv_destroy"

    def __init__(self, ctx, w, h, src_fmt, flip=False):
        self.ctx = ctx
        self.w, self.height, self.fmt = w, h, src_fmt
        hd = C.c_void_p()
        check(ctx.L.mi_pixconv_create(ctx.h, w, h, src_fmt, 1 if flip else 0, C.byref(hd)))
        self.h = hd
        self.src_bytes = ctx.L.mi_pixconv_src_bytes(hd)
        self.dst_bytes = ctx.L.mi_pixconv_dst_bytes(hd)

    def process(self, src, out=None):
        """src [nframes, src_bytes] uint8 -> [nframes, dst_bytes] I420."""
        nf = src.shape[0]
        if isinstance(src, np.ndarray):
            src = np.ascontiguousarray(src, np.uint8)
            out = np.zeros((nf, self.dst_bytes), np.uint8) if out is None else out
            check(self.ctx.L.mi_pixconv_process_host(self.h, nf, _ptr(src), src.shape[1], _ptr(out), out.shape[1]))
            return out
        import torch
        if out is None:
            out = torch.zeros((nf, self.dst_bytes), dtype=torch.uint8, device=src.device)
            torch.cuda.current_stream().synchronize()  # torch filled it on ITS stream; the kernel below writes it on the context's
        check(self.ctx.L.mi_pixconv_process(self.h, nf, _ptr(src), src.stride(0), _ptr(out), out.stride(0)))
        return out


class FifoBatch(_Batch):
    """nstreams device-resident MSBufferizers (msqueue.c:70-113): torch tensors in, torch tensors out."""
    _destroy = "mi_fifo_destroy"

    def __init__(self, ctx, nstreams, capacity):
        self.ctx, self.nstreams, self.capacity = ctx, nstreams, capacity
        h = C.c_void_p()
        check(ctx.L.mi_fifo_create(ctx.h, nstreams, capacity, C.byref(h)))
        self.h = h

    def push(self, x, nsamples=None, count=None, gate=None):
        """x [nstreams, >=nsamples] int16 (device); count [nstreams] int32 or gate [nstreams] uint8 (device) or None."""
        n = x.shape[1] if nsamples is None else nsamples
        if gate is not None:
            check(self.ctx.L.mi_fifo_push_gated(self.h, _ptr(x), n, x.stride(0), _ptr(gate)))
        else:
            check(self.ctx.L.mi_fifo_push(self.h, _ptr(x), n, x.stride(0), _ptr(count)))

    def pop(self, frame, out, ok=None, gate=None, zero_fill=True):
        check(self.ctx.L.mi_fifo_pop(self.h, frame, _ptr(out), out.stride(0), _ptr(ok), _ptr(gate), 1 if zero_fill else 0))
        return out

    def pop_frames(self, frame, max_frames, out, nframes_out=None, wanted=None, zero_fill=True):
        """up to max_frames whole frames per stream, back to back in out's rows; see mi_fifo_pop_frames"""
        check(self.ctx.L.mi_fifo_pop_frames(self.h, frame, max_frames, _ptr(out), out.stride(0), _ptr(nframes_out), _ptr(wanted),
                                            1 if zero_fill else 0))
        return out

    def push_frames(self, x, frame, max_frames, nframes):
        check(self.ctx.L.mi_fifo_push_frames(self.h, _ptr(x), frame, max_frames, x.stride(0), _ptr(nframes)))

    def levels(self, out):
        check(self.ctx.L.mi_fifo_levels(self.h, _ptr(out)))
        return out

    def overflows(self):
        n = C.c_int32(0)
        check(self.ctx.L.mi_fifo_overflows(self.h, C.byref(n)))
        return n.value

    def snapshot(self):
        """(rings [nstreams, capacity] int16, head [nstreams], level [nstreams]) as they lie on the device (parity read-back; syncs)"""
        rings = np.zeros((self.nstreams, self.capacity), np.int16)
        head, level = np.zeros(self.nstreams, np.int32), np.zeros(self.nstreams, np.int32)
        check(self.ctx.L.mi_fifo_snapshot(self.h, _ptr(rings), _ptr(head), _ptr(level)))
        return rings, head, level

    def export_range(self, first, count):
        """the queues of streams [first, first + count) as a host sees bufferizers: a list of int16 arrays, oldest sample first (syncs)"""
        x = np.zeros((count, self.capacity), np.int16)
        level = np.zeros(count, np.int32)
        check(self.ctx.L.mi_fifo_export_range(self.h, first, count, _ptr(x), self.capacity, _ptr(level)))
        return [x[k, :level[k]].copy() for k in range(count)]

    def import_range(self, first, queues, tail_at_end=False):
        """... and back: every stream of the range holds exactly its list entry (mi_fifo_import_range)"""
        x = np.zeros((len(queues), self.capacity), np.int16)
        level = np.zeros(len(queues), np.int32)
        for k, q in enumerate(queues):
            x[k, :len(q)] = q
            level[k] = len(q)
        check(self.ctx.L.mi_fifo_import_range(self.h, first, len(queues), _ptr(x), self.capacity, _ptr(level), int(tail_at_end)))

    def push_silence(self, count):
        """count [nstreams] int32 (device): samples of silence appended per stream (mi_fifo_push_silence)"""
        check(self.ctx.L.mi_fifo_push_silence(self.h, _ptr(count)))

    def reset_range(self, first, count):
        check(self.ctx.L.mi_fifo_reset_range(self.h, first, count))

    def reset_range_at(self, first, count, head):
        check(self.ctx.L.mi_fifo_reset_range_at(self.h, first, count, head))

    def reset(self):
        check(self.ctx.L.mi_fifo_reset(self.h))


MI_LAW_PCMA, MI_LAW_PCMU = 0, 1
MI_CHAN_MONO_TO_STEREO, MI_CHAN_STEREO_TO_MONO, MI_CHAN_TWO_MONO_TO_STEREO = 0, 1, 2
MI_FLOWCTL_BASIC, MI_FLOWCTL_SOFT = 0, 1


def g711_decode(ctx, law, codes, pcm, length=None, lens=None):
    """codes [rows, >=length] uint8 -> pcm [rows, >=length] int16, device tensors; lens [rows] int32 (device) or None."""
    n = codes.shape[1] if length is None else length
    check(ctx.L.mi_g711_decode(ctx.h, law, _ptr(codes), codes.stride(0), _ptr(pcm), pcm.stride(0), _ptr(lens), n, codes.shape[0]))
    return pcm


def g711_encode(ctx, law, pcm, codes, length=None, lens=None):
    n = pcm.shape[1] if length is None else length
    check(ctx.L.mi_g711_encode(ctx.h, law, _ptr(pcm), pcm.stride(0), _ptr(codes), codes.stride(0), _ptr(lens), n, pcm.shape[0]))
    return codes


def l16_swap(ctx, x, out):
    check(ctx.L.mi_l16_swap(ctx.h, _ptr(x), _ptr(out), x.numel()))
    return out


def chan_adapt(ctx, mode, a, out, b=None):
    frames = a.numel() // 2 if mode == MI_CHAN_STEREO_TO_MONO else a.numel()
    check(ctx.L.mi_chan_adapt(ctx.h, mode, _ptr(a), _ptr(b), _ptr(out), frames))
    return out


class FlowControlBatch(_Batch):
    """nstreams MSAudioFlowControllers (flowcontrol.c:30-152) with their state on the device."""
    _destroy = "mi_flowctl_destroy"

    def __init__(self, ctx, nstreams, max_block):
        self.ctx, self.nstreams = ctx, nstreams
        h = C.c_void_p()
        check(ctx.L.mi_flowctl_create(ctx.h, nstreams, max_block, C.byref(h)))
        self.h = h

    def set_config(self, strategy, silent_threshold, first=0, count=None):
        check(self.ctx.L.mi_flowctl_set_config(self.h, first, self.nstreams - first if count is None else count, strategy, silent_threshold))

    def request_drop(self, samples_to_drop, total_samples):
        d = np.ascontiguousarray(samples_to_drop, np.uint32)
        t = np.ascontiguousarray(total_samples, np.uint32)
        assert d.size == self.nstreams and t.size == self.nstreams
        check(self.ctx.L.mi_flowctl_request_drop(self.h, d.ctypes.data, t.ctypes.data))

    def process(self, x, out, out_len, length=None, lens=None):
        n = x.shape[1] if length is None else length
        check(self.ctx.L.mi_flowctl_process(self.h, _ptr(x), x.stride(0), _ptr(lens), n, _ptr(out), out.stride(0), _ptr(out_len)))
        return out, out_len

    def state(self, stream):
        v = (C.c_uint32 * 4)()
        check(self.ctx.L.mi_flowctl_get_state(self.h, stream, v))
        return dict(target=v[0], total=v[1], pos=v[2], dropped=v[3])

    def reset(self, first=0, count=None):
        check(self.ctx.L.mi_flowctl_reset(self.h, first, self.nstreams - first if count is None else count))


MI_PLC_NONE, MI_PLC_RECEIVED, MI_PLC_CONCEAL, MI_PLC_CNG_RESUME = 0, 1, 2, 4


class PlcBatch(_Batch):
    """nstreams plc_context_t (genericplc.c) on the device: process(rows, lens, modes) edits / fills the rows in place."""
    _destroy = "mi_plc_destroy"

    def __init__(self, ctx, nstreams, rate, max_block=960):
        self.ctx, self.nstreams, self.rate = ctx, nstreams, rate
        h = C.c_void_p()
        check(ctx.L.mi_plc_create(ctx.h, nstreams, rate, max_block, C.byref(h)))
        self.h = h

    def process(self, rows, lens, modes):
        check(self.ctx.L.mi_plc_process(self.h, _ptr(rows), rows.stride(0), _ptr(lens), _ptr(modes)))
        return rows

    def info(self, stream):
        v = (C.c_int32 * 3)()
        check(self.ctx.L.mi_plc_info(self.h, stream, v))
        return dict(nb=v[0], index=v[1], used=v[2])

    def reset(self, first=0, count=None):
        check(self.ctx.L.mi_plc_reset(self.h, first, self.nstreams - first if count is None else count))


class SessionConfig(C.Structure):
    _fields_ = [("nstreams", C.c_int32), ("members_per_conference", C.c_int32), ("in_rate", C.c_int32),
                ("rate", C.c_int32), ("tail_ms", C.c_int32), ("agc", C.c_int32), ("use_graphs", C.c_int32),
                ("mic_codec", C.c_int32), ("out_rate", C.c_int32), ("out_codec", C.c_int32),
                ("ref_loopback", C.c_int32), ("ref_delay_ms", C.c_int32), ("plc", C.c_int32), ("stagger", C.c_int32)]


MI_SESSION_PCM16, MI_SESSION_PCMA, MI_SESSION_PCMU = 0, 1, 2


class Session(_Batch):
    """mi_session: the chained path ([G.711 ->] resample -> AEC -> AGC -> conference mix [-> resample -> G.711]) fed
    from host buffers, three ticks in flight on three HIP streams."""
    _destroy = "mi_session_destroy"

    def __init__(self, ctx, nstreams, members=32, in_rate=16000, rate=48000, tail_ms=128, agc=True, use_graphs=True,
                 mic_codec=0, out_rate=0, out_codec=0, ref_loopback=False, ref_delay_ms=0, plc=False, stagger=False):
        self.ctx = ctx
        cfg = SessionConfig()
        ctx.L.mi_session_default_config(C.byref(cfg))
        cfg.nstreams, cfg.members_per_conference, cfg.in_rate, cfg.rate = nstreams, members, in_rate, rate
        cfg.tail_ms, cfg.agc, cfg.use_graphs = tail_ms, int(agc), int(use_graphs)
        cfg.mic_codec, cfg.out_rate, cfg.out_codec = mic_codec, out_rate, out_codec
        cfg.ref_loopback, cfg.ref_delay_ms, cfg.plc = int(ref_loopback), ref_delay_ms, int(plc)
        cfg.stagger = int(stagger)  # (the C default is 1; the wrapper's tests compare with hand-made chains that start empty)
        h = C.c_void_p()
        check(ctx.L.mi_session_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self.n, self.in_len, self.len = nstreams, in_rate // 100, rate // 100
        self.members = members
        self.out_len = (out_rate or rate) // 100
        self.mic_dtype = C.c_uint8 if mic_codec else C.c_int16
        self.out_dtype = C.c_uint8 if out_codec else C.c_int16
        self.loopback = bool(ref_loopback)

    def _view(self, ptr, cols, ctype=C.c_int16):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(self.n, cols))

    def tick_bytes(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.ctx.L.mi_session_tick_bytes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def acquire(self):
        """numpy views of the pinned staging of the next tick: (mic [n, in_len], ref [n, len] or None with loopback)."""
        pm, pr = C.c_void_p(), C.c_void_p()
        check(self.ctx.L.mi_session_acquire(self.h, C.byref(pm), C.byref(pr)))
        return self._view(pm, self.in_len, self.mic_dtype), (None if self.loopback else self._view(pr, self.len))

    def events(self):
        """numpy view [n] uint8 of the tick being filled (plc sessions): MI_PLC_RECEIVED preset, set MI_PLC_CONCEAL for lost legs."""
        pe = C.c_void_p()
        check(self.ctx.L.mi_session_events(self.h, C.byref(pe)))
        return np.ctypeslib.as_array(C.cast(pe, C.POINTER(C.c_uint8)), shape=(self.n,))

    def submit(self):
        check(self.ctx.L.mi_session_submit(self.h))

    def collect(self):
        """numpy view of the oldest in-flight tick's output [n, out_len] (pinned; valid for three more submits)."""
        po = C.c_void_p()
        check(self.ctx.L.mi_session_collect(self.h, C.byref(po)))
        return self._view(po, self.out_len, self.out_dtype)

    def in_flight(self):
        return self.ctx.L.mi_session_in_flight(self.h)

    def set_controls(self, flags=None, gain=None):
        f = None if flags is None else np.ascontiguousarray(flags, np.uint8)
        g = None if gain is None else np.ascontiguousarray(gain, np.float32)
        check(self.ctx.L.mi_session_set_controls(self.h, _ptr(f), _ptr(g)))

    def reset_streams(self, first, count):
        check(self.ctx.L.mi_session_reset_streams(self.h, first, count))

    def levels(self):
        out = np.zeros(self.n, np.float32)
        check(self.ctx.L.mi_session_get_levels(self.h, _ptr(out)))
        return out

    def add_member(self, stream):
        """ms_audio_conference_add_member: a NEW endpoint takes the slot (fresh per-leg state), its mixer pin is plumbed."""
        check(self.ctx.L.mi_session_add_member(self.h, int(stream)))

    def remove_member(self, stream):
        """ms_audio_conference_remove_member: the pin is unplumbed; the other members carry on untouched."""
        check(self.ctx.L.mi_session_remove_member(self.h, int(stream)))

    def member_count(self, conference):
        n = self.ctx.L.mi_session_member_count(self.h, int(conference))
        check(min(n, 0))
        return n

    def active_speakers(self, now_ms):
        """(winner stream per conference or -1, its MS_VOLUME_GET_MAX in dBm0): audioconference.c:436-452"""
        nconf = self.n // self.members
        win = np.zeros(nconf, np.int32)
        db = np.zeros(nconf, np.float32)
        check(self.ctx.L.mi_session_active_speakers(self.h, C.c_uint64(int(now_ms)), _ptr(win), _ptr(db)))
        return win, db
