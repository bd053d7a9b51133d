"""ctypes loader for libmsmi355x.so (the C ABI of include/msmi355x.h).

There is no CPU fallback: if the shared library is missing it must be built
(`python -c "import __graft_entry__ as g; g.build()"`), and on a machine
without a HIP device every `mi_*_create` returns MI_ENODEV, which surfaces
here as MiError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsmi355x.so")

MI_OK, MI_EINVAL, MI_ENODEV, MI_ENOMEM, MI_ENOTSUP = 0, -1, -2, -3, -4


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmsmi355x error {code}: {msg}")
        self.code = code


# every symbol include/msmi355x.h declares (tests check the export table against this)
EXPORTS = [
    "mi_abi_version", "mi_last_error", "mi_device_count",
    "mi_ctx_create", "mi_ctx_destroy", "mi_ctx_sync", "mi_warmup", "mi_ctx_stream", "mi_ctx_device", "mi_ctx_props",
    "mi_dev_alloc", "mi_dev_free", "mi_host_alloc", "mi_host_free", "mi_copy_h2d", "mi_copy_d2h", "mi_copy_h2d_pinned", "mi_copy_d2h_pinned", "mi_memset",
    "mi_ctx_capture_begin", "mi_ctx_capture_end", "mi_graph_launch", "mi_graph_destroy",
    "mi_timer_start", "mi_timer_stop",
    "mi_resampler_create", "mi_resampler_destroy", "mi_resampler_reset", "mi_resampler_state_bytes", "mi_resampler_get_state", "mi_resampler_set_state", "mi_resampler_get_states", "mi_resampler_set_states", "mi_resampler_out_capacity",
    "mi_resampler_info", "mi_resampler_get_table", "mi_resampler_process", "mi_resampler_process_host",
    "mi_resampler_process_masked", "mi_mixer_process_masked", "mi_equalizer_process_masked",
    "mi_mixer_create", "mi_mixer_destroy", "mi_mixer_set_controls", "mi_mixer_process",
    "mi_mixer_process_host", "mi_mixer_partial_sum", "mi_mixer_finalize",
    "mi_exchange_unique_id", "mi_exchange_create", "mi_exchange_destroy", "mi_exchange_ranks", "mi_exchange_allreduce_i32",
    "mi_volume_create", "mi_volume_destroy", "mi_volume_default_params", "mi_volume_set_params",
    "mi_volume_get_state", "mi_volume_set_state", "mi_volume_set_peer_batch", "mi_volume_get_max", "mi_volume_reset_max", "mi_volume_process", "mi_volume_process_host", "mi_volume_process_fifo", "mi_volume_process_fifo_range", "mi_volume_process_fifo_flags", "mi_mixer_process_volume_fifo", "mi_mixer_process_volume_fifo_flags", "mi_volume_get_state_async",
    "mi_equalizer_create", "mi_equalizer_destroy", "mi_equalizer_fir_len", "mi_equalizer_set_gain",
    "mi_equalizer_flatten", "mi_equalizer_set_active", "mi_equalizer_prepare", "mi_equalizer_dump", "mi_equalizer_get_taps",
    "mi_equalizer_set_taps", "mi_equalizer_get_history", "mi_equalizer_set_history", "mi_equalizer_process", "mi_equalizer_process_host",
    "mi_aec_framesize", "mi_aec_create", "mi_aec_destroy", "mi_aec_reset", "mi_aec_process", "mi_aec_process_frames", "mi_aec_process_fifos", "mi_aec_process_fifos_resampled", "mi_aec_process_fifos_masked", "mi_aec_process_fifos_resampled_masked",
    "mi_aec_process_host", "mi_aec_state_bytes", "mi_aec_blob_bytes", "mi_aec_export_state", "mi_aec_import_state", "mi_aec_copy_state", "mi_aec_get", "mi_aec_stagger_info", "mi_aec_stagger_fifos",
    "mi_scaler_create", "mi_scaler_destroy", "mi_scaler_src_bytes", "mi_scaler_dst_bytes",
    "mi_scaler_process", "mi_scaler_process_host", "mi_scaler_process_planes_host",
    "mi_scaler_pipe_create", "mi_scaler_pipe_destroy", "mi_scaler_pipe_acquire", "mi_scaler_pipe_submit", "mi_scaler_pipe_collect", "mi_scaler_pipe_in_flight",
    "mi_pixconv_create", "mi_pixconv_destroy", "mi_pixconv_src_bytes", "mi_pixconv_dst_bytes",
    "mi_pixconv_process", "mi_pixconv_process_host",
    "mi_session_default_config", "mi_session_create", "mi_session_destroy", "mi_session_tick_samples", "mi_session_tick_bytes", "mi_session_events",
    "mi_session_acquire", "mi_session_submit", "mi_session_collect", "mi_session_in_flight",
    "mi_session_set_controls", "mi_session_get_levels", "mi_session_add_member", "mi_session_remove_member", "mi_session_member_count", "mi_session_active_speakers", "mi_session_reset_streams",
    "mi_g711_decode", "mi_g711_encode", "mi_l16_swap", "mi_chan_adapt",
    "mi_flowctl_create", "mi_flowctl_destroy", "mi_flowctl_set_config", "mi_flowctl_request_drop", "mi_flowctl_process",
    "mi_flowctl_get_state", "mi_flowctl_reset",
    "mi_plc_create", "mi_plc_destroy", "mi_plc_reset", "mi_plc_process", "mi_plc_info",
    "mi_fifo_create", "mi_fifo_destroy", "mi_fifo_push", "mi_fifo_push_gated", "mi_fifo_pop", "mi_fifo_pop_frames", "mi_fifo_push_frames", "mi_fifo_levels", "mi_fifo_push_lead", "mi_fifo_phase_of", "mi_fifo_overflows", "mi_fifo_reset", "mi_fifo_reset_range", "mi_fifo_reset_range_at", "mi_fifo_push_silence", "mi_fifo_snapshot", "mi_fifo_export_range", "mi_fifo_import_range",
]


class VolumeParams(C.Structure):
    _fields_ = [
        ("static_gain", C.c_float),
        ("vol_upramp", C.c_float), ("vol_fast_upramp", C.c_float), ("vol_downramp", C.c_float),
        ("ea_thres", C.c_float), ("ea_transmit_thres", C.c_float), ("force", C.c_float),
        ("sustain_time", C.c_int32), ("ng_cut_time", C.c_int32),
        ("ng_threshold", C.c_float), ("ng_floorgain", C.c_float),
        ("agc_enabled", C.c_int32), ("noise_gate_enabled", C.c_int32), ("remove_dc", C.c_int32),
        ("peer", C.c_int32),
    ]


class VolumeState(C.Structure):
    _fields_ = [
        ("energy", C.c_float), ("level_pk", C.c_float), ("instant_energy", C.c_float),
        ("lt_speaker_en", C.c_float), ("gain", C.c_float), ("target_gain", C.c_float),
        ("ng_gain", C.c_float), ("dc_offset", C.c_int32), ("sustain_dur", C.c_int32),
        ("ng_noise_dur", C.c_int32), ("fast_upramp", C.c_int32),
    ]


_lib = None


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 and must be the
    # first to load it, otherwise torch.cuda later reports "No HIP GPUs are available".
    # (Only a load-order fix -- nothing in this package computes with torch.)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_float
    pp = C.POINTER(vp)
    L.mi_abi_version.restype = i32
    L.mi_last_error.restype = C.c_char_p
    L.mi_ctx_create.argtypes = [i32, vp, pp]
    L.mi_ctx_destroy.argtypes = [vp]
    L.mi_ctx_destroy.restype = None
    L.mi_ctx_sync.argtypes = [vp]
    L.mi_warmup.argtypes = [vp]
    L.mi_ctx_stream.argtypes = [vp]
    L.mi_ctx_stream.restype = vp
    L.mi_ctx_device.argtypes = [vp]
    L.mi_ctx_props.argtypes = [vp, C.POINTER(i32), C.POINTER(sz), C.c_char_p, i32]
    L.mi_dev_alloc.argtypes = [vp, sz]
    L.mi_dev_alloc.restype = vp
    L.mi_dev_free.argtypes = [vp, vp]
    L.mi_dev_free.restype = None
    L.mi_host_alloc.argtypes = [vp, sz]
    L.mi_host_alloc.restype = vp
    L.mi_host_free.argtypes = [vp, vp]
    L.mi_host_free.restype = None
    L.mi_copy_h2d.argtypes = [vp, vp, vp, sz]
    L.mi_copy_d2h.argtypes = [vp, vp, vp, sz]
    L.mi_copy_h2d_pinned.argtypes = [vp, vp, vp, sz]
    L.mi_copy_d2h_pinned.argtypes = [vp, vp, vp, sz]
    L.mi_memset.argtypes = [vp, vp, i32, sz]
    L.mi_ctx_capture_begin.argtypes = [vp]
    L.mi_ctx_capture_end.argtypes = [vp, pp]
    L.mi_graph_launch.argtypes = [vp]
    L.mi_graph_destroy.argtypes = [vp]
    L.mi_graph_destroy.restype = None
    L.mi_timer_start.argtypes = [vp]
    L.mi_timer_stop.argtypes = [vp, C.POINTER(f32)]

    L.mi_resampler_create.argtypes = [vp, i32, u32, u32, i32, pp]
    L.mi_resampler_destroy.argtypes = [vp]
    L.mi_resampler_destroy.restype = None
    L.mi_resampler_reset.argtypes = [vp, i32, i32]
    L.mi_resampler_state_bytes.argtypes = [vp]
    L.mi_resampler_get_state.argtypes = [vp, i32, vp, C.c_size_t]
    L.mi_resampler_set_state.argtypes = [vp, i32, vp, C.c_size_t]
    L.mi_resampler_get_states.argtypes = [vp, i32, i32, vp, C.c_size_t]
    L.mi_resampler_set_states.argtypes = [vp, i32, i32, vp, C.c_size_t]
    L.mi_resampler_out_capacity.argtypes = [vp, i32]
    L.mi_resampler_info.argtypes = [vp] + [C.POINTER(i32)] * 4
    L.mi_resampler_get_table.argtypes = [vp, vp, i32]
    L.mi_resampler_process.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    L.mi_resampler_process_host.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    L.mi_resampler_process_masked.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
    L.mi_mixer_process_masked.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.mi_equalizer_process_masked.argtypes = [vp, vp, i32, i32, vp]

    L.mi_mixer_create.argtypes = [vp, i32, i32, i32, pp]
    L.mi_mixer_destroy.argtypes = [vp]
    L.mi_mixer_destroy.restype = None
    L.mi_mixer_set_controls.argtypes = [vp, vp, vp]
    L.mi_mixer_process.argtypes = [vp, vp, vp, i32, vp]
    L.mi_mixer_process_host.argtypes = [vp, vp, vp, i32, vp]
    L.mi_mixer_partial_sum.argtypes = [vp, vp, vp, vp]
    L.mi_mixer_finalize.argtypes = [vp, vp, vp, vp, i32, vp]

    L.mi_volume_create.argtypes = [vp, i32, i32, pp]
    L.mi_volume_destroy.argtypes = [vp]
    L.mi_volume_destroy.restype = None
    L.mi_volume_default_params.argtypes = [C.POINTER(VolumeParams)]
    L.mi_volume_default_params.restype = None
    L.mi_volume_set_params.argtypes = [vp, i32, i32, C.POINTER(VolumeParams)]
    L.mi_volume_get_state.argtypes = [vp, i32, i32, C.POINTER(VolumeState)]
    L.mi_volume_set_state.argtypes = [vp, i32, i32, C.POINTER(VolumeState)]
    L.mi_volume_set_peer_batch.argtypes = [vp, vp]
    L.mi_exchange_unique_id.argtypes = [vp, sz]
    L.mi_exchange_create.argtypes = [vp, i32, i32, vp, pp]
    L.mi_exchange_destroy.argtypes = [vp]
    L.mi_exchange_destroy.restype = None
    L.mi_exchange_ranks.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.mi_exchange_allreduce_i32.argtypes = [vp, vp, sz]
    L.mi_aec_process_fifos_resampled.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, u32, vp]
    L.mi_aec_process_fifos_resampled_masked.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, u32, vp, vp]
    L.mi_aec_copy_state.argtypes = [vp, i32, vp, i32, i32]
    L.mi_aec_stagger_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.mi_aec_stagger_fifos.argtypes = [vp, vp, vp, i32, i32, i32]
    L.mi_fifo_push_lead.argtypes = [vp, i32, i32, i32, i32]
    L.mi_fifo_phase_of.argtypes = [i32, i32]
    L.mi_volume_process_fifo_range.argtypes = [vp, vp, vp, i32, i32, i32, i32]
    L.mi_mixer_process_volume_fifo.argtypes = [vp, vp, i32, vp, vp]
    L.mi_mixer_process_volume_fifo_flags.argtypes = [vp, vp, i32, vp, vp, u32, vp]
    L.mi_scaler_pipe_create.argtypes = [vp, i32, i32, pp]
    L.mi_scaler_pipe_destroy.argtypes = [vp]
    L.mi_scaler_pipe_destroy.restype = None
    L.mi_scaler_pipe_acquire.argtypes = [vp, pp, C.POINTER(sz)]
    L.mi_scaler_pipe_submit.argtypes = [vp, i32]
    L.mi_scaler_pipe_collect.argtypes = [vp, pp, C.POINTER(sz), C.POINTER(i32)]
    L.mi_scaler_pipe_in_flight.argtypes = [vp]
    L.mi_volume_get_state_async.argtypes = [vp, i32, i32, vp]
    L.mi_volume_process_fifo_flags.argtypes = [vp, vp, vp, i32, i32, u32]
    L.mi_volume_get_max.argtypes = [vp, i32, i32, vp]
    L.mi_volume_reset_max.argtypes = [vp, i32, i32]
    L.mi_volume_process.argtypes = [vp, vp, i32, i32, vp]
    L.mi_volume_process_host.argtypes = [vp, vp, i32, i32, vp]
    L.mi_volume_process_fifo.argtypes = [vp, vp, vp, i32, i32]

    if hasattr(L, "mi_equalizer_create"):
        L.mi_equalizer_create.argtypes = [vp, i32, i32, pp]
        L.mi_equalizer_destroy.argtypes = [vp]
        L.mi_equalizer_destroy.restype = None
        L.mi_equalizer_fir_len.argtypes = [vp]
        L.mi_equalizer_set_gain.argtypes = [vp, i32, f32, f32, f32]
        L.mi_equalizer_flatten.argtypes = [vp, i32]
        L.mi_equalizer_set_active.argtypes = [vp, i32, i32]
        L.mi_equalizer_prepare.argtypes = [vp]
        L.mi_equalizer_dump.argtypes = [vp, i32, vp, i32]
        L.mi_equalizer_get_taps.argtypes = [vp, i32, vp, i32]
        L.mi_equalizer_set_taps.argtypes = [vp, i32, vp, i32]
        L.mi_equalizer_get_history.argtypes = [vp, i32, vp, i32]
        L.mi_equalizer_set_history.argtypes = [vp, i32, vp, i32]
        L.mi_equalizer_process.argtypes = [vp, vp, i32, i32]
        L.mi_equalizer_process_host.argtypes = [vp, vp, i32, i32]
    if hasattr(L, "mi_aec_create"):
        L.mi_aec_framesize.argtypes = [i32, i32]
        L.mi_aec_create.argtypes = [vp, i32, i32, i32, i32, pp]
        L.mi_aec_destroy.argtypes = [vp]
        L.mi_aec_destroy.restype = None
        L.mi_aec_reset.argtypes = [vp, i32, i32]
        L.mi_aec_process.argtypes = [vp, vp, vp, vp, i32, vp, C.c_uint]
        L.mi_aec_process_host.argtypes = [vp, vp, vp, vp, i32, vp, C.c_uint]
        L.mi_aec_process_frames.argtypes = [vp, vp, vp, vp, i32, vp, i32, C.c_uint]
        L.mi_aec_process_fifos.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, C.c_uint, vp]
        L.mi_aec_process_fifos_masked.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, C.c_uint, vp, vp]
        L.mi_aec_state_bytes.argtypes = [vp]
        L.mi_aec_state_bytes.restype = sz
        L.mi_aec_blob_bytes.argtypes = [vp]
        L.mi_aec_blob_bytes.restype = sz
        L.mi_aec_export_state.argtypes = [vp, i32, vp, sz]
        L.mi_aec_import_state.argtypes = [vp, i32, vp, sz]
        L.mi_aec_get.argtypes = [vp, i32, C.c_char_p, vp, i32]
    if hasattr(L, "mi_scaler_create"):
        L.mi_scaler_create.argtypes = [vp, i32, i32, i32, i32, i32, pp]
        L.mi_scaler_destroy.argtypes = [vp]
        L.mi_scaler_destroy.restype = None
        L.mi_scaler_src_bytes.argtypes = [vp]
        L.mi_scaler_src_bytes.restype = sz
        L.mi_scaler_dst_bytes.argtypes = [vp]
        L.mi_scaler_dst_bytes.restype = sz
        L.mi_scaler_process.argtypes = [vp, i32, vp, sz, vp, sz]
        L.mi_scaler_process_host.argtypes = [vp, i32, vp, sz, vp, sz]
        L.mi_scaler_process_planes_host.argtypes = [vp, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(i32)]
    if hasattr(L, "mi_session_create"):
        L.mi_session_default_config.argtypes = [vp]
        L.mi_session_default_config.restype = None
        L.mi_session_create.argtypes = [vp, vp, pp]
        L.mi_session_destroy.argtypes = [vp]
        L.mi_session_destroy.restype = None
        L.mi_session_tick_samples.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.mi_session_tick_bytes.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.mi_session_events.argtypes = [vp, pp]
        L.mi_session_acquire.argtypes = [vp, pp, pp]
        L.mi_session_submit.argtypes = [vp]
        L.mi_session_collect.argtypes = [vp, pp]
        L.mi_session_in_flight.argtypes = [vp]
        L.mi_session_set_controls.argtypes = [vp, vp, vp]
        L.mi_session_get_levels.argtypes = [vp, vp]
        L.mi_session_add_member.argtypes = [vp, i32]
        L.mi_session_remove_member.argtypes = [vp, i32]
        L.mi_session_member_count.argtypes = [vp, i32]
        L.mi_session_active_speakers.argtypes = [vp, C.c_uint64, vp, vp]
        L.mi_session_reset_streams.argtypes = [vp, i32, i32]
    if hasattr(L, "mi_fifo_create"):
        L.mi_fifo_create.argtypes = [vp, i32, i32, pp]
        L.mi_fifo_destroy.argtypes = [vp]
        L.mi_fifo_destroy.restype = None
        L.mi_fifo_push.argtypes = [vp, vp, i32, i32, vp]
        L.mi_fifo_push_gated.argtypes = [vp, vp, i32, i32, vp]
        L.mi_fifo_pop.argtypes = [vp, i32, vp, i32, vp, vp, i32]
        L.mi_fifo_pop_frames.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32]
        L.mi_fifo_push_frames.argtypes = [vp, vp, i32, i32, i32, vp]
        L.mi_fifo_levels.argtypes = [vp, vp]
        L.mi_fifo_overflows.argtypes = [vp, C.POINTER(i32)]
        L.mi_fifo_reset.argtypes = [vp]
        L.mi_fifo_reset_range.argtypes = [vp, i32, i32]
        L.mi_fifo_reset_range_at.argtypes = [vp, i32, i32, i32]
        L.mi_fifo_push_silence.argtypes = [vp, vp]
        L.mi_fifo_snapshot.argtypes = [vp, vp, vp, vp]
        L.mi_fifo_export_range.argtypes = [vp, i32, i32, vp, i32, vp]
        L.mi_fifo_import_range.argtypes = [vp, i32, i32, vp, i32, vp, i32]
    if hasattr(L, "mi_g711_decode"):
        L.mi_g711_decode.argtypes = [vp, i32, vp, sz, vp, sz, vp, i32, sz]
        L.mi_g711_encode.argtypes = [vp, i32, vp, sz, vp, sz, vp, i32, sz]
        L.mi_l16_swap.argtypes = [vp, vp, vp, sz]
        L.mi_chan_adapt.argtypes = [vp, i32, vp, vp, vp, sz]
        L.mi_flowctl_create.argtypes = [vp, i32, i32, pp]
        L.mi_flowctl_destroy.argtypes = [vp]
        L.mi_flowctl_destroy.restype = None
        L.mi_flowctl_set_config.argtypes = [vp, i32, i32, i32, C.c_float]
        L.mi_flowctl_request_drop.argtypes = [vp, vp, vp]
        L.mi_flowctl_process.argtypes = [vp, vp, sz, vp, i32, vp, sz, vp]
        L.mi_flowctl_get_state.argtypes = [vp, i32, C.POINTER(C.c_uint32)]
        L.mi_flowctl_reset.argtypes = [vp, i32, i32]
    if hasattr(L, "mi_plc_create"):
        L.mi_plc_create.argtypes = [vp, i32, i32, i32, pp]
        L.mi_plc_destroy.argtypes = [vp]
        L.mi_plc_destroy.restype = None
        L.mi_plc_reset.argtypes = [vp, i32, i32]
        L.mi_plc_process.argtypes = [vp, vp, sz, vp, vp]
        L.mi_plc_info.argtypes = [vp, i32, C.POINTER(i32)]
    if hasattr(L, "mi_pixconv_create"):
        L.mi_pixconv_create.argtypes = [vp, i32, i32, i32, i32, pp]
        L.mi_pixconv_destroy.argtypes = [vp]
        L.mi_pixconv_destroy.restype = None
        L.mi_pixconv_src_bytes.argtypes = [vp]
        L.mi_pixconv_src_bytes.restype = sz
        L.mi_pixconv_dst_bytes.argtypes = [vp]
        L.mi_pixconv_dst_bytes.restype = sz
        L.mi_pixconv_process.argtypes = [vp, i32, vp, sz, vp, sz]
        L.mi_pixconv_process_host.argtypes = [vp, i32, vp, sz, vp, sz]
    _lib = L
    return L


def check(rc):
    if rc != MI_OK:
        raise MiError(rc, load().mi_last_error().decode(errors="replace"))
    return rc
