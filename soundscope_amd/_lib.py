"""ctypes loader for the C-ABI library (include/soundscope_hip.h).

The library is built in-tree by soundscope_amd/csrc/Makefile (hipcc, gfx950).
There is no fallback of any kind: a missing library raises ImportError at load,
and a missing GPU makes every compute entry point return SS_ERR_DEVICE, which
the wrappers turn into DeviceError.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOUNDSCOPE_HIP_LIB: load another build of the same library (A/B runs of kernel variants, tools/ab_libs.sh)
LIB_PATH = os.environ.get("SOUNDSCOPE_HIP_LIB") or os.path.join(_HERE, "lib", "libsoundscope_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/soundscope_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ss_status_string", "ss_abi_version", "ss_device_count", "ss_set_device", "ss_last_device_error",
    "ss_analyzer_create", "ss_analyzer_destroy", "ss_analyzer_configure", "ss_get_fft", "ss_get_fft_error_values", "ss_get_waveform",
    "ss_add_samples", "ss_reset", "ss_get_shortterm_lufs", "ss_get_integrated_lufs", "ss_get_loudness_range",
    "ss_get_true_peak", "ss_sample_rate", "ss_calculate_integrated_lufs", "ss_mid_side",
    "ss_get_momentary_lufs", "ss_get_true_peak_channel", "ss_get_sample_peak_channel",
    "ss_analyzer_set_true_peak_factor", "ss_analyzer_set_true_peak_arith",
    "ss_batch_create", "ss_batch_destroy", "ss_batch_layout_get", "ss_batch_upload", "ss_batch_download_input",
    "ss_batch_input_device_ptr", "ss_batch_synthesize", "ss_batch_run", "ss_batch_sync", "ss_batch_results",
    "ss_batch_download_fft", "ss_batch_bin_tables", "ss_batch_download_waveform", "ss_batch_download_subblocks",
    "ss_batch_histograms", "ss_batch_histograms_device", "ss_corpus_integrated_lufs", "ss_corpus_loudness_range",
    "ss_batch_timing_enable", "ss_batch_timing_read", "ss_kernel_name",
    "ss_wav_parse", "ss_pcm_sample_bytes", "ss_pcm_decode", "ss_batch_upload_pcm",
    "ss_session_open_file", "ss_session_open_capture", "ss_session_close", "ss_session_analyzer",
    "ss_session_waveform", "ss_session_gain_db", "ss_session_duration_ms", "ss_session_tick_file",
    "ss_session_tick_capture", "ss_session_capture_push", "ss_session_tick_capture_resident", "ss_session_restart",
    "ss_session_lufs_history",
    "ss_batch_render_spectrum", "ss_batch_download_spectrum_columns", "ss_batch_render_waveform",
    "ss_batch_download_waveform_columns", "ss_waveform_view", "ss_batch_kernel_name",
    "ss_host_register", "ss_host_unregister", "ss_batch_upload_pcm_async",
    "ss_batch_set_lengths", "ss_batch_stream_shape", "ss_batch_upload_samples",
    "ss_device_synchronize", "ss_batch_peaks", "ss_batch_geometry_get", "ss_batch_set_overlap",
    "ss_comm_init", "ss_comm_init_on_device", "ss_comm_init_from_env", "ss_comm_destroy", "ss_comm_rank", "ss_comm_size",
    "ss_comm_transport_name", "ss_comm_library_version", "ss_comm_barrier", "ss_comm_allreduce_u64_sum", "ss_comm_allreduce_f64_max",
    "ss_batch_allreduce_histograms", "ss_batch_traffic_floor",
    "ss_batch_corpus_gate_enqueue", "ss_batch_corpus_gate_read", "ss_batch_checksums", "ss_inspect_filter_state", "ss_batch_set_true_peak_arith", "ss_batch_get_true_peak_arith", "ss_batch_set_time_domain_mode", "ss_batch_set_columns_gain",
    "ss_release_caches", "ss_batch_geometry_get_sized",
    "ss_inspect_kweight", "ss_inspect_true_peak", "ss_inspect_hann", "ss_inspect_bins", "ss_inspect_histogram",
]

SS_ABI_VERSION = 2          # include/soundscope_hip.h; checked at load
SS_OK = 0
SS_ERR_NOMEM, SS_ERR_INVALID_MODE, SS_ERR_INVALID_CHANNEL = 1, 2, 3
SS_ERR_TOO_FEW_SAMPLES, SS_ERR_NAN, SS_ERR_INFINITY, SS_ERR_NOT_POW2, SS_ERR_FREQ_LIMIT, SS_ERR_SCALING = 10, 11, 12, 13, 14, 15
SS_ERR_CAPACITY, SS_ERR_UNSUPPORTED, SS_ERR_INVALID_ARG, SS_ERR_DEVICE = 20, 21, 22, 30

SS_BATCH_FFT, SS_BATCH_LUFS, SS_BATCH_TRUE_PEAK, SS_BATCH_WAVEFORM, SS_BATCH_ALL = 1, 2, 4, 8, 15
SS_BATCH_FFT_COLUMNS = 16
SS_PCM_U8, SS_PCM_S16, SS_PCM_S24, SS_PCM_S32, SS_PCM_F32, SS_PCM_F64 = 1, 2, 3, 4, 5, 6
SS_GAIN_FIXED, SS_GAIN_REFERENCE = 0, 1
SS_COMM_RCCL, SS_COMM_HOST_TCP = 0, 1
SS_TP_ARITH_F16X3, SS_TP_ARITH_F32 = 0, 1
SS_TD_AUTO, SS_TD_RUN_IN, SS_TD_WHOLE_STREAMS = 0, 1, 2
SS_KERNEL_FFT, SS_KERNEL_TIME_DOMAIN, SS_KERNEL_FINALIZE, SS_KERNEL_WAVEFORM, SS_KERNEL_COUNT = 0, 1, 2, 3, 4


class BatchConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("n_streams", C.c_uint32),
                ("fft_n", C.c_uint32), ("hop_frames", C.c_uint32), ("flags", C.c_uint32),
                ("true_peak_factor", C.c_int32), ("spectrum_columns", C.c_uint32),
                ("frames_per_stream", C.c_uint64), ("waveform_window", C.c_double)]


class StreamResult(C.Structure):
    _fields_ = [("integrated_lufs", C.c_double), ("loudness_range", C.c_double),
                ("true_peak", C.c_double * 2), ("sample_peak", C.c_double * 2),
                ("n_gating_blocks", C.c_uint32), ("n_st_blocks", C.c_uint32)]


class WavInfo(C.Structure):
    _fields_ = [("format", C.c_uint32), ("channels", C.c_uint32), ("sample_rate", C.c_uint32),
                ("bits_per_sample", C.c_uint32), ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64),
                ("frames", C.c_uint64)]


class TickResult(C.Structure):
    _fields_ = [("playhead", C.c_uint64), ("fft_ran", C.c_int32), ("mid_status", C.c_int32),
                ("side_status", C.c_int32), ("n_mid", C.c_uint32), ("n_side", C.c_uint32),
                ("lufs_ran", C.c_int32), ("fed", C.c_int32), ("add_status", C.c_int32),
                ("shortterm_status", C.c_int32), ("reserved", C.c_uint32), ("shortterm", C.c_double)]


class StreamShape(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("n_windows", C.c_uint32), ("n_subblocks", C.c_uint32),
                ("n_wave_points", C.c_uint32), ("reserved", C.c_uint32)]


class BatchGeometry(C.Structure):
    _fields_ = [("fft_windows_per_block", C.c_uint32), ("fft_blocks", C.c_uint32), ("td_segments", C.c_uint32),
                ("td_segment_subblocks", C.c_uint32), ("td_warm_subblocks", C.c_uint32),
                ("td_true_peak_factor", C.c_uint32), ("waveform_fused", C.c_uint32), ("overlap", C.c_uint32),
                ("td_split", C.c_uint32), ("td_fixup_subblocks", C.c_uint32)]


class BatchLayout(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("fft_channels", C.c_uint32), ("n_bins", C.c_uint32),
                ("first_bin", C.c_uint32), ("n_wave_points", C.c_uint32), ("n_subblocks", C.c_uint32),
                ("fft_bin_stride", C.c_uint32), ("reserved", C.c_uint32),
                ("input_bytes", C.c_uint64), ("fft_bytes", C.c_uint64)]


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _bind(lib):
    vp, f32p, f64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)
    szp, u64p = C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)
    sig = {
        "ss_status_string": (C.c_char_p, [C.c_int]),
        "ss_abi_version": (C.c_int, []),
        "ss_device_count": (C.c_int, []),
        "ss_set_device": (C.c_int, [C.c_int]),
        "ss_last_device_error": (C.c_char_p, []),
        "ss_analyzer_create": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "ss_analyzer_destroy": (None, [vp]),
        "ss_analyzer_configure": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "ss_get_fft": (C.c_int, [vp, f32p, C.c_size_t, f64p, C.c_size_t, szp]),
        "ss_get_fft_error_values": (C.c_int, [vp, f32p, f32p]),
        "ss_get_waveform": (C.c_int, [f32p, C.c_size_t, C.c_double, f64p, C.c_size_t, szp]),
        "ss_add_samples": (C.c_int, [vp, f32p, C.c_size_t]),
        "ss_reset": (None, [vp]),
        "ss_get_shortterm_lufs": (C.c_int, [vp, f64p]),
        "ss_get_integrated_lufs": (C.c_int, [vp, f64p]),
        "ss_get_loudness_range": (C.c_int, [vp, f64p]),
        "ss_get_momentary_lufs": (C.c_int, [vp, f64p]),
        "ss_get_true_peak": (C.c_int, [vp, f64p, f64p]),
        "ss_get_true_peak_channel": (C.c_int, [vp, C.c_uint32, f64p]),
        "ss_get_sample_peak_channel": (C.c_int, [vp, C.c_uint32, f64p]),
        "ss_analyzer_set_true_peak_factor": (C.c_int, [vp, C.c_int]),
        "ss_sample_rate": (C.c_uint32, [vp]),
        "ss_calculate_integrated_lufs": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t, f64p]),
        "ss_mid_side": (C.c_int, [f32p, C.c_size_t, f32p, f32p, szp]),
        "ss_batch_create": (C.c_int, [C.POINTER(BatchConfig), C.POINTER(vp)]),
        "ss_batch_destroy": (None, [vp]),
        "ss_batch_layout_get": (C.c_int, [vp, C.POINTER(BatchLayout)]),
        "ss_batch_upload": (C.c_int, [vp, C.c_uint32, C.c_uint32, f32p]),
        "ss_batch_download_input": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t]),
        "ss_batch_input_device_ptr": (vp, [vp]),
        "ss_batch_synthesize": (C.c_int, [vp, C.c_uint64, C.c_uint32]),
        "ss_batch_run": (C.c_int, [vp]),
        "ss_batch_sync": (C.c_int, [vp]),
        "ss_batch_results": (C.c_int, [vp, C.POINTER(StreamResult), C.c_uint32]),
        "ss_batch_download_fft": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t]),
        "ss_batch_bin_tables": (C.c_int, [vp, f64p, f64p, f64p]),
        "ss_batch_download_waveform": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t]),
        "ss_batch_download_subblocks": (C.c_int, [vp, C.c_uint32, f64p, C.c_size_t]),
        "ss_batch_histograms": (C.c_int, [vp, u64p]),
        "ss_batch_histograms_device": (C.c_int, [vp, vp]),
        "ss_corpus_integrated_lufs": (C.c_double, [u64p]),
        "ss_corpus_loudness_range": (C.c_double, [u64p]),
        "ss_batch_timing_enable": (C.c_int, [vp, C.c_int]),
        "ss_batch_timing_read": (C.c_int, [vp, C.c_int, f64p, u64p]),
        "ss_kernel_name": (C.c_char_p, [C.c_int]),
        "ss_wav_parse": (C.c_int, [vp, C.c_size_t, C.POINTER(WavInfo)]),
        "ss_pcm_sample_bytes": (C.c_size_t, [C.c_int]),
        "ss_pcm_decode": (C.c_int, [vp, C.c_size_t, C.c_int, f32p]),
        "ss_batch_upload_pcm": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_int]),
        "ss_session_open_file": (C.c_int, [f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "ss_session_open_capture": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "ss_session_close": (None, [vp]),
        "ss_session_analyzer": (vp, [vp]),
        "ss_session_waveform": (C.c_int, [vp, f64p, C.c_size_t, szp]),
        "ss_session_gain_db": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "ss_session_duration_ms": (C.c_int, [vp, u64p]),
        "ss_session_tick_file": (C.c_int, [vp, C.c_size_t, f64p, f64p, C.c_size_t, C.POINTER(TickResult)]),
        "ss_session_tick_capture": (C.c_int, [vp, f32p, C.c_size_t, f64p, f64p, C.c_size_t, f64p, C.c_size_t,
                                              szp, C.POINTER(TickResult)]),
        "ss_session_capture_push": (C.c_int, [vp, f32p, C.c_size_t]),
        "ss_session_tick_capture_resident": (C.c_int, [vp, f64p, f64p, C.c_size_t, f64p, C.c_size_t, szp, C.POINTER(TickResult)]),
        "ss_session_restart": (C.c_int, [vp]),
        "ss_session_lufs_history": (C.c_int, [vp, f64p]),
        "ss_batch_render_spectrum": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_float]),
        "ss_batch_download_spectrum_columns": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t]),
        "ss_batch_render_waveform": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ss_batch_download_waveform_columns": (C.c_int, [vp, C.c_uint32, f32p, C.c_size_t]),
        "ss_waveform_view": (None, [C.c_double, C.c_double, C.c_size_t, f64p, f64p]),
        "ss_batch_kernel_name": (C.c_char_p, [vp, C.c_int]),
        "ss_host_register": (C.c_int, [vp, C.c_size_t]),
        "ss_host_unregister": (C.c_int, [vp]),
        "ss_batch_upload_pcm_async": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_int]),
        "ss_batch_set_lengths": (C.c_int, [vp, u64p, C.c_uint32]),
        "ss_batch_upload_samples": (C.c_int, [vp, C.c_uint32, vp, C.c_size_t, C.c_int]),
        "ss_batch_stream_shape": (C.c_int, [vp, C.c_uint32, C.POINTER(StreamShape)]),
        "ss_device_synchronize": (C.c_int, []),
        "ss_batch_peaks": (C.c_int, [vp, C.c_uint32, f64p, f64p, C.c_uint32]),
        "ss_batch_geometry_get": (C.c_int, [vp, C.POINTER(BatchGeometry)]),
        "ss_batch_set_overlap": (C.c_int, [vp, C.c_int]),
        "ss_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
        "ss_comm_init_on_device": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
        "ss_comm_init_from_env": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "ss_comm_destroy": (None, [vp]),
        "ss_comm_rank": (C.c_int, [vp]),
        "ss_comm_size": (C.c_int, [vp]),
        "ss_comm_transport_name": (C.c_char_p, [vp]),
        "ss_comm_library_version": (C.c_int, [vp]),
        "ss_comm_barrier": (C.c_int, [vp]),
        "ss_comm_allreduce_u64_sum": (C.c_int, [vp, u64p, C.c_size_t]),
        "ss_comm_allreduce_f64_max": (C.c_int, [vp, f64p, C.c_size_t]),
        "ss_batch_allreduce_histograms": (C.c_int, [vp, vp, u64p]),
        "ss_batch_traffic_floor": (C.c_int, [vp, C.c_uint32, f64p]),
        "ss_batch_corpus_gate_enqueue": (C.c_int, [vp, vp]),
        "ss_batch_corpus_gate_read": (C.c_int, [vp, f64p, f64p]),
        "ss_batch_checksums": (C.c_int, [vp, u64p, C.c_uint32]),
        "ss_inspect_filter_state": (C.c_int, [vp, C.c_uint32, f64p]),
        "ss_batch_set_true_peak_arith": (C.c_int, [vp, C.c_int]),
        "ss_analyzer_set_true_peak_arith": (C.c_int, [vp, C.c_int]),
        "ss_batch_get_true_peak_arith": (C.c_int, [vp]),
        "ss_batch_set_time_domain_mode": (C.c_int, [vp, C.c_int]),
        "ss_batch_set_columns_gain": (C.c_int, [vp, C.c_int, C.c_float]),
        "ss_release_caches": (C.c_int, []),
        "ss_batch_geometry_get_sized": (C.c_int, [vp, vp, C.c_size_t]),
        "ss_inspect_kweight": (C.c_int, [C.c_uint32, f64p, f64p]),
        "ss_inspect_true_peak": (C.c_int, [C.c_int, f32p, C.c_uint32, C.POINTER(C.c_uint32)]),
        "ss_inspect_hann": (C.c_int, [C.c_uint32, f32p]),
        "ss_inspect_bins": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "ss_inspect_histogram": (C.c_int, [f64p, f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib():
    """The loaded library.  Raises ImportError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            try:                      # build on demand (hipcc cross-compiles gfx950 without a GPU)
                build()
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C soundscope_amd/csrc`). "
                "soundscope_amd has no CPU fallback.")
        cand = _bind(C.CDLL(LIB_PATH))
        got = cand.ss_abi_version()
        if got != SS_ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {SS_ABI_VERSION}: rebuild "
                              "(make -C soundscope_amd/csrc)")
        _LIB = cand
    return _LIB
