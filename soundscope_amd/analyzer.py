"""Host-side mirror of soundscope's `Analyzer` (reference src/analyzer.rs:29-183).

Same method names, argument meaning and error behaviour as the Rust type, on
top of the C ABI in include/soundscope_hip.h.  Rust `Result::Err` becomes an
exception carrying the status code; `Option::None` becomes `None`.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class AnalyzerError(Exception):
    """An `Err(..)` of the reference API; `.code` is the ss_status."""

    def __init__(self, code):
        self.code = int(code)
        msg = L.lib().ss_status_string(self.code).decode()
        if self.code == L.SS_ERR_DEVICE:
            msg += ": " + L.lib().ss_last_device_error().decode()
        super().__init__(msg)


class DeviceError(AnalyzerError):
    """No HIP device / HIP runtime failure.  There is no CPU fallback."""


def _check(rc):
    if rc == L.SS_OK:
        return
    if rc == L.SS_ERR_DEVICE:
        raise DeviceError(rc)
    raise AnalyzerError(rc)


def _f32(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def get_mid_and_side_samples(samples):
    """audio_player.rs:400-419 — (mid, side) of an interleaved stereo buffer."""
    a, ap = _f32(samples)
    frames = a.size // 2
    mid = np.empty(frames, np.float32)
    side = np.empty(frames, np.float32)
    n = C.c_size_t(0)
    _check(L.lib().ss_mid_side(ap, a.size, mid.ctypes.data_as(C.POINTER(C.c_float)),
                               side.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
    return mid, side


class Analyzer:
    """`pub struct Analyzer` — default is 2 channels at 44 100 Hz (analyzer.rs:34-45)."""

    def __init__(self, channels: int = 2, rate: int = 44100):
        self._h = C.c_void_p()
        rc = L.lib().ss_analyzer_create(channels, rate, C.byref(self._h))
        if rc != L.SS_OK:
            self._h = None
            _check(rc)          # Analyzer::default() panics; here it raises

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ss_analyzer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- analyzer.rs:49-53
    def create_loudness_meter(self, channels: int, rate: int) -> None:
        _check(L.lib().ss_analyzer_configure(self._h, channels, rate))

    # -- analyzer.rs:55-105
    def get_fft(self, samples) -> np.ndarray:
        """-> array [nbins, 2] of (chart_x, dB) like Vec<(f64, f64)>."""
        a, ap = _f32(samples)
        cap = a.size // 2 + 1
        out = np.empty((max(cap, 1), 2), np.float64)
        n = C.c_size_t(0)
        rc = L.lib().ss_get_fft(self._h, ap, a.size, out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n))
        if rc in (L.SS_ERR_SCALING, L.SS_ERR_FREQ_LIMIT):
            # the payload of SpectrumAnalyzerError::ScalingError(orig, scaled) / InvalidFrequencyLimit(ValueAboveNyquist(limit))
            va, vb = C.c_float(0), C.c_float(0)
            L.lib().ss_get_fft_error_values(self._h, C.byref(va), C.byref(vb))
            e = AnalyzerError(rc)
            e.values = (va.value, vb.value)
            raise e
        _check(rc)
        return out[:n.value].copy()

    # -- analyzer.rs:107-137 (associated function)
    @staticmethod
    def get_waveform(samples, waveform_window: float) -> np.ndarray:
        a, ap = _f32(samples)
        wd = waveform_window * 1000.0
        w = int(wd) if wd == wd and wd > 0 else 0
        cap = 2 * w + 2
        out = np.empty((cap, 2), np.float64)
        n = C.c_size_t(0)
        _check(L.lib().ss_get_waveform(ap, a.size, float(waveform_window),
                                       out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)))
        return out[:n.value].copy()

    # -- analyzer.rs:139-141
    def add_samples(self, samples) -> None:
        a, ap = _f32(samples)
        _check(L.lib().ss_add_samples(self._h, ap, a.size))

    # -- analyzer.rs:143-145
    def reset(self) -> None:
        L.lib().ss_reset(self._h)

    def _scalar(self, fn):
        v = C.c_double()
        _check(fn(self._h, C.byref(v)))
        return v.value

    # -- analyzer.rs:147-157
    def get_shortterm_lufs(self) -> float:
        return self._scalar(L.lib().ss_get_shortterm_lufs)

    def get_integrated_lufs(self) -> float:
        return self._scalar(L.lib().ss_get_integrated_lufs)

    def get_loudness_range(self) -> float:
        return self._scalar(L.lib().ss_get_loudness_range)

    def get_momentary_lufs(self) -> float:
        return self._scalar(L.lib().ss_get_momentary_lufs)

    # -- analyzer.rs:159-164
    def get_true_peak(self):
        l, r = C.c_double(), C.c_double()
        _check(L.lib().ss_get_true_peak(self._h, C.byref(l), C.byref(r)))
        return l.value, r.value

    def get_true_peak_channel(self, ch: int) -> float:
        v = C.c_double()
        _check(L.lib().ss_get_true_peak_channel(self._h, ch, C.byref(v)))
        return v.value

    def get_sample_peak_channel(self, ch: int) -> float:
        v = C.c_double()
        _check(L.lib().ss_get_sample_peak_channel(self._h, ch, C.byref(v)))
        return v.value

    def filter_state(self, ch: int) -> np.ndarray:
        """carried DF-II state v1..v4 of the K-weighting filter of channel `ch` (inspection; ebur128 Filter state)"""
        v = np.zeros(4, np.float64)
        _check(L.lib().ss_inspect_filter_state(self._h, ch, v.ctypes.data_as(C.POINTER(C.c_double))))
        return v

    def set_true_peak_factor(self, factor: int) -> None:
        _check(L.lib().ss_analyzer_set_true_peak_factor(self._h, factor))

    def set_true_peak_arith(self, arith: int) -> None:
        """L.SS_TP_ARITH_F32 (default, the width of ebur128's interpolator) or L.SS_TP_ARITH_F16X3 (opt-in f16x3 split)."""
        _check(L.lib().ss_analyzer_set_true_peak_arith(self._h, arith))

    # -- analyzer.rs:166-168
    def sample_rate(self) -> int:
        return L.lib().ss_sample_rate(self._h)

    # -- analyzer.rs:170-182
    def calculate_integrated_lufs(self, channels: int, samples):
        a, ap = _f32(samples)
        v = C.c_double()
        rc = L.lib().ss_calculate_integrated_lufs(self._h, channels, ap, a.size, C.byref(v))
        if rc == L.SS_ERR_DEVICE:
            raise DeviceError(rc)
        return v.value if rc == L.SS_OK else None
