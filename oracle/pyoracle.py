"""ctypes binding of oracle/libss_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/ss_oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(native: bool = False) -> str:
    target = "libss_oracle_native.so" if native else "libss_oracle.so"
    subprocess.run(["make", "-C", _HERE, "native" if native else "all"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, target)


def _load(native: bool = False):
    path = os.path.join(_HERE, "libss_oracle_native.so" if native else "libss_oracle.so")
    if not os.path.exists(path):
        path = build(native)
    lib = C.CDLL(path)
    f32p, f64p, szp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_size_t)
    lib.so_hann_window.argtypes = [f32p, C.c_size_t, f32p]
    lib.so_rfft.argtypes = [f32p, C.c_size_t, f32p, f32p]
    lib.so_fft_bins.argtypes = [C.c_uint32, C.c_size_t, szp]
    lib.so_fft_bins.restype = C.c_size_t
    lib.so_get_fft_ex.argtypes = [C.c_uint32, f32p, C.c_size_t, f64p, f32p, C.c_size_t, szp, C.c_int]
    lib.so_get_waveform.argtypes = [f32p, C.c_size_t, C.c_double, f64p, C.c_size_t]
    lib.so_get_waveform.restype = C.c_size_t
    lib.so_pcm_to_f32.argtypes = [C.c_char_p, C.c_size_t, C.c_int, f32p]
    lib.so_mid_side.argtypes = [f32p, C.c_size_t, f32p, f32p]
    lib.so_mid_side.restype = C.c_size_t
    vp = C.c_void_p
    lib.so_meter_new_ex.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.so_meter_free.argtypes = [vp]
    lib.so_meter_reset.argtypes = [vp]
    lib.so_meter_set_ftz.argtypes = [vp, C.c_int]
    lib.so_meter_add_frames_f32.argtypes = [vp, f32p, C.c_size_t]
    for n in ("momentary", "shortterm", "global", "range"):
        getattr(lib, "so_meter_loudness_" + n).argtypes = [vp, f64p]
    lib.so_meter_sample_peak.argtypes = [vp, C.c_uint32, f64p]
    lib.so_meter_true_peak.argtypes = [vp, C.c_uint32, f64p]
    lib.so_meter_block_hist.argtypes = [vp]
    lib.so_meter_block_hist.restype = C.POINTER(C.c_uint64)
    lib.so_meter_st_hist.argtypes = [vp]
    lib.so_meter_st_hist.restype = C.POINTER(C.c_uint64)
    lib.so_meter_filter_coeffs.argtypes = [vp, f64p, f64p]
    lib.so_meter_filter_state.argtypes = [vp, C.c_uint32, f64p]
    lib.so_analyze_streams_mt.argtypes = [C.c_uint32, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                          C.c_int, C.c_int, f64p]
    lib.so_gated_loudness_hist.argtypes = [C.POINTER(C.c_uint64)]
    lib.so_gated_loudness_hist.restype = C.c_double
    lib.so_loudness_range_hist.argtypes = [C.POINTER(C.c_uint64)]
    lib.so_loudness_range_hist.restype = C.c_double
    lib.so_interp_layout.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.so_interp_coeffs.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.POINTER(C.c_int), C.c_size_t]
    lib.so_interp_coeffs.restype = C.c_size_t
    lib.so_calculate_integrated_lufs.argtypes = [C.c_uint32, C.c_uint32, f32p, C.c_size_t, f64p]
    lib.so_analyze_stream.argtypes = [C.c_uint32, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                      f32p, f64p, C.c_void_p]
    return lib


_LIB = None
_LIB_NATIVE = None


def lib(native: bool = False):
    global _LIB, _LIB_NATIVE
    if native:
        if _LIB_NATIVE is None:
            _LIB_NATIVE = _load(True)
        return _LIB_NATIVE
    if _LIB is None:
        _LIB = _load(False)
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle status {code}")
        self.code = code


def hann_window(x):
    x, xp = _f32(x)
    out = np.empty_like(x)
    lib().so_hann_window(xp, x.size, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def rfft(x):
    x, xp = _f32(x)
    re = np.empty(x.size // 2 + 1, np.float32)
    im = np.empty(x.size // 2 + 1, np.float32)
    lib().so_rfft(xp, x.size, re.ctypes.data_as(C.POINTER(C.c_float)), im.ctypes.data_as(C.POINTER(C.c_float)))
    return re + 1j * im.astype(np.complex64)


def fft_bins(sample_rate, n):
    first = C.c_size_t(0)
    cnt = lib().so_fft_bins(sample_rate, n, C.byref(first))
    return cnt, first.value


def get_fft(sample_rate, x, with_dbfs=False):
    """Analyzer::get_fft -> array [nbins, 2] of (chart_x, dB) f64."""
    x, xp = _f32(x)
    cap = x.size // 2 + 1 if x.size else 1
    out = np.empty((cap, 2), np.float64)
    dbfs = np.empty(cap, np.float32)
    n = C.c_size_t(0)
    rc = lib().so_get_fft_ex(sample_rate, xp, x.size, out.ctypes.data_as(C.POINTER(C.c_double)),
                             dbfs.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), 0)
    if rc:
        raise OracleError(rc)
    if with_dbfs:
        return out[:n.value].copy(), dbfs[:n.value].copy()
    return out[:n.value].copy()


def get_waveform(x, window_s):
    x, xp = _f32(x)
    w = int(window_s * 1000.0) if window_s > 0 else 0
    out = np.empty((2 * w + 2, 2), np.float64)
    n = lib().so_get_waveform(xp, x.size, float(window_s), out.ctypes.data_as(C.POINTER(C.c_double)), 2 * w + 2)
    return out[:n].copy()


def pcm_to_f32(raw: bytes, fmt: int):
    sb = {1: 1, 2: 2, 3: 3, 4: 4, 5: 4, 6: 8}[fmt]
    n = len(raw) // sb
    out = np.empty(n, np.float32)
    lib().so_pcm_to_f32(raw, n, fmt, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def mid_side(x):
    x, xp = _f32(x)
    f = x.size // 2
    mid = np.empty(f, np.float32)
    side = np.empty(f, np.float32)
    lib().so_mid_side(xp, x.size, mid.ctypes.data_as(C.POINTER(C.c_float)), side.ctypes.data_as(C.POINTER(C.c_float)))
    return mid, side


class Meter:
    """ebur128::EbuR128 with Mode::all()."""

    def __init__(self, channels, rate, force_tp_factor=0):
        self._h = C.c_void_p()
        rc = lib().so_meter_new_ex(channels, rate, force_tp_factor, C.byref(self._h))
        if rc:
            raise OracleError(rc)
        self.channels, self.rate = channels, rate

    def __del__(self):
        if getattr(self, "_h", None):
            lib().so_meter_free(self._h)
            self._h = None

    def reset(self):
        lib().so_meter_reset(self._h)

    def set_ftz(self, per_op):
        """False: sub-normal filter state flushed at the end of every internal filter call (default; the crate without SSE2);
        True: MXCSR flush-to-zero during the call (the crate's x86 build)."""
        rc = lib().so_meter_set_ftz(self._h, 1 if per_op else 0)
        if rc:
            raise OracleError(rc)

    def add_frames(self, x):
        x, xp = _f32(x)
        rc = lib().so_meter_add_frames_f32(self._h, xp, x.size)
        if rc:
            raise OracleError(rc)

    def _get(self, name):
        v = C.c_double()
        rc = getattr(lib(), "so_meter_loudness_" + name)(self._h, C.byref(v))
        if rc:
            raise OracleError(rc)
        return v.value

    def momentary(self):
        return self._get("momentary")

    def shortterm(self):
        return self._get("shortterm")

    def integrated(self):
        return self._get("global")

    def loudness_range(self):
        return self._get("range")

    def sample_peak(self, ch):
        v = C.c_double()
        rc = lib().so_meter_sample_peak(self._h, ch, C.byref(v))
        if rc:
            raise OracleError(rc)
        return v.value

    def true_peak(self, ch):
        v = C.c_double()
        rc = lib().so_meter_true_peak(self._h, ch, C.byref(v))
        if rc:
            raise OracleError(rc)
        return v.value

    def block_hist(self):
        return np.ctypeslib.as_array(lib().so_meter_block_hist(self._h), (1000,)).copy()

    def st_hist(self):
        return np.ctypeslib.as_array(lib().so_meter_st_hist(self._h), (1000,)).copy()

    def filter_state(self, ch):
        """carried DF-II state v1..v4 of channel `ch` after the last add_frames"""
        v = (C.c_double * 4)()
        rc = lib().so_meter_filter_state(self._h, ch, v)
        if rc:
            raise OracleError(rc)
        return np.array(v)

    def coeffs(self):
        b = (C.c_double * 5)()
        a = (C.c_double * 5)()
        lib().so_meter_filter_coeffs(self._h, b, a)
        return np.array(b), np.array(a)


def gated_loudness_hist(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    return lib().so_gated_loudness_hist(h.ctypes.data_as(C.POINTER(C.c_uint64)))


def loudness_range_hist(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    return lib().so_loudness_range_hist(h.ctypes.data_as(C.POINTER(C.c_uint64)))


def interp_layout(taps, factor):
    counts = (C.c_int * factor)()
    delay = C.c_int()
    lib().so_interp_layout(taps, factor, counts, C.byref(delay))
    return list(counts), delay.value


def interp_coeffs(taps, factor, phase):
    co = np.empty(64, np.float32)
    ix = np.empty(64, np.int32)
    n = lib().so_interp_coeffs(taps, factor, phase, co.ctypes.data_as(C.POINTER(C.c_float)),
                               ix.ctypes.data_as(C.POINTER(C.c_int)), 64)
    return co[:n].copy(), ix[:n].copy()


def calculate_integrated_lufs(sample_rate, channels, x):
    """Analyzer::calculate_integrated_lufs -> float or None."""
    x, xp = _f32(x)
    v = C.c_double()
    rc = lib().so_calculate_integrated_lufs(sample_rate, channels, xp, x.size, C.byref(v))
    return None if rc else v.value


class StreamResult(C.Structure):
    _fields_ = [("integrated", C.c_double), ("lra", C.c_double), ("true_peak", C.c_double * 2),
                ("sample_peak", C.c_double * 2), ("n_windows", C.c_size_t), ("n_bins", C.c_size_t),
                ("n_wave_points", C.c_size_t)]


def analyze_stream(sample_rate, x, fft_n=4096, hop=1024, force_tp_factor=0, want_fft=True, want_wave=True,
                   native=False):
    """One pass of the whole hot path over one interleaved stereo stream."""
    x, xp = _f32(x)
    frames = x.size // 2
    nb, _ = fft_bins(sample_rate, fft_n)
    nwin = max(0, frames // hop - fft_n // hop)
    fft = np.zeros((max(nwin, 1), 2, nb), np.float32) if want_fft else None
    w = int(frames / sample_rate * 1000.0)
    wave = np.zeros((2 * w + 2, 2), np.float64) if want_wave else None
    res = StreamResult()
    rc = lib(native).so_analyze_stream(
        sample_rate, xp, x.size, fft_n, hop, force_tp_factor,
        fft.ctypes.data_as(C.POINTER(C.c_float)) if want_fft else None,
        wave.ctypes.data_as(C.POINTER(C.c_double)) if want_wave else None, C.byref(res))
    if rc:
        raise OracleError(rc)
    return {"integrated": res.integrated, "lra": res.lra, "true_peak": list(res.true_peak),
            "sample_peak": list(res.sample_peak), "n_windows": res.n_windows, "n_bins": res.n_bins,
            "fft": fft[:res.n_windows] if want_fft else None,
            "wave": wave[:res.n_wave_points] if want_wave else None}


def analyze_streams_all_cores(sample_rate, xs, n_streams, fft_n, hop, n_threads, reps=1, native=False):
    """Wall-clock seconds for `n_streams` whole analyze_stream passes (stream s = row s % len(xs) of the 2-D array xs) on
    `n_threads` POSIX threads, `reps` times — timed inside the C library (bench.py's all-cores CPU leg)."""
    xs = np.ascontiguousarray(xs, dtype=np.float32)
    el = C.c_double()
    rc = lib(native).so_analyze_streams_mt(sample_rate, xs.ctypes.data_as(C.POINTER(C.c_float)), xs.shape[1], xs.shape[0],
                                           n_streams, fft_n, hop, n_threads, reps, C.byref(el))
    if rc:
        raise OracleError(rc)
    return el.value
