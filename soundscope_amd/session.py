"""Host-side mirror of the reference App's analysis state and tick drivers.

`FileSession`  = `receive_audio_file` (tui.rs:1207-1241) + `analyze_audio_file_samples` (tui.rs:1482-1552)
`CaptureSession` = device selection (tui.rs:1780-1808) + `analyze_microphone_input` (tui.rs:1427-1480)

Field names follow the reference: `fft_data.mid_fft` / `side_fft` -> `mid_fft` / `side_fft`, `lufs` (300
short-term values), `waveform.audio_file_chart` -> `audio_file_chart`, `fft_gain_compensation_db`.
One tick is one C call (`ss_session_tick_*`): the file is resident in HBM, nothing is uploaded per tick.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .analyzer import Analyzer, _check, _f32

LUFS_HISTORY = 300
TICK_WINDOW = 16384


class _BorrowedAnalyzer(Analyzer):
    """The session's own Analyzer (file_analyzer / device_analyzer); owned by the session."""

    def __init__(self, handle):        # no create: the handle belongs to the session
        self._h = handle

    def close(self):
        self._h = None


class _Session:
    def __init__(self):
        self._s = C.c_void_p()
        self.mid_fft = np.zeros((0, 2))          # fft_data.mid_fft
        self.side_fft = np.zeros((0, 2))
        self.last = None                         # ss_tick_result of the latest tick

    def _after_open(self, cap_pairs):
        self._cap = max(int(cap_pairs), 1)
        self._mid = np.empty((self._cap, 2), np.float64)
        self._side = np.empty((self._cap, 2), np.float64)
        self.analyzer = _BorrowedAnalyzer(C.c_void_p(L.lib().ss_session_analyzer(self._s)))

    def close(self):
        if getattr(self, "_s", None):
            self.analyzer.close()
            L.lib().ss_session_close(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def lufs(self) -> np.ndarray:
        out = np.empty(LUFS_HISTORY, np.float64)
        _check(L.lib().ss_session_lufs_history(self._s, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def restart(self) -> None:
        """play / seek: `lufs = [-100.; 300]; analyzer.reset()` (tui.rs:1586-1614)."""
        _check(L.lib().ss_session_restart(self._s))

    def _store(self, res):
        self.last = res
        if res.fft_ran:
            self.mid_fft = self._mid[:res.n_mid].copy()
            self.side_fft = self._side[:res.n_side].copy()


class FileSession(_Session):
    def __init__(self, samples, channels: int, sample_rate: int):
        super().__init__()
        a, ap = _f32(samples)
        _check(L.lib().ss_session_open_file(ap, a.size, channels, sample_rate, C.byref(self._s)))
        self._after_open(TICK_WINDOW // 2 + 1)
        g = C.c_float()
        _check(L.lib().ss_session_gain_db(self._s, C.byref(g)))
        self.fft_gain_compensation_db = g.value
        d = C.c_uint64()
        _check(L.lib().ss_session_duration_ms(self._s, C.byref(d)))
        self.duration_ms = d.value
        cap = 2 * self.duration_ms + 2
        out = np.empty((cap, 2), np.float64)
        n = C.c_size_t(0)
        _check(L.lib().ss_session_waveform(self._s, out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)))
        self.audio_file_chart = out[:n.value].copy()

    def analyze_audio_file_samples(self, pos: int):
        """One tick at playback position `pos` (interleaved samples).  Returns the tick record."""
        res = L.TickResult()
        dp = C.POINTER(C.c_double)
        _check(L.lib().ss_session_tick_file(self._s, pos, self._mid.ctypes.data_as(dp),
                                            self._side.ctypes.data_as(dp), self._cap, C.byref(res)))
        self._store(res)
        return res


class CaptureSession(_Session):
    def __init__(self, channels: int, sample_rate: int):
        super().__init__()
        _check(L.lib().ss_session_open_capture(channels, sample_rate, C.byref(self._s)))
        self._after_open(TICK_WINDOW // 2 + 1)
        self.sample_rate = sample_rate
        self._wave = np.empty((2 * 15000 + 2, 2), np.float64)
        self.microphone_input_chart = np.zeros((0, 2))

    def analyze_microphone_input(self, latest_captured_samples):
        """One tick on a snapshot of the 30*rate-sample capture ring (oldest first)."""
        a, ap = _f32(latest_captured_samples)
        res = L.TickResult()
        n = C.c_size_t(0)
        dp = C.POINTER(C.c_double)
        _check(L.lib().ss_session_tick_capture(self._s, ap, a.size, self._mid.ctypes.data_as(dp),
                                               self._side.ctypes.data_as(dp), self._cap,
                                               self._wave.ctypes.data_as(dp), self._wave.shape[0], C.byref(n),
                                               C.byref(res)))
        self._store(res)
        self.microphone_input_chart = self._wave[:n.value].copy()
        return res

    def push(self, samples) -> None:
        """The capture callback's `audio_buf.extend(data)` for a ring that lives on the device (host work only)."""
        a, ap = _f32(samples)
        _check(L.lib().ss_session_capture_push(self._s, ap, a.size))

    def analyze_resident(self):
        """One tick on the device-resident ring as it stands after every push so far (nothing but the new samples is uploaded)."""
        res = L.TickResult()
        n = C.c_size_t(0)
        dp = C.POINTER(C.c_double)
        _check(L.lib().ss_session_tick_capture_resident(self._s, self._mid.ctypes.data_as(dp), self._side.ctypes.data_as(dp), self._cap,
                                                        self._wave.ctypes.data_as(dp), self._wave.shape[0], C.byref(n), C.byref(res)))
        self._store(res)
        self.microphone_input_chart = self._wave[:n.value].copy()
        return res
