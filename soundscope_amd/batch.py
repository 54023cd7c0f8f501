"""Batch extension: many equal-length streams resident in HBM, analysed in one pass.

Not part of the reference API (soundscope analyses one file at a time); it is the
data-parallel form of receive_audio_file + analyze_audio_file_samples
(reference src/tui.rs:1207-1241, :1482-1552) over a corpus.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .analyzer import _check


class Batch:
    def __init__(self, sample_rate=48000, channels=2, n_streams=1, frames_per_stream=480000,
                 fft_n=4096, hop_frames=1024, flags=L.SS_BATCH_ALL, true_peak_factor=0,
                 waveform_window=0.0, spectrum_columns=0):
        """spectrum_columns > 0 (with L.SS_BATCH_FFT_COLUMNS in flags, added here if missing): columns-only spectrum — the
        render-side reduction runs inside the spectrum kernel and only that many chart columns per row are kept."""
        if spectrum_columns:
            flags |= L.SS_BATCH_FFT_COLUMNS
            self._render_cols = int(spectrum_columns)
        cfg = L.BatchConfig(sample_rate, channels, n_streams, fft_n, hop_frames, flags, true_peak_factor, int(spectrum_columns),
                            frames_per_stream, waveform_window)
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = L.lib().ss_batch_create(C.byref(cfg), C.byref(self._h))
        if rc != L.SS_OK:
            self._h = None
            _check(rc)
        lay = L.BatchLayout()
        _check(L.lib().ss_batch_layout_get(self._h, C.byref(lay)))
        self.layout = lay

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ss_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def samples_per_stream(self):
        return int(self.cfg.frames_per_stream) * int(self.cfg.channels)

    def upload(self, first, pcm):
        a = np.ascontiguousarray(pcm, dtype=np.float32)
        count = a.size // self.samples_per_stream
        assert count * self.samples_per_stream == a.size
        _check(L.lib().ss_batch_upload(self._h, first, count, a.ctypes.data_as(C.POINTER(C.c_float))))

    def download_input(self, stream):
        out = np.empty(self.samples_per_stream, np.float32)
        _check(L.lib().ss_batch_download_input(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def input_device_ptr(self):
        return L.lib().ss_batch_input_device_ptr(self._h)

    def synthesize(self, seed=0x5EED0000, first_stream_id=0):
        _check(L.lib().ss_batch_synthesize(self._h, seed, first_stream_id))

    def run(self):
        _check(L.lib().ss_batch_run(self._h))

    def sync(self):
        _check(L.lib().ss_batch_sync(self._h))

    # -- ragged batches: every stream its own length inside its frames_per_stream slot
    def set_lengths(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint64)
        _check(L.lib().ss_batch_set_lengths(self._h, a.ctypes.data_as(C.POINTER(C.c_uint64)), a.size))
        self._ragged = True

    def stream_shape(self, stream):
        sh = L.StreamShape()
        _check(L.lib().ss_batch_stream_shape(self._h, stream, C.byref(sh)))
        return sh

    def results(self):
        n = int(self.cfg.n_streams)
        arr = (L.StreamResult * n)()
        _check(L.lib().ss_batch_results(self._h, arr, n))
        return arr

    def peaks(self, stream):
        """Every channel's (true_peak, sample_peak) of one stream, linear: arrays [channels] f64."""
        ch = int(self.cfg.channels)
        tp, sp = np.empty(ch, np.float64), np.empty(ch, np.float64)
        dp = C.POINTER(C.c_double)
        _check(L.lib().ss_batch_peaks(self._h, stream, tp.ctypes.data_as(dp), sp.ctypes.data_as(dp), ch))
        return tp, sp

    @property
    def geometry(self):
        g = L.BatchGeometry()
        _check(L.lib().ss_batch_geometry_get(self._h, C.byref(g)))
        return g

    def set_overlap(self, mode=True):
        """0 / False: sequential; 1 / True: spectrum kernel on a second HIP stream beside the time-domain chain;
        2: beside the chain's tail only (time-domain kernel first and alone).  Same results in every mode."""
        _check(L.lib().ss_batch_set_overlap(self._h, int(mode)))

    def set_time_domain_mode(self, mode):
        """L.SS_TD_AUTO (time segments with the exact state hand-over), L.SS_TD_RUN_IN (segments with the 0.1 s run-in of earlier
        rounds, for comparison), L.SS_TD_WHOLE_STREAMS (a workgroup per stream, where the shape allows)."""
        _check(L.lib().ss_batch_set_time_domain_mode(self._h, int(mode)))

    def set_true_peak_arith(self, arith):
        """L.SS_TP_ARITH_F32 (default: f32 MFMA, the width of the crate's interpolator) or L.SS_TP_ARITH_F16X3 (opt-in: f16x3
        split on the matrix cores, within 2^-21 of the tile peak of the f32 result, faster)."""
        _check(L.lib().ss_batch_set_true_peak_arith(self._h, int(arith)))

    @property
    def true_peak_arith(self):
        return int(L.lib().ss_batch_get_true_peak_arith(self._h))

    def allreduce_histograms(self, comm):
        """The corpus gate's exchange: in-place SUM all-reduce of this batch's corpus histograms over `comm`
        (soundscope_amd.distributed.Comm).  Returns (block_hist, shortterm_hist) of the whole corpus."""
        out = np.empty(2000, np.uint64)
        _check(L.lib().ss_batch_allreduce_histograms(self._h, comm._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out[:1000].copy(), out[1000:].copy()

    def corpus_gate_enqueue(self, comm=None):
        """Queue the corpus gate on the device behind run(): all-reduce over `comm` (if any) + gate + LRA; no wait."""
        _check(L.lib().ss_batch_corpus_gate_enqueue(self._h, comm._h if comm is not None else None))

    def corpus_gate_read(self):
        """(integrated LUFS, LRA) of the last queued corpus gate (waits for the batch's stream)."""
        i, r = C.c_double(), C.c_double()
        _check(L.lib().ss_batch_corpus_gate_read(self._h, C.byref(i), C.byref(r)))
        return i.value, r.value

    def traffic_floor(self, reps=5):
        """ms per launch of the spectrum kernel's loads and stores alone (measurement utility; clobbers the spectra)."""
        ms = C.c_double()
        _check(L.lib().ss_batch_traffic_floor(self._h, reps, C.byref(ms)))
        return ms.value

    def checksums(self):
        """[n_streams][3] u64, computed on the device: order-independent checksums of every stream's whole spectrum block,
        decimation bins and sub-block energies (bit patterns).  Equal checksums <=> bit-equal data (up to 2^-64)."""
        n = int(self.cfg.n_streams)
        out = np.zeros((n, 3), np.uint64)
        _check(L.lib().ss_batch_checksums(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n))
        return out

    def fft(self, stream):
        lay = self.layout
        out = np.empty((lay.n_windows, lay.fft_channels, lay.n_bins), np.float32)
        _check(L.lib().ss_batch_download_fft(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out[:self.stream_shape(stream).n_windows] if getattr(self, "_ragged", False) else out

    def bin_tables(self):
        n = self.layout.n_bins
        x, f, p = (np.empty(n, np.float64) for _ in range(3))
        dp = C.POINTER(C.c_double)
        _check(L.lib().ss_batch_bin_tables(self._h, x.ctypes.data_as(dp), f.ctypes.data_as(dp), p.ctypes.data_as(dp)))
        return x, f, p

    def waveform(self, stream):
        pts = self.layout.n_wave_points
        out = np.empty(pts, np.float32)
        _check(L.lib().ss_batch_download_waveform(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        if getattr(self, "_ragged", False):
            out = out[:self.stream_shape(stream).n_wave_points]
        return out.reshape(-1, 2)          # [bin] -> (min, max)

    # -- render-side reductions (tui.rs:49-51, :801-821, :664-681)
    def render_spectrum(self, cols, gain_db=None):
        """Reduce every spectrum row to `cols` chart columns on the device.  gain_db=None uses the
        reference's per-file rule -13 - integrated (tui.rs:1234)."""
        mode = L.SS_GAIN_REFERENCE if gain_db is None else L.SS_GAIN_FIXED
        _check(L.lib().ss_batch_render_spectrum(self._h, cols, mode, 0.0 if gain_db is None else float(gain_db)))
        self._render_cols = cols

    def set_columns_gain(self, gain_db=None):
        """columns-only batches: None = the reference's per-file rule -13 - integrated (needs the meter pass), else a fixed gain"""
        mode = L.SS_GAIN_REFERENCE if gain_db is None else L.SS_GAIN_FIXED
        _check(L.lib().ss_batch_set_columns_gain(self._h, mode, 0.0 if gain_db is None else float(gain_db)))

    def spectrum_columns(self, stream):
        lay = self.layout
        out = np.empty((lay.n_windows, lay.fft_channels, self._render_cols), np.float32)
        _check(L.lib().ss_batch_download_spectrum_columns(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def render_waveform(self, cols, x_min, x_max):
        _check(L.lib().ss_batch_render_waveform(self._h, cols, int(x_min), int(x_max)))
        self._render_wave_cols = cols

    def waveform_columns(self, stream):
        out = np.empty((self._render_wave_cols, 2), np.float32)
        _check(L.lib().ss_batch_download_waveform_columns(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def subblocks(self, stream):
        n = self.layout.n_subblocks
        out = np.empty((n, int(self.cfg.channels)), np.float64)
        _check(L.lib().ss_batch_download_subblocks(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def histograms(self):
        out = np.empty(2000, np.uint64)
        _check(L.lib().ss_batch_histograms(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out[:1000].copy(), out[1000:].copy()

    def histograms_to_device(self, dev_ptr):
        _check(L.lib().ss_batch_histograms_device(self._h, C.c_void_p(dev_ptr)))

    def timing_enable(self, on=True):
        _check(L.lib().ss_batch_timing_enable(self._h, 1 if on else 0))

    def timing_read(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        _check(L.lib().ss_batch_timing_read(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def corpus_integrated_lufs(block_hist):
    h = np.ascontiguousarray(block_hist, dtype=np.uint64)
    return L.lib().ss_corpus_integrated_lufs(h.ctypes.data_as(C.POINTER(C.c_uint64)))


def corpus_loudness_range(st_hist):
    h = np.ascontiguousarray(st_hist, dtype=np.uint64)
    return L.lib().ss_corpus_loudness_range(h.ctypes.data_as(C.POINTER(C.c_uint64)))


def waveform_view(playhead_ms, waveform_window_s, chart_points):
    """Player-mode x bounds of the waveform chart (tui.rs:664-681) -> (x_min, x_max)."""
    lo, hi = C.c_double(), C.c_double()
    L.lib().ss_waveform_view(float(playhead_ms), float(waveform_window_s), int(chart_points), C.byref(lo), C.byref(hi))
    return lo.value, hi.value
