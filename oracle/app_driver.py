"""CPU restatement of the reference App's per-file / per-device analysis state and tick drivers.

TEST INFRASTRUCTURE ONLY (see oracle/ss_oracle.h).  Follows /root/reference/src/tui.rs line by line:
  FileApp.__init__            receive_audio_file           tui.rs:1207-1241
                              AudioFile::from_file         audio_player.rs:146-166
  FileApp.analyze_audio_file_samples                       tui.rs:1482-1552
  CaptureApp.analyze_microphone_input                      tui.rs:1427-1480
  restart                     play / seek handlers         tui.rs:1586-1614
All arithmetic goes through the oracle's C functions (oracle/ss_oracle.c).
"""
import numpy as np

from . import pyoracle as O

FFT_TARGET_LUFS = np.float32(-13.0)      # tui.rs:49


class _Analyzer:
    """analyzer.rs:29-183 on the oracle: one meter + sample_rate."""

    def __init__(self):
        self.sample_rate = 44100
        self.meter = O.Meter(2, 44100)

    def create_loudness_meter(self, channels, rate):
        self.sample_rate = rate                      # analyzer.rs:50: before the fallible call
        self.meter = O.Meter(channels, rate)

    def get_fft(self, x):
        return O.get_fft(self.sample_rate, x)


class _App:
    def __init__(self):
        self.lufs = np.full(300, -100.0)             # tui.rs:463
        self.mid_fft = np.zeros((0, 2))
        self.side_fft = np.zeros((0, 2))
        self.errors = []

    def restart(self):
        self.lufs = np.full(300, -100.0)
        self.analyzer.meter.reset()

    def _fft_or_fallback(self, x):
        try:
            return self.analyzer.get_fft(x), 0
        except O.OracleError as e:
            return np.zeros((1, 2)), e.code          # vec![(0., 0.)]

    def _feed_and_read(self, x):
        add_status = st_status = 0
        try:
            self.analyzer.meter.add_frames(x)
        except O.OracleError as e:
            add_status = e.code
        try:
            self.lufs[299] = self.analyzer.meter.shortterm()
        except O.OracleError as e:
            st_status = e.code
            self.lufs[299] = 0.0
        return add_status, st_status


class FileApp(_App):
    def __init__(self, samples, channels, sample_rate):
        super().__init__()
        self.samples = np.ascontiguousarray(samples, np.float32)
        self.channels = channels
        self.mid, self.side = O.mid_side(self.samples)
        duration = self.mid.size / float(sample_rate) * 1000.0
        self.duration_ms = int(duration)                                     # Duration::from_millis(d as u64)
        secs = float(self.duration_ms // 1000) + float((self.duration_ms % 1000) * 1000000) / 1e9
        self.audio_file_chart = O.get_waveform(self.samples, secs)
        self.analyzer = _Analyzer()
        self.analyzer.create_loudness_meter(2, sample_rate)
        integrated = O.calculate_integrated_lufs(sample_rate, 2, self.samples)
        self.fft_gain_compensation_db = float(FFT_TARGET_LUFS - np.float32(integrated)) if integrated is not None else 0.0

    def analyze_audio_file_samples(self, pos):
        r = dict(fft_ran=0, mid_status=0, side_status=0, lufs_ran=0, fed=0, add_status=0, shortterm_status=0)
        pos = pos // self.channels
        r["playhead"] = pos
        fft_lb = max(pos - 16384, 0)
        if fft_lb != 0:
            r["fft_ran"] = 1
            mid = self.mid[fft_lb:pos] if (pos <= self.mid.size and fft_lb < self.mid.size) else self.mid[:0]
            side = self.side[fft_lb:pos] if (pos <= self.side.size and fft_lb < self.side.size) else self.side[:0]
            self.mid_fft, r["mid_status"] = self._fft_or_fallback(mid)
            self.side_fft, r["side_status"] = self._fft_or_fallback(side)
        pos = pos * self.channels
        lufs_lb = max(pos - 16384, 0)
        if lufs_lb != 0:
            r["lufs_ran"] = 1
            self.lufs[:-1] = self.lufs[1:]
            if pos <= self.samples.size and lufs_lb < self.samples.size:
                r["fed"] = 1
                r["add_status"], r["shortterm_status"] = self._feed_and_read(self.samples[lufs_lb:pos])
        r["shortterm"] = self.lufs[299]
        return r


class CaptureApp(_App):
    def __init__(self, channels, sample_rate):
        super().__init__()
        self.analyzer = _Analyzer()
        self.analyzer.create_loudness_meter(channels, sample_rate)
        self.microphone_input_chart = np.zeros((0, 2))

    def analyze_microphone_input(self, latest):
        samples = np.ascontiguousarray(latest, np.float32)
        r = dict(fft_ran=1, lufs_ran=1, fed=1)
        mid, side = O.mid_side(samples)
        sr = self.analyzer.sample_rate
        lb = 15 * sr - 2 ** 14
        self.mid_fft, r["mid_status"] = self._fft_or_fallback(mid[lb:15 * sr])
        self.side_fft, r["side_status"] = self._fft_or_fallback(side[lb:15 * sr])
        self.microphone_input_chart = O.get_waveform(mid, 15.0)
        self.lufs[:-1] = self.lufs[1:]
        lb = 30 * sr - 2 ** 14
        r["add_status"], r["shortterm_status"] = self._feed_and_read(samples[lb:30 * sr])
        r["shortterm"] = self.lufs[299]
        return r
