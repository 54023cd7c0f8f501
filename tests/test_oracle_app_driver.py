"""The CPU restatement of the reference's tick drivers (oracle/app_driver.py) against the rules
readable in /root/reference/src/tui.rs:1207-1241, :1427-1552 (restated here as assertions)."""
import numpy as np

from conftest import make_stereo
from oracle import pyoracle as O
from oracle.app_driver import CaptureApp, FileApp


def test_file_driver_rules():
    rate = 48000
    x = make_stereo(11, rate * 3, rate=rate)
    app = FileApp(x, 2, rate)
    # AudioFile::from_file: duration = mid.len() / rate * 1000 ms (audio_player.rs:153); chart = 2 points per ms
    assert app.duration_ms == 3000
    assert app.audio_file_chart.shape == (6000, 2)
    assert np.array_equal(app.audio_file_chart, O.get_waveform(x, 3.0))
    # gain = -13 - integrated, in f32 (tui.rs:1229-1235)
    want = np.float32(-13.0) - np.float32(O.calculate_integrated_lufs(rate, 2, x))
    assert app.fft_gain_compensation_db == float(want)
    # pos <= 16384 frames: saturating_sub gives 0 -> both blocks skipped, history untouched (tui.rs:1489,1530)
    r = app.analyze_audio_file_samples(2 * 16384)
    assert (r["fft_ran"], r["lufs_ran"]) == (0, 1)          # frames 16384 -> fft skipped; samples 32768 -> lufs runs
    r = app.analyze_audio_file_samples(16384)
    assert (r["fft_ran"], r["lufs_ran"], r["fed"]) == (0, 0, 0)
    # a regular tick: 16384-sample mid/side windows ending at the playhead, LUFS fed with the last 16384 samples
    before = app.lufs.copy()
    r = app.analyze_audio_file_samples(2 * 40000)
    mid, side = O.mid_side(x)
    assert np.array_equal(app.mid_fft, O.get_fft(rate, mid[40000 - 16384:40000]))
    assert np.array_equal(app.side_fft, O.get_fft(rate, side[40000 - 16384:40000]))
    assert np.array_equal(app.lufs[:-1], before[1:])
    assert app.lufs[299] == r["shortterm"] and r["fed"] == 1
    # past the end: get_fft(&[]) -> TooFewSamples -> [(0, 0)]; history shifts without a new value
    prev = app.lufs[299]
    r = app.analyze_audio_file_samples(x.size + 4096)
    assert r["mid_status"] == O.lib().so_get_fft(rate, None, 0, None, 0, None) == 10
    assert np.array_equal(app.mid_fft, np.zeros((1, 2)))
    assert r["fed"] == 0 and app.lufs[299] == prev and app.lufs[298] == prev
    app.restart()
    assert np.all(app.lufs == -100.0)


def test_capture_driver_rules():
    rate = 44100
    ring = make_stereo(4, 15 * rate, rate=rate)
    app = CaptureApp(2, rate)
    r = app.analyze_microphone_input(ring)
    mid, _ = O.mid_side(ring)
    assert np.array_equal(app.mid_fft, O.get_fft(rate, mid[15 * rate - 16384:]))
    assert np.array_equal(app.microphone_input_chart, O.get_waveform(mid, 15.0))
    assert app.microphone_input_chart.shape == (30000, 2)
    m = O.Meter(2, rate)
    m.add_frames(ring[30 * rate - 16384:])
    assert r["shortterm"] == m.shortterm() == app.lufs[299]
    assert app.lufs[298] == -100.0
