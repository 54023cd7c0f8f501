"""The reference's own unit tests (src/analyzer.rs:185-399, src/tui.rs:2272-2368) restated one-to-one on the
HIP path through the `Analyzer` mirror — same inputs, same assertions, plus the value the CPU oracle gives."""
import numpy as np
import pytest

import soundscope_amd as ssa

pytestmark = pytest.mark.gpu

F = np.float32


def _tone(sr, target):
    res = F(sr) / F(16384.0)
    f = F(np.round(F(target) / res)) * res
    t = np.arange(16384, dtype=np.float32) / F(sr)
    return np.sin(F(2.0) * F(np.pi) * f * t).astype(np.float32)


def test_get_fft(oracle):
    """analyzer.rs:191-220: 440 Hz sine, 16384 samples at 44.1 kHz -> non-empty spectrum."""
    an = ssa.Analyzer()
    t = np.arange(16384, dtype=np.float32) / F(44100)
    x = np.sin(F(2.0) * F(np.pi) * F(440.0) * t).astype(np.float32)
    out = an.get_fft(x)
    assert len(out) > 0 and out.shape == (7423, 2)
    ref = oracle.get_fft(44100, x)
    assert abs(out[:, 1].max() - ref[:, 1].max()) <= 0.01


def test_dbfs_calibration(oracle):
    """analyzer.rs:225-263: full-scale bin-centred sine near 1 kHz reads 0 dB within +-1 dB."""
    an = ssa.Analyzer()
    mx = an.get_fft(_tone(44100, 1000.0))[:, 1].max()
    assert -1.0 <= mx <= 1.0
    assert abs(mx - oracle.get_fft(44100, _tone(44100, 1000.0))[:, 1].max()) <= 0.01


def test_pink_noise_compensation():
    """analyzer.rs:269-322: 125 Hz reads about 9 dB below 1 kHz."""
    an = ssa.Analyzer()
    d = an.get_fft(_tone(44100, 125.0))[:, 1].max() - an.get_fft(_tone(44100, 1000.0))[:, 1].max()
    assert -10.5 <= d <= -8.0


def test_get_waveform():
    """analyzer.rs:326-358: 44100 samples, 15 s window -> 30000 points, (i, min), (i, max) with min <= max."""
    x = np.sin(np.arange(44100, dtype=np.float32) / F(44100.0)).astype(np.float32)
    w = ssa.Analyzer.get_waveform(x, 15.0)
    assert w.shape == (30000, 2)
    for i in range(0, 30000, 2):
        assert w[i, 0] == w[i + 1, 0] == i // 2
        assert w[i, 1] <= w[i + 1, 1]


def test_loudness_measurements():
    """analyzer.rs:362-385: 1 s of 0.1-amplitude 440 Hz stereo; integrated in (-100, 0), true peaks in [0, 1]."""
    an = ssa.Analyzer()
    i = np.arange(88200, dtype=np.float32)
    x = (F(0.1) * np.sin(F(440.0 * 2.0) * F(np.pi) * (i / F(44100.0)))).astype(np.float32)
    an.add_samples(x)
    integrated = an.get_integrated_lufs()
    assert -100.0 < integrated < 0.0
    left, right = an.get_true_peak()
    assert 0.0 <= left <= 1.0 and 0.0 <= right <= 1.0


def test_analyzer_reinit():
    """analyzer.rs:389-398: re-creating the meter for other layouts succeeds and updates the rate."""
    an = ssa.Analyzer()
    an.create_loudness_meter(1, 48000)
    assert an.sample_rate() == 48000
    an.create_loudness_meter(6, 96000)
    assert an.sample_rate() == 96000


@pytest.mark.parametrize("sr", [44100, 48000, 96000])
def test_analyze_microphone_input(sr):
    """tui.rs:2272-2368 as the reference runs them: `create_test_app` builds the capture ring with a capacity of
    44100 * 30 samples (tui.rs:2199) and leaves the device analyzer at Analyzer::default() — 2 channels, 44 100 Hz
    (analyzer.rs:34-45) — for all three tests; the test then enqueues sr * 30 samples of a 500 Hz tone generated at `sr`,
    so for 48 k and 96 k the ring keeps the LAST 44100 * 30 of them.  analyze_microphone_input (tui.rs:1427-1480) slices
    with the analyzer's 44 100; the bin the test looks at (index 500 / (sr / 2) * len) must read below -20 dB."""
    i = np.arange(sr * 30, dtype=np.float32)
    tone = np.sin(i * F(500.0) * F(2.0) * F(np.pi) / F(sr)).astype(np.float32)
    ring = tone[-44100 * 30:]                                   # AllocRingBuffer::new(44100 * 30): oldest first, newest kept
    sess = ssa.CaptureSession(2, 44100)                         # device_analyzer: Analyzer::default()
    res = sess.analyze_microphone_input(ring)
    assert res.mid_status == 0 and sess.mid_fft.shape[0] > 0    # assert!(!app.fft_data.mid_fft.is_empty())
    idx = int(round(500.0 / (sr / 2.0) * sess.mid_fft.shape[0]))
    assert idx < sess.mid_fft.shape[0] and sess.mid_fft[idx, 1] < -20.0
    assert sess.microphone_input_chart.shape == (30000, 2)
    assert sess.lufs[299] == res.shortterm and np.isfinite(res.shortterm)
