"""CPU restatement of the render-side step after the analyzer.  TEST INFRASTRUCTURE ONLY.

Reference-defined parts: `y + fft_gain_compensation_db` in f64 (tui.rs:801-821), gain = FFT_TARGET_LUFS - integrated
in f32 (tui.rs:1229-1238), chart bounds [-100, 0] dB (tui.rs:49-51, :890), waveform view bounds (tui.rs:664-681).
The column rule (what ends up in one terminal column) is the library's own and restated here from
include/soundscope_hip.h.
"""
import numpy as np


def gain_db(integrated):
    if integrated is None:
        return 0.0
    return float(np.float32(-13.0) - np.float32(integrated))


def spectrum_columns(xy, gain, cols):
    """xy: get_fft output [(chart_x, dB)] -> cols values: max over the column of clamp(dB + gain, -100, 0)."""
    x, y = np.asarray(xy)[:, 0], np.asarray(xy)[:, 1]
    c = np.minimum(np.floor(x / 100.0 * cols), cols - 1).astype(np.int64)
    v = np.clip(y + float(gain), -100.0, 0.0)
    out = np.full(cols, np.nan)
    for ci in np.unique(c):
        out[ci] = v[c == ci].max()
    return out


def waveform_view(playhead_ms, window_s, chart_points):
    half = window_s * 500.0
    max_x = chart_points / 2.0
    lo = max(min(playhead_ms - half, max_x - window_s * 1000.0), 0.0)
    hi = max(min(playhead_ms + half, max_x), window_s * 1000.0)
    return lo, hi


def waveform_columns(chart, x_min, x_max, cols):
    """chart: get_waveform output [(i, min), (i, max), ...] -> [cols, 2] (min of mins, max of maxes)."""
    mm = np.asarray(chart)[:, 1].reshape(-1, 2)
    out = np.full((cols, 2), np.nan)
    span = x_max - x_min
    for i in range(int(x_min), min(int(x_max), mm.shape[0])):
        c = (i - x_min) * cols // span
        lo, hi = mm[i]
        out[c, 0] = lo if np.isnan(out[c, 0]) else min(out[c, 0], lo)
        out[c, 1] = hi if np.isnan(out[c, 1]) else max(out[c, 1], hi)
    return out
