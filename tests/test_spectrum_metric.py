"""The spectrum parity metric itself (tests/conftest.py: db_close, db_close_survey, window_peak_db) on synthetic rows — CPU only.
What the GPU tests assert with it is only as good as the metric: its two regimes, the survey wording beside it, and the second
look of the randomised tools (levels without the pink term, counted from the window's strongest component over ALL bins)."""
import numpy as np

from conftest import db_close, db_close_survey, window_peak_db


def test_row_peak_metric_regimes():
    ref = np.array([-6.0, -40.0, -75.9, -76.1, -120.0])
    assert db_close(ref, ref)
    assert db_close(ref + np.array([0.009, -0.009, 0.009, 0.0, 0.0]), ref)
    assert not db_close(ref + np.array([0.0, 0.011, 0.0, 0.0, 0.0]), ref)            # within 70 dB of the row's peak: 0.01 dB
    assert not db_close(ref + np.array([0.0, 0.0, 0.011, 0.0, 0.0]), ref)            # 69.9 dB under it: still the dB regime
    assert db_close(ref + np.array([0.0, 0.0, 0.0, 0.5, 3.0]), ref)                  # 70.1 dB under it and below: 1e-4 of the peak's amplitude
    assert not db_close(ref + np.array([0.0, 0.0, 0.0, 6.0, 0.0]), ref)              # ... which -76.1 -> -70.1 dB exceeds (1.56e-4)
    # a quiet row is held to its OWN peak, not to full scale
    quiet = ref - 60.0
    assert not db_close(quiet + np.array([0.0, 0.011, 0.0, 0.0, 0.0]), quiet)
    # the survey wording: 0.01 dB at >= -90 dBFS whatever the row's peak
    loud = np.array([-1.0, -85.0, -95.0])
    assert db_close(loud + np.array([0.0, 0.05, 0.0]), loud) and not db_close_survey(loud + np.array([0.0, 0.05, 0.0]), loud)
    assert not db_close(loud + np.array([0.0, 0.05, 0.0]), loud, survey=True)


def test_second_look_counts_from_the_windows_strongest_component(oracle):
    """A DC offset sits in bin 0, outside the retained band: the row's own peak does not see it, the transform's rounding noise does."""
    n, rate = 16384, 48000
    t = np.arange(n) / rate
    x = (0.05 + 0.005 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)           # DC 20 dB above the tone
    pk_all = window_peak_db(oracle, x)
    assert abs(pk_all - 20 * np.log10(0.05 * 2)) < 0.01                              # Hann's DC gain is 1/2, the scale 4 / N: 0.1 -> -20 dB
    ref = oracle.get_fft(rate, x)[:, 1]
    cnt, first = oracle.fft_bins(rate, n)
    f = (np.arange(cnt) + first) * (np.float32(rate) / np.float32(n))
    pink = 10 * np.log10(f.astype(np.float64) / 1000.0)
    assert ref.max() - pink[np.argmax(ref)] < pk_all - 15                            # the tone, without its pink term, is far under the DC
    k = int(np.argmin(np.abs((ref - pink) - (pk_all - 80.0))))                       # a bin 80 dB under the DC term ...
    assert (ref - pink)[k] > (ref - pink).max() - 70.0                               # ... yet within 70 dB of the ROW's own peak
    got = ref.copy(); got[k] += 0.02
    assert not db_close(got, ref, 0.015)                                             # the row-peak metric flags it,
    assert db_close(got, ref, 0.015, peak=pk_all, pink=pink)                         # the second look files it under the linear regime
    got[k] += 20.0
    assert not db_close(got, ref, 0.015, peak=pk_all, pink=pink)                     # (which still bounds it: 1e-4 of the strongest component)
