"""The PRODUCT's constant tables (designed on the host by soundscope_amd/csrc/ss_tables.cpp and uploaded to the
GPU) against published numbers and independent numpy / scipy derivations — NOT against the test oracle, whose table
code has the same author (a table error would otherwise be common-mode).  No GPU needed: the
`ss_inspect_*` entry points return the tables exactly as the kernels receive them."""
import ctypes as C

import numpy as np
import pytest
from scipy import signal

from soundscope_amd import _lib as L

f64p, f32p, u32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint32)


def kweight(rate):
    b, a = np.empty(5), np.empty(5)
    assert L.lib().ss_inspect_kweight(rate, b.ctypes.data_as(f64p), a.ctypes.data_as(f64p)) == 0
    return b, a


def test_kweighting_equals_the_bs1770_table_at_48k():
    """ITU-R BS.1770-4, Tables 1 and 2 (48 kHz): shelving stage and RLB high-pass, as printed."""
    shelf_b = [1.53512485958697, -2.69169618940638, 1.19839281085285]
    shelf_a = [1.0, -1.69065929318241, 0.73248077421585]
    hp_b = [1.0, -2.0, 1.0]
    hp_a = [1.0, -1.99004745483398, 0.99007225036621]
    b, a = kweight(48000)
    assert np.allclose(b, np.convolve(shelf_b, hp_b), rtol=0, atol=2e-13)
    assert np.allclose(a, np.convolve(shelf_a, hp_a), rtol=0, atol=2e-13)


@pytest.mark.parametrize("rate", [22050, 44100, 48000, 88200, 96000, 192000])
def test_kweighting_response_across_rates(rate):
    """libebur128 (and the ebur128 crate after it) re-derives the two sections from their analog parameters for every
    rate but leaves the high-pass numerator at [1, -2, 1], so the overall gain drifts slightly with the rate: +0.691 dB at
    997 Hz at 48 kHz (the constant BS.1770 subtracts), +0.72 at 22.05 kHz, +0.66 at 192 kHz.  The product must show
    exactly that family: within 0.045 dB of the 48 kHz table's response below 0.2 fs, a double zero at DC, stable poles."""
    b, a = kweight(rate)
    b48, a48 = kweight(48000)
    f = np.array([30.0, 100.0, 997.0, 2000.0, 4000.0])
    _, h = signal.freqz(b, a, worN=f, fs=rate)
    _, h48 = signal.freqz(b48, a48, worN=f, fs=48000)
    db, db48 = 20 * np.log10(np.abs(h)), 20 * np.log10(np.abs(h48))
    assert abs(db48[2] - 0.691) < 0.0005
    assert np.all(np.abs(db - db48) < 0.045 + (0.03 if rate < 30000 else 0.0)), (db, db48)
    assert abs(np.polyval(b[::-1], 1.0)) < 1e-9                      # H(z = 1) = 0: double zero at DC
    assert np.all(np.abs(np.roots(a)) < 1.0)                          # stable
    # the shelf: about +4 dB above 5 kHz relative to 500 Hz
    _, hs = signal.freqz(b, a, worN=np.array([500.0, min(8000.0, 0.35 * rate)]), fs=rate)
    assert 3.0 < 20 * np.log10(abs(hs[1]) / abs(hs[0])) < 4.3


@pytest.mark.parametrize("factor", [2, 4])
def test_true_peak_interpolator(factor):
    """libebur128's interpolator by definition: 49-tap Hann-windowed sinc, polyphase; derived here with numpy."""
    n_len = C.c_uint32()
    taps = np.zeros(3 * 24, np.float32)
    assert L.lib().ss_inspect_true_peak(factor, taps.ctypes.data_as(f32p), taps.size, C.byref(n_len)) == 0
    n = n_len.value
    assert n == (12 if factor == 4 else 24)
    got = taps[:(factor - 1) * n].reshape(factor - 1, n)
    j = np.arange(49)
    m = j - 24.0
    c = np.sinc(m / factor) * 0.5 * (1.0 - np.cos(2 * np.pi * j / 48.0))
    for f in range(1, factor):
        ref = np.zeros(n, np.float32)
        for jj in range(49):
            if jj % factor == f and abs(c[jj]) > 1e-6:
                ref[jj // factor] = np.float32(c[jj])
        assert np.array_equal(got[f - 1], ref), (f, got[f - 1], ref)
        assert abs(float(got[f - 1].astype(np.float64).sum()) - 1.0) < 0.02     # an interpolation branch has unit DC gain
    if factor == 4:                                                             # branches 1 and 3 mirror each other
        assert np.array_equal(got[0], got[2][::-1])
    assert L.lib().ss_inspect_true_peak(3, None, 0, None) == L.SS_ERR_INVALID_ARG
    assert L.lib().ss_inspect_true_peak(4, taps.ctypes.data_as(f32p), 5, None) == L.SS_ERR_CAPACITY


@pytest.mark.parametrize("n", [8, 4096, 16384])
def test_hann_window(n):
    """Periodic Hann, f32: w[i] = 0.5 (1 - cos(2 pi i / n)); against numpy in f64 to a few f32 ulps."""
    w = np.empty(n, np.float32)
    assert L.lib().ss_inspect_hann(n, w.ctypes.data_as(f32p)) == 0
    ref = 0.5 * (1.0 - np.cos(2 * np.pi * np.arange(n) / n))
    assert np.abs(w.astype(np.float64) - ref).max() < 4e-7
    assert w[0] == 0.0 and abs(float(w[n // 2]) - 1.0) < 1e-7
    assert np.allclose(w[1:], w[1:][::-1], atol=4e-7)                 # periodic window: symmetric about n / 2


def test_retained_bins_match_the_survey():
    """1705 bins from k = 2 (48 k / 4096), 6820 from 7 (48 k / 16384), 7423 from 8 (44.1 k / 16384),
    3410 from 4 (96 k / 16384) — and the rule itself (20 <= k sr / n <= 20000 in f32) with numpy."""
    for rate, n, first, count in [(48000, 4096, 2, 1705), (48000, 16384, 7, 6820), (44100, 16384, 8, 7423), (96000, 16384, 4, 3410)]:
        fb, nb = C.c_uint32(), C.c_uint32()
        assert L.lib().ss_inspect_bins(rate, n, C.byref(fb), C.byref(nb)) == 0
        assert (fb.value, nb.value) == (first, count)
        f = np.arange(n // 2 + 1, dtype=np.float32) * (np.float32(rate) / np.float32(n))
        keep = np.nonzero((f >= 20.0) & (f <= 20000.0))[0]
        assert (int(keep[0]), keep.size) == (first, count)


def test_histogram_tables():
    """ebur128 histogram mode: bin i spans [-70 + i/10, -70 + (i+1)/10) LUFS, energy = 10^((L + 0.691) / 10),
    representative value at the bin centre."""
    e, b = np.empty(1000), np.empty(1001)
    assert L.lib().ss_inspect_histogram(e.ctypes.data_as(f64p), b.ctypes.data_as(f64p)) == 0
    i = np.arange(1001)
    assert np.allclose(b, 10.0 ** ((-70.0 + i / 10.0 + 0.691) / 10.0), rtol=1e-14)
    assert np.allclose(e, 10.0 ** ((-69.95 + i[:1000] / 10.0 + 0.691) / 10.0), rtol=1e-14)
    assert np.all(np.diff(b) > 0) and np.all((e > b[:-1]) & (e < b[1:]))
    # the absolute gate: -70 LUFS
    assert abs(10 * np.log10(b[0]) - 0.691 + 70.0) < 1e-12
