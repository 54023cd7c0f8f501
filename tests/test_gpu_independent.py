"""The DEVICE path against oracle-independent computations (numpy / scipy in f64 from the published definitions) — the
second line of evidence next to the oracle comparisons, since product and oracle share an author:

  * spectrum: mid / side, periodic Hann, |rfft| in f64, 20 log10(mag 4 / N), pink compensation 10 log10(f / 1000),
    bins 20 Hz ... 20 kHz (analyzer.rs:11-27, :55-105; audio_player.rs:400-419);
  * K-weighted loudness: BS.1770 gating on 400 ms blocks of a sequential f64 filter (coefficients from the meter);
  * true peak: the 49-tap Hann-windowed sinc interpolator as a polyphase convolution in f64.
"""
import numpy as np
import pytest
from scipy import signal

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import db_close, db_report, make_stereo

pytestmark = pytest.mark.gpu


def spectrum_f64(x, rate, n):
    """One window of the reference's get_fft restated with numpy in f64 (window values rounded to f32 as the crate's)."""
    i = np.arange(n, dtype=np.float64)
    w = (0.5 * (1.0 - np.cos(2.0 * np.pi * i / n))).astype(np.float32).astype(np.float64)
    mag = np.abs(np.fft.rfft(x.astype(np.float64) * w))
    freq32 = np.arange(n // 2 + 1, dtype=np.float32) * (np.float32(rate) / np.float32(n))
    keep = (freq32 >= 20.0) & (freq32 <= 20000.0)
    with np.errstate(divide="ignore"):
        db = np.where(mag[keep] == 0.0, -150.0, 20.0 * np.log10(mag[keep] * 4.0 / n))
    return db + 10.0 * np.log10(freq32[keep].astype(np.float64) / 1000.0)


@pytest.mark.parametrize("rate,n,hop", [(48000, 4096, 1024), (44100, 4096, 1024), (48000, 16384, 1024), (96000, 16384, 1024)])
def test_spectrum_against_numpy_f64(rate, n, hop):
    frames = n + hop * 9 + 5
    xs = [make_stereo(40 + i, frames, rate, level=0.2 + 0.3 * i) for i in range(2)]
    b = ssa.Batch(rate, 2, 2, frames, n, hop, flags=L.SS_BATCH_FFT)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    lay = b.layout
    for i, x in enumerate(xs):
        lr = x.reshape(-1, 2)
        mid = ((lr[:, 0] + lr[:, 1]) / np.float32(2.0)).astype(np.float32)
        side = ((lr[:, 0] - lr[:, 1]) / np.float32(2.0)).astype(np.float32)
        got = b.fft(i)
        assert got.shape[0] == lay.n_windows > 0
        for wdx in range(lay.n_windows):
            p = (wdx + n // hop + 1) * hop                       # window = frames [p - N, p), left edge > 0 (tui.rs:1489)
            for c, sig in enumerate((mid, side)):
                ref = spectrum_f64(sig[p - n:p], rate, n)
                assert db_close(got[wdx, c], ref, 0.01), (i, wdx, c, db_report(got[wdx, c], ref))     # 0.01 dB down to 70 dB under the row's peak


def lufs_f64(x, rate, channels, coeffs):
    """BS.1770-4 integrated loudness with exact (non-histogram) gating of a sequential f64 K-weighting."""
    b, a = coeffs
    y = signal.lfilter(b, a, x.reshape(-1, channels).astype(np.float64), axis=0)
    s100 = (rate + 5) // 10
    n = y.shape[0] // s100
    sub = np.array([np.sum(y[k * s100:(k + 1) * s100] ** 2, axis=0) for k in range(n)])          # [sub-block][channel]
    blocks = np.array([sub[k - 3:k + 1].sum(axis=0).sum() / (4.0 * s100) for k in range(3, n)])  # L = R = 1.0
    absolute = blocks[10.0 * np.log10(np.maximum(blocks, 1e-300)) - 0.691 >= -70.0]
    if absolute.size == 0:
        return -np.inf
    rel = absolute.mean() * 0.1
    gated = absolute[absolute >= rel]
    return 10.0 * np.log10(gated.mean()) - 0.691


@pytest.mark.parametrize("rate", [44100, 48000, 96000, 192000])
def test_integrated_loudness_against_exact_f64_gating(oracle, rate):
    """The device bins block energies in 0.1 LU steps like the crate's histogram mode; exact gating of the same blocks
    differs from that by at most half a bin in each block's weight — 0.05 LU is the bar here, the oracle comparisons hold
    the 0.01 LU one."""
    frames = rate * 6 + 17
    xs = [make_stereo(70 + i, frames, rate, level=0.05 + 0.4 * i, gap=(i == 1)) for i in range(3)]
    b = ssa.Batch(rate, 2, 3, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    coeffs = oracle.Meter(2, rate).coeffs()            # the design is pinned separately (tests/test_product_tables.py)
    for i, x in enumerate(xs):
        assert abs(b.results()[i].integrated_lufs - lufs_f64(x, rate, 2, coeffs)) <= 0.05, i


def interpolator_taps(factor):
    j = np.arange(49, dtype=np.float64)
    m = j - 24.0
    with np.errstate(invalid="ignore", divide="ignore"):
        c = np.where(np.abs(m) > 1e-6, np.sin(m * np.pi / factor) / (m * np.pi / factor), 1.0)
    c *= 0.5 * (1.0 - np.cos(2.0 * np.pi * j / 48.0))
    return c.astype(np.float32).astype(np.float64)      # the crate keeps f32 taps


@pytest.mark.parametrize("rate,factor", [(48000, 4), (44100, 4), (96000, 2)])
def test_true_peak_against_f64_polyphase(rate, factor):
    frames = rate * 2 + 33
    xs = [make_stereo(90 + i, frames, rate, level=0.3 + 0.6 * i) for i in range(2)]
    xs[1][2 * 5000:2 * 5000 + 2 * 64] *= np.float32(3.0)             # an inter-sample-peak prone burst
    b = ssa.Batch(rate, 2, 2, frames, 4096, 1024, flags=L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    taps = interpolator_taps(factor)
    for i, x in enumerate(xs):
        tp, sp = b.peaks(i)
        for c in range(2):
            ch = x[c::2].astype(np.float64)
            up = signal.upfirdn(taps, ch, up=factor)               # zero-stuffed input through the 49 taps: every phase
            want = max(np.abs(up[:factor * ch.size]).max(), np.abs(ch).max())
            assert abs(tp[c] - want) <= 2e-6 * want, (i, c, tp[c], want)
            assert sp[c] == np.abs(x[c::2]).max()


def lra_f64(x, rate, channels, coeffs):
    """EBU Tech 3342 loudness range with exact short-term values (3 s window every 1 s, the crate's cadence) instead of
    the 0.1 LU histogram: absolute gate -70 LUFS, relative gate -20 LU under the gated power mean, 10th / 95th percentile
    entries at ranks floor((n - 1) p + 0.5)."""
    b, a = coeffs
    y = signal.lfilter(b, a, x.reshape(-1, channels).astype(np.float64), axis=0)
    s100 = (rate + 5) // 10
    n = y.shape[0] // s100
    sub = np.array([np.sum(y[k * s100:(k + 1) * s100] ** 2) for k in range(n)])
    st = np.array([sub[k - 29:k + 1].sum() / (30.0 * s100) for k in range(29, n, 10)])
    st = st[10.0 * np.log10(np.maximum(st, 1e-300)) - 0.691 >= -70.0]
    if st.size == 0:
        return 0.0
    st = np.sort(st[st >= st.mean() * 0.01])
    lo = st[int((st.size - 1) * 0.10 + 0.5)]
    hi = st[int((st.size - 1) * 0.95 + 0.5)]
    return 10.0 * np.log10(hi / lo)


@pytest.mark.parametrize("rate", [44100, 48000, 96000])
def test_loudness_range_against_exact_f64_percentiles(oracle, rate):
    """Programme-like material: 40 s whose level moves over 25 dB in steps and ramps; histogram mode quantises each
    short-term value to 0.1 LU, so the device may differ from the exact percentiles by up to 0.1 LU at either end."""
    frames = rate * 40
    rng = np.random.default_rng(5)
    xs = []
    for i in range(2):
        x = make_stereo(120 + i, frames, rate, level=0.9).reshape(-1, 2)
        t = np.arange(frames) / rate
        gain_db = -25.0 + 12.0 * np.sin(2 * np.pi * t / (13.0 + 4 * i)) + 6.0 * np.floor(t / 7.0) % 3 + rng.uniform(-1, 1)
        xs.append((x * (10.0 ** (gain_db / 20.0))[:, None]).astype(np.float32).reshape(-1))
    b = ssa.Batch(rate, 2, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    coeffs = oracle.Meter(2, rate).coeffs()
    for i, x in enumerate(xs):
        want = lra_f64(x, rate, 2, coeffs)
        assert want > 5.0                                   # the material really has a range
        assert abs(b.results()[i].loudness_range - want) <= 0.21, (i, b.results()[i].loudness_range, want)


@pytest.mark.parametrize("rate,slice_frames", [(48000, 8192), (44100, 3000), (96000, 16384), (192000, 9999)])
def test_streaming_shortterm_and_momentary_against_f64_windows(oracle, rate, slice_frames):
    """The streaming handle (the twelve-method mirror of `Analyzer`): after every `add_samples` the short-term and
    momentary readings are -0.691 + 10 log10 of the mean square of the K-weighted signal over the last 3 s / 400 ms
    (zeros before the start) — no histogram in between, so the bar is tight: 1e-6 LU against a sequential f64 filter."""
    frames = int(rate * 4.3)
    x = make_stereo(150, frames, rate, level=0.4)
    bq, aq = oracle.Meter(2, rate).coeffs()
    y = signal.lfilter(bq, aq, x.reshape(-1, 2).astype(np.float64), axis=0)
    csum = np.concatenate([[0.0], np.cumsum((y ** 2).sum(axis=1))])
    a = ssa.Analyzer(2, rate)
    fed = 0
    while fed < frames:
        n = min(slice_frames, frames - fed)
        a.add_samples(x[2 * fed:2 * (fed + n)])
        fed += n
        for win, got in ((3.0, a.get_shortterm_lufs()), (0.4, a.get_momentary_lufs())):
            w = int(round(win * rate))
            if win == 3.0:
                w = ((w + (rate + 5) // 10 - 1) // ((rate + 5) // 10)) * ((rate + 5) // 10)    # ring length: a multiple of 100 ms
            lo = max(fed - w, 0)
            e = (csum[fed] - csum[lo]) / w
            want = -np.inf if e <= 0 else 10.0 * np.log10(e) - 0.691
            assert abs(got - want) <= 1e-6, (fed, win, got, want)
    a.close()


def waveform_numpy(x, window_s):
    """analyzer.rs:107-137 with numpy: W = window_s * 1000 bins, spp = len / W in f64, bin i = [floor(i spp),
    min(ceil((i + 1) spp), len)), points (i, min), (i, max); stops at the first bin that starts past the end."""
    w = int(window_s * 1000.0)
    spp = len(x) / w
    out = []
    for i in range(w):
        bs = int(np.floor(i * spp))
        be = min(int(np.ceil((i + 1) * spp)), len(x))
        if bs >= len(x):
            break
        seg = x[bs:be]
        out += [np.nanmin(seg) if seg.size and not np.all(np.isnan(seg)) else (np.nan if seg.size else 0.0),
                np.nanmax(seg) if seg.size and not np.all(np.isnan(seg)) else (np.nan if seg.size else 0.0)]
    return np.array(out, np.float32)


@pytest.mark.parametrize("rate,channels,seconds", [(48000, 2, 2.0), (44100, 2, 2.0), (96000, 8, 1.0), (48000, 1, 0.0294), (22050, 2, 3.7)])
def test_decimation_against_numpy(rate, channels, seconds):
    """Min-max decimation, fused in the batch kernel or standalone, and the handle-less `get_waveform`: bit for bit
    against the definition restated with numpy (integer and fractional samples per bin)."""
    frames = max(int(rate * seconds), 4)
    rng = np.random.default_rng(9)
    xs = [(rng.standard_normal(frames * channels) * 0.2).astype(np.float32) for _ in range(2)]
    b = ssa.Batch(rate, channels, 2, frames, 4096, 1024, flags=L.SS_BATCH_WAVEFORM | L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    for i, x in enumerate(xs):
        want = waveform_numpy(x, frames / rate)
        got = b.waveform(i).reshape(-1)[:want.size]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
        pts = ssa.Analyzer.get_waveform(x, frames / rate)
        assert np.array_equal(np.asarray(pts)[:, 1].astype(np.float32).view(np.uint32), want.view(np.uint32)), i


@pytest.mark.parametrize("rate,n", [(48000, 1024), (48000, 2048), (44100, 4096), (48000, 8192), (96000, 16384), (48000, 32768)])
def test_single_window_get_fft_against_numpy_f64(rate, n):
    """`Analyzer::get_fft` (one mono window of any power-of-two length: the generic, the 4096- and the 16384-point kernels)
    against the numpy restatement; the x axis of the pairs is the log-frequency chart position (analyzer.rs:88-98)."""
    rng = np.random.default_rng(n)
    t = np.arange(n) / rate
    x = (0.4 * np.sin(2 * np.pi * 1234.5 * t) + 0.1 * np.sin(2 * np.pi * 77.0 * t) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    a = ssa.Analyzer(2, rate)
    got = np.asarray(a.get_fft(x))
    a.close()
    ref = spectrum_f64(x, rate, n)
    assert got.shape == (ref.size, 2)
    assert db_close(got[:, 1], ref, 0.01), db_report(got[:, 1], ref)
    freq32 = np.arange(n // 2 + 1, dtype=np.float32) * (np.float32(rate) / np.float32(n))
    keep = (freq32 >= 20.0) & (freq32 <= 20000.0)
    xpos = (np.log10(freq32[keep].astype(np.float64)) - np.log10(20.0)) / (np.log10(20000.0) - np.log10(20.0)) * 100.0
    assert np.abs(got[:, 0] - xpos).max() <= 1e-9
