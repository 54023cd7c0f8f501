"""Columns-only spectrum (SS_BATCH_FFT_COLUMNS): the render-side reduction of SURVEY 8f N3 — the reference's gain
(tui.rs:801-821, :1229-1238), its chart bounds [-100, 0] dB (tui.rs:49-51) and the library's column rule — fused into the
spectrum kernel's epilogue, so that the full rows are never stored.  Checked bit for bit against the two-pass path
(ss_batch_run with rows + ss_batch_render_spectrum) and against the restatement oracle/render.py on the oracle's rows."""
import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("cols", [1, 64, 160, 512])
@pytest.mark.parametrize("gain", [None, 0.0, 7.5, -40.0])
def test_fused_columns_equal_the_two_pass_result(cols, gain):
    rate, frames, ns = 48000, 48000 * 3 + 333, 5
    xs = [make_stereo(300 + s, frames, rate=rate, level=0.05 + 0.2 * s, gap=(s == 2)) for s in range(ns)]
    xs[4][1::2] = xs[4][0::2]                                    # a dual-mono stream: its side row is the -150 dB floor
    xs[3][2 * 70000] = np.nan                                    # four windows whose every bin is NaN: -100 in every column that owns a bin
    xs[1][2 * 30000 + 1] = np.inf
    two = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    two.upload(0, np.concatenate(xs)); two.run(); two.render_spectrum(cols, gain)
    one = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
    one.upload(0, np.concatenate(xs))
    one.set_columns_gain(gain)
    for rep in range(2):                                         # the LDS accumulators are reset between windows AND passes
        one.run(); one.sync()
        for s in range(ns):
            a, b = one.spectrum_columns(s), two.spectrum_columns(s)
            assert a.shape == b.shape == (two.layout.n_windows, 2, cols)
            assert _same(a, b), (cols, gain, s, rep, int(np.sum(~((a == b) | (np.isnan(a) & np.isnan(b))))))
    # everything else of the pass is unchanged by the mode
    r1, r2 = one.results(), two.results()
    for s in range(ns):
        assert r1[s].integrated_lufs == r2[s].integrated_lufs and r1[s].true_peak[0] == r2[s].true_peak[0]
        assert np.array_equal(one.waveform(s), two.waveform(s), equal_nan=True)
    with pytest.raises(ssa.AnalyzerError):
        one.fft(0)                                               # the rows do not exist in this mode
    one.close(); two.close()


@pytest.mark.parametrize("seed", [87, 5, 140, 263])
def test_randomised_columns_programme(seed):
    """tools/fuzz_columns.py: random rate / column count / gain / stream count / length with dual-mono, silent, NaN and infinite
    material, fused columns against the two-pass result bit for bit.  Seed 87 (88.2 kHz, a dual-mono stream) is the first run's
    finding: the floor row's columns were computed with another expression than the floor row itself (2 ulp apart at that rate)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_columns
    ok, msg = fuzz_columns.programme(seed)
    assert ok, msg


def test_fused_columns_match_the_restatement(oracle):
    from oracle import render as R
    rate, frames, cols = 48000, 48000 * 3, 160
    x = make_stereo(77, frames, rate=rate, level=0.3)
    b = ssa.Batch(rate, 2, 1, frames, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
    b.upload(0, x); b.run(); b.sync()
    got = b.spectrum_columns(0)
    ref = oracle.analyze_stream(rate, x, 4096, 1024)
    g = R.gain_db(ref["integrated"])
    chart_x, _, _ = b.bin_tables()
    for w in range(0, got.shape[0], 7):
        for ch in (0, 1):
            want = R.spectrum_columns(np.stack([chart_x, ref["fft"][w, ch].astype(np.float64)], 1), g, cols)
            assert np.array_equal(np.isnan(got[w, ch]), np.isnan(want))
            ok = ~np.isnan(want)
            assert np.abs(got[w, ch][ok] - want[ok]).max() <= 0.011      # 0.01 dB spectrum bar + the gain's f32 addition
    b.close()


def test_fused_columns_at_the_bench_shape_and_modes():
    """1024 streams x 10 s (the bench geometry): fused columns equal the two-pass result for a strided sample of streams;
    shapes the fused mode does not cover are refused."""
    rate, frames, ns, cols = 48000, 480000, 1024, 160
    one = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
    one.synthesize(0x5EED0000, 0); one.run(); one.sync()
    assert one.layout.fft_bytes == ns * 464 * 2 * cols * 4
    picks = [0, 1, 255, 511, 512, 1023]
    keep = {s: one.spectrum_columns(s).copy() for s in picks}
    one.close()
    two = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    two.synthesize(0x5EED0000, 0); two.run(); two.render_spectrum(cols, None)
    for s in picks:
        assert _same(keep[s], two.spectrum_columns(s)), s
    two.close()
    for kw in (dict(channels=1), dict(fft_n=16384), dict(hop_frames=512)):
        args = dict(sample_rate=rate, channels=2, n_streams=1, frames_per_stream=48000 * 2, fft_n=4096, hop_frames=1024,
                    flags=L.SS_BATCH_FFT, spectrum_columns=64)
        args.update(kw)
        with pytest.raises(ssa.AnalyzerError):
            ssa.Batch(**args)
