"""How far down do the spectrum kernels follow the oracle?  The synthetic corpus carries 0.05 * U(-1, 1) noise, so none of
its bins lies more than ~60 dB under its window's loudest bin; here the inputs are chosen to expose the floor:
  * pure tones WITHOUT noise, bin-centred and off-bin, at N = 4096 and 16384 (Hann side lobes down to the rounding noise);
  * a programme scaled by 2^-20 and 2^-40 (exact scaling: every dB value must move by exactly k * 6.0206 dB);
  * windows of a near-silent passage (x 1e-4) held to the same peak-relative bar as loud ones;
  * narrow stereo (side 40 dB under mid) and dual mono (L == R, L == -R) through the packed mid/side kernels.
Bar (conftest.db_close): 0.01 dB for every bin within 70 dB of its row's loudest bin, 1e-4 of that bin's amplitude below.
"""
import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import db_close, db_report, make_stereo

pytestmark = pytest.mark.gpu


def _tone(n_frames, rate, freq, amp=0.5, phase=0.3):
    t = np.arange(n_frames, dtype=np.float64) / rate
    return (amp * np.sin(2 * np.pi * freq * t + phase)).astype(np.float32)


@pytest.mark.parametrize("n", [4096, 16384])
@pytest.mark.parametrize("kind", ["bin_centred", "off_bin"])
def test_pure_tone_without_noise_single_window(oracle, n, kind):
    """Analyzer::get_fft on a noise-free tone: the k_fft4096 / k_fft16k single-window paths against the oracle's radix-2."""
    rate = 48000
    for k0 in (40, 997):
        f = (k0 + (0.0 if kind == "bin_centred" else 0.37)) * rate / n
        x = _tone(n, rate, f)
        a = ssa.Analyzer(2, rate)
        got = np.asarray(a.get_fft(x))[:, 1]
        a.close()
        ref = oracle.get_fft(rate, x)[:, 1]
        assert db_close(got, ref), (n, kind, k0, db_report(got, ref))
        assert abs(got.max() - ref.max()) <= 0.001                      # the tone's own bin: far inside the bar


@pytest.mark.parametrize("n,channels", [(4096, 2), (16384, 2), (16384, 8), (4096, 1)])
@pytest.mark.parametrize("kind", ["bin_centred", "off_bin"])
def test_pure_tone_without_noise_batch_kernels(oracle, n, channels, kind):
    """The batch kernels (k_fft4096_ms1: mid/side packed; k_fft16k_run: mid/side and per channel; k_fft4096_pairw: two
    windows packed) on noise-free tones, every window of the run."""
    rate = 48000 if channels <= 2 else 96000
    frames = n + 1024 * 40
    chans = []
    for c in range(channels):
        k0 = 33 + 61 * c
        f = (k0 + (0.0 if kind == "bin_centred" else 0.41 + 0.05 * c)) * rate / n
        chans.append(_tone(frames, rate, f, amp=0.6 / (1 + c), phase=0.2 * c))
    x = np.stack(chans, axis=1).reshape(-1)
    b = ssa.Batch(rate, channels, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
    b.upload(0, x); b.run(); b.sync()
    got = b.fft(0)
    lay = b.layout
    assert got.shape[0] == lay.n_windows >= 30
    if channels == 2:
        lr = x.reshape(-1, 2)
        sigs = [((lr[:, 0] + lr[:, 1]) / np.float32(2)).astype(np.float32), ((lr[:, 0] - lr[:, 1]) / np.float32(2)).astype(np.float32)]
    else:
        sigs = [np.ascontiguousarray(x.reshape(-1, channels)[:, c]) for c in range(channels)]
    worst = (0.0, 0.0)
    for w in range(0, lay.n_windows, 3):
        p = (w + n // 1024 + 1) * 1024
        for c, sig in enumerate(sigs):
            ref = oracle.get_fft(rate, sig[p - n:p])[:, 1]
            assert db_close(got[w, c], ref), (n, channels, kind, w, c, db_report(got[w, c], ref))
            r = db_report(got[w, c], ref)
            worst = (max(worst[0], r[0]), max(worst[1], r[1]))
    print(f"\\nN={n} ch={channels} {kind}: worst |d| within 70 dB of the row peak {worst[0]:.5f} dB, worst linear error below {worst[1]:.2e} of the peak")


@pytest.mark.parametrize("n", [4096, 16384])
def test_scaled_programme_shifts_by_exactly_6db_per_bit(oracle, n):
    """Scaling the input by 2^-k is exact in f32, and every operation of the transform scales with it: the dB values must
    move by k * 20 log10(2) — on the device to 1e-3 dB, and against the oracle of the scaled input at the usual bar."""
    rate, frames = 48000, n + 1024 * 12
    x = make_stereo(77, frames, rate, level=0.5)
    outs = {}
    for k in (0, 20, 40):
        xs = (x * np.float32(2.0 ** -k)).astype(np.float32)
        assert np.array_equal(xs.astype(np.float64) * 2.0 ** k, x.astype(np.float64))          # exact scaling, no sub-normals
        b = ssa.Batch(rate, 2, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
        b.upload(0, xs); b.run(); b.sync()
        outs[k] = b.fft(0).astype(np.float64)
        ref = oracle.analyze_stream(rate, xs, n, 1024, want_wave=False)["fft"]
        for w in range(outs[k].shape[0]):
            for c in range(2):
                assert db_close(outs[k][w, c], ref[w, c]), (n, k, w, c, db_report(outs[k][w, c], ref[w, c]))
        b.close()
    step = 20.0 * np.log10(2.0)
    for k in (20, 40):
        d = outs[0] - outs[k] - k * step
        assert np.abs(d).max() <= 1e-3, (n, k, float(np.abs(d).max()))


def test_near_silent_passage_at_the_peak_relative_bar(oracle):
    """The corpus' gap streams: three seconds at x 1e-4.  Every window inside the gap — all bins under -90 dBFS — is held to
    0.01 dB down to 70 dB under ITS OWN loudest bin (an absolute -90 dB floor would wave these windows through)."""
    rate, frames = 48000, 48000 * 6
    x = make_stereo(5, frames, rate, level=0.5, gap=True)           # gap = frames [frames/3, frames/3 + 3 s)
    for n in (4096, 16384):
        b = ssa.Batch(rate, 2, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
        b.upload(0, x); b.run(); b.sync()
        got = b.fft(0)
        ref = oracle.analyze_stream(rate, x, n, 1024, want_wave=False)["fft"]
        quiet = 0
        for w in range(got.shape[0]):
            for c in range(2):
                assert db_close(got[w, c], ref[w, c]), (n, w, c, db_report(got[w, c], ref[w, c]))
            quiet += ref[w].max() < -90.0
        assert quiet >= 60                                             # the gap's windows were really among them
        b.close()


@pytest.mark.parametrize("n", [4096, 16384])
def test_narrow_stereo_rows_share_one_transform(oracle, n):
    """Mid and side of the packed kernels ride ONE complex transform, so the weaker row's rounding noise is set by the
    stronger row: with the side signal 40 dB under the mid signal, the side row is held to 0.01 dB down to 70 dB under the
    PAIR's loudest bin (= 30 dB under its own), and its own-peak figures are printed (DESIGN section 6 quotes them)."""
    rate, frames = 48000, n + 1024 * 10
    rng = np.random.default_rng(9)
    t = np.arange(frames) / rate
    m = (0.4 * np.sin(2 * np.pi * 523.0 * t) + 0.05 * rng.uniform(-1, 1, frames)).astype(np.float32)
    s = (0.01 * (0.4 * np.sin(2 * np.pi * 1777.0 * t) + 0.05 * rng.uniform(-1, 1, frames))).astype(np.float32)
    x = np.empty(2 * frames, np.float32); x[0::2] = m + s; x[1::2] = m - s
    b = ssa.Batch(rate, 2, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
    b.upload(0, x); b.run(); b.sync()
    got = b.fft(0)
    ref = oracle.analyze_stream(rate, x, n, 1024, want_wave=False)["fft"]
    own = (0.0, 0.0)
    for w in range(got.shape[0]):
        pair_peak = float(ref[w].max())
        assert db_close(got[w, 0], ref[w, 0]), (n, w, db_report(got[w, 0], ref[w, 0]))
        assert db_close(got[w, 1], ref[w, 1], peak_db=pair_peak), (n, w)
        r = db_report(got[w, 1], ref[w, 1])
        own = (max(own[0], r[0]), max(own[1], r[1]))
    print(f"\\nN={n}: side row 40 dB under mid, against its OWN peak: worst |d| within 70 dB {own[0]:.4f} dB, linear below {own[1]:.2e}")
    b.close()


@pytest.mark.parametrize("n", [4096, 16384])
@pytest.mark.parametrize("which", ["L == R", "L == -R"])
def test_dual_mono_empty_row_reads_the_floor(oracle, n, which):
    """A stereo file with identical channels has an exactly-zero side signal (mid for L == -R): the reference transforms a
    buffer of zeros there and reports its floor, -150 dB (+ pink compensation), in every bin (analyzer.rs:20-22)."""
    rate, frames = 48000, n + 1024 * 8
    m = make_stereo(3, frames, rate, level=0.6)[0::2].copy()
    x = np.empty(2 * frames, np.float32); x[0::2] = m; x[1::2] = m if which == "L == R" else -m
    b = ssa.Batch(rate, 2, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
    b.upload(0, x); b.run(); b.sync()
    got = b.fft(0)
    ref = oracle.analyze_stream(rate, x, n, 1024, want_wave=False)["fft"]
    empty = 1 if which == "L == R" else 0
    _, _, pink = b.bin_tables()
    for w in range(got.shape[0]):
        assert np.allclose(ref[w, empty], -150.0 + pink, atol=1e-4)                          # the oracle's floor row
        assert np.allclose(got[w, empty], ref[w, empty], rtol=0.0, atol=2e-4), (n, which, w, float(got[w, empty].max()))
        assert db_close(got[w, 1 - empty], ref[w, 1 - empty]), (n, which, w)
    b.close()


@pytest.mark.parametrize("channels", [1, 6])
def test_digital_silence_in_front_of_a_programme_pair_kernel(oracle, channels):
    """k_fft4096_pairw rides TWO consecutive windows of one channel on one complex transform.  Digital silence followed by a
    programme: the last all-zero window shares its transform with the first window that touches the programme, and must
    still read the reference's floor (-150 dB + pink) in every bin, not the partner's rounding noise."""
    rate, n = 48000, 4096
    frames = 21504 + 1024 * 30 + 100
    rng = np.random.default_rng(4)
    x = np.zeros((frames, channels), np.float32)
    for c in range(channels):
        start = 21504 - 1024 * (c % 2)                     # odd channels: the boundary falls between the windows of a pair
        t = np.arange(frames - start) / rate
        x[start:, c] = (0.3 * np.sin(2 * np.pi * (300.0 + 170.0 * c) * t) + 0.03 * rng.uniform(-1, 1, frames - start)).astype(np.float32)
    b = ssa.Batch(rate, channels, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
    b.upload(0, x.reshape(-1)); b.run(); b.sync()
    got = b.fft(0)
    _, _, pink = b.bin_tables()
    n_floor = 0
    for w in range(got.shape[0]):
        p = (w + n // 1024 + 1) * 1024
        for c in range(channels):
            seg = np.ascontiguousarray(x[p - n:p, c])
            ref = oracle.get_fft(rate, seg)[:, 1]
            if not seg.any():
                assert np.allclose(ref, -150.0 + pink, atol=1e-4)
                assert np.allclose(got[w, c], ref, rtol=0.0, atol=2e-4), (channels, w, c, float(got[w, c].max()))
                n_floor += 1
            else:
                assert db_close(got[w, c], ref), (channels, w, c, db_report(got[w, c], ref))
    assert n_floor >= 16 * channels
    b.close()
