"""How far down do the spectrum kernels follow the oracle?  The synthetic corpus carries 0.05 * U(-1, 1) noise, so none of
its bins lies more than ~60 dB under its window's loudest bin; here the inputs are chosen to expose the floor:
  * pure tones WITHOUT noise, bin-centred and off-bin, at N = 4096 and 16384 (Hann side lobes down to the rounding noise);
  * a programme scaled by 2^-20 and 2^-40 (exact scaling: every dB value must move by exactly k * 6.0206 dB);
  * windows of a near-silent passage (x 1e-4) held to the same peak-relative bar as loud ones;
  * narrow stereo (side 20 .. 120 dB under mid, and the other way round), side-only onsets, level steps, fades and onsets
    between the two windows of k_fft4096_pairw — every row against its OWN peak (block exponents);
  * dual mono (L == R, L == -R) through the packed mid/side kernels.
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


def _programme(rng, frames, rate, f0, amp=0.4, noise=0.05):
    t = np.arange(frames) / rate
    return amp * np.sin(2 * np.pi * f0 * t) + noise * rng.uniform(-1, 1, frames)


def _stereo_from_mid_side(m, s):
    x = np.empty(2 * len(m), np.float32)
    x[0::2] = (m + s).astype(np.float32); x[1::2] = (m - s).astype(np.float32)
    return x


def _check_every_row(oracle, x, rate, channels, n, hop, tag):
    """Every window row of the batch spectrum against the oracle at the plain per-row bar (its OWN peak); returns the worst
    (|d| within 70 dB of the row peak, linear error below) per row index."""
    frames = len(x) // channels
    b = ssa.Batch(rate, channels, 1, frames, n, hop, flags=L.SS_BATCH_FFT)
    b.upload(0, x); b.run(); b.sync()
    got = b.fft(0)
    if channels == 2:
        ref = oracle.analyze_stream(rate, x, n, hop, want_wave=False)["fft"]
    else:
        lay = b.layout
        xc = x.reshape(-1, channels)
        ref = np.empty_like(got, dtype=np.float64)
        for w in range(lay.n_windows):
            p0 = (w + 1) * hop                                  # the reference skips the window whose left bound is 0 (tui.rs:1489)
            for c in range(channels):
                ref[w, c] = oracle.get_fft(rate, np.ascontiguousarray(xc[p0:p0 + n, c]))[:, 1]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    worst = np.zeros((got.shape[1], 2))
    for w in range(got.shape[0]):
        for c in range(got.shape[1]):
            assert db_close(got[w, c], ref[w, c]), (tag, n, hop, w, c, db_report(got[w, c], ref[w, c]))
            worst[c] = np.maximum(worst[c], db_report(got[w, c], ref[w, c]))
    b.close()
    return worst


@pytest.mark.parametrize("n,hop", [(4096, 1024), (4096, 512), (4096, 2048), (4096, 768), (16384, 1024)])
@pytest.mark.parametrize("under_db", [20, 40, 60, 120, -40])
def test_narrow_stereo_rows_keep_their_own_bar(oracle, n, hop, under_db):
    """Mid and side of the packed N = 4096 kernels ride ONE complex transform.  The side signal `under_db` under the mid
    signal (-40: the MID signal 40 dB under side, L ~ -R) is held to the plain bar against its OWN loudest bin, every window —
    the reference transforms each signal on its own (tui.rs:1505,1515 -> analyzer.rs:55-65).  The kernels give the weaker row
    a power-of-two block exponent per window: hop 1024 from the hop levels (k_fft4096_ms1), the other
    hops from the exact windowed levels (k_fft4096_ms<2>, <8>, _anyhop); N = 16384 transforms each signal alone."""
    rate, frames = 48000, n + hop * 14
    rng = np.random.default_rng(9)
    m = _programme(rng, frames, rate, 523.0)
    s = _programme(rng, frames, rate, 1777.0) * 10.0 ** (-abs(under_db) / 20.0)
    if under_db < 0:
        m, s = s, m
    worst = _check_every_row(oracle, _stereo_from_mid_side(m, s), rate, 2, n, hop, f"under {under_db}")
    print(f"\nN={n} hop={hop} side {under_db} dB under mid: mid row worst {worst[0][0]:.5f} dB / {worst[0][1]:.1e}, side row worst {worst[1][0]:.5f} dB / {worst[1][1]:.1e}")


@pytest.mark.parametrize("hop", [1024, 512])
def test_narrow_stereo_pure_tones(oracle, hop):
    """The same without noise: both rows are pure tones, so each has bins all the way down to 70 dB under its own peak (Hann
    side lobes), the side row 40 dB under the mid row."""
    rate, n = 48000, 4096
    frames = n + hop * 12
    t = np.arange(frames) / rate
    m = 0.5 * np.sin(2 * np.pi * (40.37 * rate / n) * t + 0.3)
    s = 0.005 * np.sin(2 * np.pi * (151.41 * rate / n) * t + 1.1)
    worst = _check_every_row(oracle, _stereo_from_mid_side(m, s), rate, 2, n, hop, "pure tones")
    print(f"\nhop={hop} pure tones, side 40 dB under mid: mid {worst[0][0]:.5f} dB / {worst[0][1]:.1e}, side {worst[1][0]:.5f} dB / {worst[1][1]:.1e}")


@pytest.mark.parametrize("onset", [1024 * 9 + 1000, 1024 * 9 + 700, 1024 * 9 + 40, 1024 * 10 - 3])
def test_side_only_onset_inside_a_window(oracle, onset):
    """A wide (side-only) element that starts abruptly while the centre plays on: for one window the side row's energy sits
    in the last few samples of the window, under Hann weights near zero — its windowed level is far below what its hop's raw
    level says.  k_fft4096_ms1 takes the exact path there (largest |sample x weight| of either row).  The decay is the mirror
    case: the side element stops abruptly, and its tail leaves through the first samples of later windows."""
    rate, n = 48000, 4096
    frames = n + 1024 * 24
    rng = np.random.default_rng(21)
    m = _programme(rng, frames, rate, 440.0)
    s = _programme(rng, frames, rate, 2500.0, amp=0.3, noise=0.02)
    s[:onset] = 0.0                                      # digital silence in the side signal up to the onset ...
    s[onset + 1024 * 9 + 511:] *= 1e-5                   # ... and an abrupt drop by 100 dB later on
    _check_every_row(oracle, _stereo_from_mid_side(m, s), rate, 2, n, 1024, f"onset {onset}")


def test_side_level_steps_both_ways(oracle):
    """The block exponent follows the programme: the side signal drops by 50 dB, comes back, and finally exceeds the mid signal by
    30 dB (the exponent changes sign); every window of both rows at the plain bar."""
    rate, n = 48000, 4096
    frames = n + 1024 * 60
    rng = np.random.default_rng(33)
    m = _programme(rng, frames, rate, 700.0)
    s = _programme(rng, frames, rate, 3100.0) * 0.1
    q = frames // 4
    s[q:2 * q] *= 10.0 ** (-50 / 20)
    m[3 * q:] *= 10.0 ** (-50 / 20)
    _check_every_row(oracle, _stereo_from_mid_side(m, s), rate, 2, n, 1024, "steps")


@pytest.mark.parametrize("channels", [1, 6])
@pytest.mark.parametrize("kind", ["fade_in", "fade_out", "onset", "steps"])
def test_pair_kernel_windows_at_different_levels(oracle, channels, kind):
    """k_fft4096_pairw rides two CONSECUTIVE windows of one channel on one transform: a fade (2.5 dB per hop), an abrupt onset
    at an arbitrary sample and 40 dB level steps put the two windows of a pair at different levels.  Every window at the plain
    bar against its own peak (the second window of a pair carries a block exponent)."""
    rate, n = 48000, 4096
    frames = n + 1024 * 41 + 17
    rng = np.random.default_rng(55)
    x = np.empty((frames, channels), np.float32)
    hops = np.arange(frames) / 1024.0
    for c in range(channels):
        y = _programme(rng, frames, rate, 300.0 + 170.0 * c, amp=0.3, noise=0.03)
        if kind == "fade_in":
            y *= 10.0 ** (np.minimum(0.0, -90.0 + 2.5 * hops) / 20.0)
        elif kind == "fade_out":
            y *= 10.0 ** (np.minimum(0.0, -2.5 * (hops - 8.0 - c)) / 20.0)
        elif kind == "onset":
            y[:1024 * 11 + 333 * c + 5] *= 1e-6
        else:
            for k in range(0, frames, 1024 * 7):
                y[k:k + 1024 * 3 + 100 * c] *= 0.01
        x[:, c] = y.astype(np.float32)
    _check_every_row(oracle, x.reshape(-1), rate, channels, n, 1024, kind)


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


@pytest.mark.parametrize("seed", range(10))
def test_random_level_programmes_every_row_at_its_own_bar(oracle, seed):
    """Randomised: mid and side programmes whose levels jump independently between 0 and -140 dB (and to exact digital silence)
    at random sample positions, random hops of the packed N = 4096 kernels — every row of every window against the oracle at the
    plain bar.  Exercises the ordinary hop-level path, the exact path at the jumps, the rescale in both directions, empty rows
    next to loud ones and the sign change of the block exponent in one run."""
    rng = np.random.default_rng(1000 + seed)
    rate, n = 48000, 4096
    hop = int(rng.choice([1024, 1024, 1024, 512, 2048, 768]))
    frames = n + hop * int(rng.integers(20, 40)) + int(rng.integers(0, hop))

    def programme(f0):
        y = _programme(rng, frames, rate, f0, amp=0.35, noise=0.04)
        pos = 0
        while pos < frames:
            seg = int(rng.integers(300, 6000))
            r = rng.uniform()
            g = 0.0 if r < 0.12 else 10.0 ** (-rng.uniform(0, 140) / 20.0)
            y[pos:pos + seg] *= g
            pos += seg
        return y

    m, sd = programme(float(rng.uniform(100, 4000))), programme(float(rng.uniform(100, 8000)))
    _check_every_row(oracle, _stereo_from_mid_side(m, sd), rate, 2, n, hop, f"random {seed} hop {hop}")


@pytest.mark.parametrize("seed", range(4))
def test_random_level_programmes_pair_kernel(oracle, seed):
    """The same for k_fft4096_pairw (mono and 3 channels): consecutive windows of one channel at random levels."""
    rng = np.random.default_rng(2000 + seed)
    rate, n = 48000, 4096
    channels = 1 if seed % 2 == 0 else 3
    frames = n + 1024 * int(rng.integers(24, 40)) + int(rng.integers(0, 1024))
    x = np.empty((frames, channels), np.float32)
    for c in range(channels):
        y = _programme(rng, frames, rate, float(rng.uniform(100, 6000)), amp=0.3, noise=0.03)
        pos = 0
        while pos < frames:
            seg = int(rng.integers(200, 5000))
            r = rng.uniform()
            y[pos:pos + seg] *= 0.0 if r < 0.12 else 10.0 ** (-rng.uniform(0, 140) / 20.0)
            pos += seg
        x[:, c] = y.astype(np.float32)
    _check_every_row(oracle, x.reshape(-1), rate, channels, n, 1024, f"random pair {seed}")
