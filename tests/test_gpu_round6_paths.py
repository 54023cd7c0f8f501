"""The two forms k_time_domain got in round 6, at their edges, through the C ABI against the oracle:

* the 4x (and, in the three-waves register builds, the 2x) true peak on the packed-f32 VALU (2 / 6 / 8 channels): a lane takes fifteen
  frames of one channel pair (at factor 2: two neighbouring lanes, a half of the 24-tap branch each), so the tile lengths
  that are no multiple of fifteen (the last lane takes the tile's last fifteen frames again), tiles shorter than fifteen frames (the
  crate's own loop) and streaming calls of every small size are the cases;
* the same with plain v_fma_f32 over ONE channel per lane, for the channel counts that do not divide sixteen (3, 5, 7, 12 ...) and for
  factor 2 in the four-waves register builds;
* the min-max decimation with ANY samples per bin (44.1 kHz material, odd lengths), read as the aligned 16-byte pieces that overlap
  a bin with the edge pieces masked: bins starting and ending at every alignment, the shortest bins the fused path takes (16
  samples) and the longest (1000), special values sitting exactly on bin edges.

Bars: decimation bit for bit; true peak within 2e-6 relative (an f32 fma chain against the crate's multiply-then-add, both 1e-7 from
an f64 convolution; north_star's bar is 1e-4), sample peak exactly, LUFS within 0.01."""
import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_multich, make_stereo

pytestmark = pytest.mark.gpu


def _signal(seed, frames, channels, rate):
    x = make_stereo(seed, frames, rate, level=0.6) if channels == 2 else make_multich(seed, frames, channels, rate, level=0.6)
    # a few isolated full-scale clicks: inter-sample peaks that one lane's fifteen frames own alone
    rng = np.random.default_rng(seed)
    for f in rng.integers(30, frames - 30, size=12):
        c = int(rng.integers(0, channels))
        x[channels * int(f) + c] = 0.99 * (1 if f & 1 else -1)
        x[channels * (int(f) + 1) + c] = 0.97 * (1 if f & 1 else -1)
    return x


@pytest.mark.parametrize("rate,channels", [(48000, 2), (44100, 2), (8000, 2), (32000, 2), (22050, 2), (11025, 2), (48000, 6), (44100, 6),
                                           (8000, 6), (48000, 8), (32000, 8), (22050, 8),
                                           (96000, 2), (96000, 6), (96000, 8), (176400, 2), (128000, 6),       # factor 2: the branch's halves on lane pairs
                                           (48000, 3), (44100, 5), (8000, 7), (96000, 3), (48000, 12),        # plain-FMA form: counts that do not divide 16
                                           (48000, 4), (48000, 1), (96000, 4)])                               # ... and the ones that stay on the matrix pipe
def test_packed_true_peak_tile_lengths(oracle, rate, channels):
    """Batches whose 100 ms sub-blocks cut into tiles of every length class: 960 and 1470 frames (multiples of fifteen), 800, 1103,
    2205 ... (not), over three streams of a length that leaves a short last tile.  Every channel's true and sample peak, the loudness."""
    frames = int(rate * 2.5) + 7                                   # the last tile of a stream is seven frames long (the crate's loop)
    xs = [_signal(40 + i, frames, channels, rate) for i in range(3)]
    b = ssa.Batch(rate, channels, 3, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM)
    assert b.geometry.td_true_peak_factor == (4 if rate < 96000 else 2) and b.true_peak_arith == L.SS_TP_ARITH_F32
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    for i, x in enumerate(xs):
        m = oracle.Meter(channels, rate)
        m.add_frames(x)
        tp, sp = b.peaks(i)
        for c in range(channels):
            want = max(m.true_peak(c), m.sample_peak(c))
            assert abs(tp[c] - want) <= 2e-6 * want, (i, c, tp[c], want)
            assert sp[c] == m.sample_peak(c), (i, c)
        assert abs(res[i].integrated_lufs - m.integrated()) <= 0.01, i


@pytest.mark.parametrize("rate", [48000, 96000])
@pytest.mark.parametrize("channels", [2, 6, 8, 3, 5])
def test_packed_true_peak_streaming_call_sizes(oracle, channels, rate):
    """The handle's add_samples in calls of 1 ... 20 frames (tiles under fifteen frames: the crate's loop; fifteen and up: one lane
    of the packed form reading its history from the carried frames), then a few hundred, then a tick-sized call: the peak of a
    click moves through every position of a lane's window.  96 kHz: factor 2, a lane pair per fifteen frames."""
    frames = 3 * rate // 2
    x = _signal(70 + channels, frames, channels, rate)
    an = ssa.Analyzer(); an.create_loudness_meter(channels, rate)
    m = oracle.Meter(channels, rate)
    sizes = list(range(1, 21)) + [29, 30, 31, 45, 59, 61, 200, 481, 959, 960, 961, 4801, 8192]
    pos, k = 0, 0
    while pos < frames:
        n = min(sizes[k % len(sizes)], frames - pos); k += 1
        piece = x[channels * pos: channels * (pos + n)]
        an.add_samples(piece); m.add_frames(piece)
        pos += n
        if k % 7 == 0 or pos == frames:
            for c in range(channels):
                got, want = an.get_true_peak_channel(c), max(m.true_peak(c), m.sample_peak(c))
                assert abs(got - want) <= 2e-6 * max(want, 1e-30), (pos, c, got, want)
                assert an.get_sample_peak_channel(c) == m.sample_peak(c)
    assert abs(an.get_shortterm_lufs() - m.shortterm()) <= 0.01 or (np.isinf(m.shortterm()) and an.get_shortterm_lufs() == m.shortterm())


def _spiked(rng, n, spp):
    x = (rng.standard_normal(n) * 0.2).astype(np.float32)
    nb = int(n / spp)
    for i in rng.integers(1, max(nb - 1, 2), size=40):
        s, e = int(np.floor(i * spp)), int(np.ceil((i + 1) * spp))          # the reference's own bin bounds (analyzer.rs:107-137)
        kind = int(rng.integers(0, 6))
        if kind == 0: x[s] = np.nan                                         # first sample of a bin
        elif kind == 1: x[min(e, n) - 1] = np.nan                           # last sample of a bin (shared with the next when spp is fractional)
        elif kind == 2: x[s:min(e, n)] = np.nan                             # a bin of nothing but NaN
        elif kind == 3: x[s] = np.inf; x[min(e, n) - 1] = -np.inf
        elif kind == 4: x[s:min(s + 4, n)] = -0.0
        else: x[max(s - 1, 0)] = 7.0; x[min(e, n - 1)] = -7.0               # loud samples just OUTSIDE the bin: an edge piece must mask them
    return x


@pytest.mark.parametrize("rate,channels,frames,window", [
    (44100, 2, 88200, 2.0),        # 88.2 samples per bin
    (48000, 2, 47001, 1.0),        # 94.002
    (48000, 2, 48000, 5.999),      # 16.0027: the shortest bins the fused form takes
    (48000, 2, 96000, 0.1921),     # 999.5: the longest
    (48000, 1, 48000, 1.7),        # 28.2, mono
    (44100, 3, 44100, 1.0),        # 132.3, three channels (a tile's first sample at any alignment)
    (48000, 6, 48001, 2.0),        # 144.003, 5.1
    (96000, 8, 96000, 3.0),        # 256 exactly, but eight channels take the general form
    (22050, 2, 33075, 1.5),        # 44.1
    (48000, 2, 480000, 9.9999),    # the bench length with a window a hair off: 96.00096 samples per bin over 5000 tiles
])
def test_decimation_any_samples_per_bin(oracle, rate, channels, frames, window):
    rng = np.random.default_rng(rate + channels + frames)
    n = frames * channels
    w = int(window * 1000.0)
    spp = n / w
    assert 16.0 <= spp <= 1000.0
    xs = [_spiked(rng, n, spp) for _ in range(2)]
    b = ssa.Batch(rate, channels, 2, frames, 4096, 1024, flags=L.SS_BATCH_WAVEFORM | L.SS_BATCH_LUFS, waveform_window=window)
    assert b.geometry.waveform_fused == 1
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    for i, x in enumerate(xs):
        ref = oracle.get_waveform(x, window)[:, 1].astype(np.float32)
        got = b.waveform(i).reshape(-1)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(got), nan), i
        bad = np.flatnonzero(got.view(np.uint32)[~nan] != ref.view(np.uint32)[~nan])
        assert bad.size == 0, (i, bad[:8], got[~nan][bad[:8]], ref[~nan][bad[:8]])
        assert nan.any() and np.isinf(ref).any()


def test_kernel_timing_ring_counts_every_pass_without_synchronising():
    """ss_batch_timing_enable keeps a ring of 32 passes' event sets: passes queue back to back (bench.py times its kernels inside its
    timed region), every pass is counted once — also beyond the ring's depth, where ss_batch_run collects first — and the results
    of a timed pass equal an untimed one's."""
    rate, frames = 48000, 48000
    b = ssa.Batch(rate, 2, 8, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(3, 0)
    b.run(); b.sync()
    want = [(r.integrated_lufs, tuple(r.true_peak[:2])) for r in b.results()]
    fft0 = b.fft(0).copy()
    b.timing_enable(True)
    base = [b.timing_read(k) for k in range(L.SS_KERNEL_COUNT)]
    for n_pass in (5, 32, 71):                                   # inside the ring, exactly the ring, beyond it
        for _ in range(n_pass):
            b.run()                                              # (no sync between passes)
        now = [b.timing_read(k) for k in range(L.SS_KERNEL_COUNT)]
        for k in (L.SS_KERNEL_FFT, L.SS_KERNEL_TIME_DOMAIN, L.SS_KERNEL_FINALIZE):
            ms, n = now[k][0] - base[k][0], now[k][1] - base[k][1]
            assert n == n_pass, (k, n, n_pass)
            assert 0.0 < ms / n < 50.0, (k, ms, n)
        base = now
    b.timing_enable(False)                                       # (switching the mode clears the counters)
    b.run(); b.sync()
    assert b.timing_read(L.SS_KERNEL_FFT) == (0.0, 0)            # an untimed pass is not counted
    got = [(r.integrated_lufs, tuple(r.true_peak[:2])) for r in b.results()]
    assert got == want and np.array_equal(b.fft(0), fft0)
