"""Non-finite samples (NaN, +Inf, -Inf) through the METER, against the oracle (oracle/ss_oracle.c:542-625, :629-652) — the reference
call sites are /root/reference/src/analyzer.rs:139-141 (add_samples), :159-164 (get_true_peak), :170-182 (calculate_integrated_lufs).

What the crate does (ebur128 0.1.10, restated in the oracle): a NaN — or an infinity, which turns into one within two steps of the
recurrence (Inf - Inf) — entering Filter::process poisons the channel's DF-II state for good.  Every later filtered sample of
that channel is NaN, so is every later gating-block and short-term energy, `energy >= boundary` is false, and no block is ever
added to a histogram again: integrated loudness and range freeze at what the blocks IN FRONT of the sample said, momentary and
short-term read NaN.  Peaks: `if v > max` ignores a NaN, an infinity is the new maximum; the interpolator loses exactly the
outputs whose taps touch the sample.  A channel the crate maps to Channel::Unused (index 3 of a 6-channel meter) is not filtered
at all: whatever it carries has no effect on any loudness reading.

On the device a batch stream is cut into time segments that each start from a zero state, so nothing of this follows from the
recurrence alone across a segment boundary: the waves note the first sub-block with a non-finite sample per channel
(TdState::bad_key) and the gating kernels honour it.  Every test signal gets LOUDER behind the sample, so that blocks wrongly
counted there move the readings far beyond the 0.01 dB bar.
"""
import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_multich, make_stereo

pytestmark = pytest.mark.gpu

TOL_DB = 0.01
AUTO, RUN_IN, WHOLE = L.SS_TD_AUTO, L.SS_TD_RUN_IN, L.SS_TD_WHOLE_STREAMS
BAD = {"nan": np.float32(np.nan), "+inf": np.float32(np.inf), "-inf": np.float32(-np.inf)}


def same_db(a, b, tol=TOL_DB):
    """two loudness readings: both NaN, both the same infinity, or within tol"""
    if np.isnan(a) or np.isnan(b):
        return bool(np.isnan(a) and np.isnan(b))
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= tol


def same_peak(a, b, rel=1e-4):
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= rel * max(abs(b), 1e-30)


def stepped(seed, frames, rate, step_at, channels=2, lo=0.05, hi=0.6):
    """a programme that is 21.6 dB louder from frame `step_at` on (what a wrongly counted block behind it would show)"""
    x = (make_stereo(seed, frames, rate, level=1.0) if channels == 2 else make_multich(seed, frames, channels, rate, level=1.0)).reshape(frames, channels)
    x[:step_at] *= np.float32(lo)
    x[step_at:] *= np.float32(hi)
    return x.reshape(-1).copy()


def check_stream(oracle, rate, channels, x, res, peaks, tag):
    m = oracle.Meter(channels, rate)
    m.add_frames(x)
    assert same_db(res.integrated_lufs, m.integrated()), (tag, res.integrated_lufs, m.integrated())
    assert same_db(res.loudness_range, m.loudness_range()), (tag, res.loudness_range, m.loudness_range())
    tp, sp = peaks
    for c in range(channels):
        assert sp[c] == m.sample_peak(c), (tag, c, sp[c], m.sample_peak(c))
        assert same_peak(tp[c], max(m.true_peak(c), m.sample_peak(c))), (tag, c, tp[c], m.true_peak(c), m.sample_peak(c))
    return m


# ---------------------------------------------------------------- (a) BASELINE config 3's geometry: 4 segments x 25 sub-blocks
@pytest.mark.parametrize("td_mode", [AUTO, RUN_IN, WHOLE])
def test_config3_geometry_nonfinite_sample_poisons_every_later_block(oracle, td_mode):
    """1024 streams x 10 s x 48 kHz stereo as bench.py runs them (asserted: four time segments of 25 sub-blocks, the exact
    hand-over's fix-up over two, or one whole-stream workgroup).  A dozen streams carry ONE non-finite sample each: in
    segment 0, in the very last frame of a segment, in the first frame of the next, inside a fix-up sub-block, in the last
    segment, in the stream's last frame, in a sub-block's last frame (the +Inf block that IS counted) — on either channel.  Every one against the
    oracle's meter: integrated, range, both peaks; and the corpus histograms (bin for bin) against the sum of all 1024
    oracle histograms, which is what the corpus gate's all-reduce carries."""
    rate, frames, ns = 48000, 480000, 1024
    seg = 25 * 4800
    b = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    b.set_time_domain_mode(td_mode)
    cases = [   # (stream, kind, frame, channel)
        (3, "nan", 48000, 0), (4, "+inf", 48000, 1), (5, "-inf", 48001, 0),
        (100, "nan", seg - 1, 1), (101, "nan", seg, 0), (102, "+inf", seg - 1, 0), (103, "-inf", seg, 1),
        (200, "nan", seg + 4800 + 7, 0), (201, "nan", 2 * seg + 2 * 4800 - 1, 1),
        (500, "nan", 3 * seg + 12345, 0), (501, "+inf", 3 * seg + 4800 * 7 - 1, 1),      # the last frame of a sub-block: an Inf block
        (1023, "nan", frames - 1, 1), (1022, "-inf", frames - 1, 0), (1021, "nan", 0, 0),
    ]
    xs = {}
    for s, kind, f, c in cases:
        x = stepped(1000 + s, frames, rate, min(f + 2400, frames - 1))
        x[2 * f + c] = BAD[kind]
        xs[s] = x
        b.upload(s, x)
    b.run(); b.sync()
    g = b.geometry
    if td_mode == WHOLE:
        assert (g.td_split, g.td_segments) == (1, 1)
    else:
        assert g.td_split == 0 and g.td_segments == 4 and g.td_segment_subblocks == 25
        assert (g.td_warm_subblocks, g.td_fixup_subblocks) == ((1, 0) if td_mode == RUN_IN else (0, 2))
    res = b.results()
    hists = {}
    for s, kind, f, c in cases:
        m = check_stream(oracle, rate, 2, xs[s], res[s], b.peaks(s), (s, kind, f, c))
        hists[s] = (m.block_hist(), m.st_hist())
        assert res[s].n_gating_blocks == 97 and res[s].n_st_blocks == 8          # blocks EVALUATED (whatever the gate said)
    if td_mode != AUTO:
        return
    # corpus histograms: the poisoned streams must not have added a single block behind their sample
    from concurrent.futures import ThreadPoolExecutor
    import os
    rest = [i for i in range(ns) if i not in xs]
    inputs = [b.download_input(i) for i in rest]

    def hist_of(x):
        m = oracle.Meter(2, rate); m.add_frames(x)
        return m.block_hist(), m.st_hist()
    with ThreadPoolExecutor(max(1, min(64, len(os.sched_getaffinity(0))))) as ex:
        hs = list(ex.map(hist_of, inputs))
    hb, hst = b.histograms()
    assert np.array_equal(hb, sum(h[0] for h in hs) + sum(h[0] for h in hists.values()))
    assert np.array_equal(hst, sum(h[1] for h in hs) + sum(h[1] for h in hists.values()))


@pytest.mark.parametrize("kind", ["nan", "+inf", "-inf"])
@pytest.mark.parametrize("arith", [L.SS_TP_ARITH_F32, L.SS_TP_ARITH_F16X3])
def test_true_peak_next_to_a_nonfinite_sample_is_the_crates(oracle, kind, arith):
    """The interpolator next to the sample (this was DESIGN section 6's one documented deviation: the matrix form lost the whole
    16-sample window of a NaN where the crate loses the 12 outputs per phase that touch it).  The programme's loudest
    inter-sample peak is PUT right next to the hole — 13 frames behind it, the first output the crate still has — so that a
    lost window shows.  Eight streams, holes at tile starts, tile ends, in the halo of the next tile and mid-tile."""
    rate, frames = 48000, 48000 * 3
    b = ssa.Batch(rate, 2, 8, frames, 4096, 1024, flags=L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_LUFS)
    b.set_true_peak_arith(arith)
    xs = []
    for s in range(8):
        x = make_stereo(4000 + s, frames, rate, level=0.2)
        f = [960 * 7, 960 * 7 - 1, 960 * 9 - 5, 960 * 11 + 500, 4800 * 5, 4800 * 5 - 12, 77777, frames - 14][s]
        c = s & 1
        x[2 * f + c] = BAD[kind]
        # an inter-sample peak (fs/4, 45 degrees: 0.9 sample values, 1.27 true peak) right behind the hole's reach
        n = np.arange(12, 40)
        burst = (0.9 * np.sin(2 * np.pi * 0.25 * n + np.pi / 4)).astype(np.float32)
        k = n + f
        ok = k < frames
        x[2 * k[ok] + c] = burst[ok]
        xs.append(x)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    for s in range(8):
        check_stream(oracle, rate, 2, xs[s], res[s], b.peaks(s), (s, kind))


@pytest.mark.parametrize("rate,channels,tp", [(44100, 2, 0), (96000, 2, 0), (96000, 8, 4), (48000, 1, 0), (48000, 6, 0), (192000, 2, 0), (8000, 2, 0)])
def test_other_shapes_nonfinite(oracle, rate, channels, tp):
    """The same through the other instantiations: the 2x interpolator (96 kHz), none (192 kHz), mono, 5.1, eight channels with
    the forced 4x (BASELINE config 5's shape), general decimation at 44.1 kHz, a low rate."""
    frames = int(rate * 4.2)
    ns = 6
    b = ssa.Batch(rate, channels, ns, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM, true_peak_factor=tp)
    xs = []
    for s in range(ns):
        f = [rate // 2, rate + 17, 2 * rate - 1, 3 * rate, frames - 1, 5][s]
        x = stepped(5000 + s, frames, rate, min(f + rate // 20, frames - 1), channels)
        c = 0 if channels < 3 else (s % 3)              # (channels 0, 1, 2 are weighted in every layout)
        x[channels * f + c] = list(BAD.values())[s % 3]
        xs.append(x)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    for s in range(ns):
        m = oracle.Meter(channels, rate, force_tp_factor=tp)
        m.add_frames(xs[s])
        assert same_db(res[s].integrated_lufs, m.integrated()), (s, res[s].integrated_lufs, m.integrated())
        assert same_db(res[s].loudness_range, m.loudness_range()), s
        tpk, spk = b.peaks(s)
        for c in range(channels):
            assert spk[c] == m.sample_peak(c)
            assert same_peak(tpk[c], max(m.true_peak(c), m.sample_peak(c))), (s, c, tpk[c], m.true_peak(c))


# ---------------------------------------------------------------- (b) calculate_integrated_lufs / receive_audio_file
@pytest.mark.parametrize("kind", ["nan", "+inf", "-inf"])
def test_one_shot_and_file_open_with_a_nonfinite_sample_at_one_second(oracle, kind):
    """analyzer.rs:170-182 on a 10 s file with the sample at 1 s, quiet in front of it, loud behind: the crate's answer is the
    quiet second's loudness; and the fft_gain_compensation_db receive_audio_file derives from it (tui.rs:1229-1238)."""
    from oracle.app_driver import FileApp
    rate, frames = 48000, 480000
    x = stepped(77, frames, rate, rate + 2400)
    x[2 * rate] = BAD[kind]
    want = oracle.calculate_integrated_lufs(rate, 2, x)
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
    got = an.calculate_integrated_lufs(2, x)
    assert want is not None and same_db(got, want), (got, want)
    clean = x.copy(); clean[2 * rate] = 0.0
    assert abs(oracle.calculate_integrated_lufs(rate, 2, clean) - want) > 10.0          # (the test has teeth)
    sess = ssa.FileSession(x, 2, rate)
    app = FileApp(x, 2, rate)
    assert same_db(sess.fft_gain_compensation_db, app.fft_gain_compensation_db), (sess.fft_gain_compensation_db, app.fft_gain_compensation_db)
    assert np.array_equal(sess.audio_file_chart, app.audio_file_chart, equal_nan=True)


# ---------------------------------------------------------------- (c) the streaming handle and a tick
@pytest.mark.parametrize("kind", ["nan", "+inf", "-inf"])
@pytest.mark.parametrize("rate", [48000, 44100])
def test_streaming_handle_across_calls(oracle, kind, rate):
    """add_samples in the reference's slices (16384 samples per tick, tui.rs:1539) with the sample in the third call: every
    reading after every call — momentary, short-term, integrated, range, both peaks, the carried filter state."""
    frames = rate * 6
    x = stepped(31, frames, rate, rate)
    bad_at = 2 * 8192 + 4001
    x[2 * bad_at + 1] = BAD[kind]
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
    m = oracle.Meter(2, rate)
    for k, off in enumerate(range(0, x.size, 16384)):
        sl = x[off:off + 16384]
        an.add_samples(sl); m.add_frames(sl)
        if k % 3 == 0 or k < 6:
            assert same_db(an.get_momentary_lufs(), m.momentary()), (k, an.get_momentary_lufs(), m.momentary())
            assert same_db(an.get_shortterm_lufs(), m.shortterm()), (k, an.get_shortterm_lufs(), m.shortterm())
            assert same_db(an.get_integrated_lufs(), m.integrated()), (k, an.get_integrated_lufs(), m.integrated())
            assert same_db(an.get_loudness_range(), m.loudness_range()), k
            for c in range(2):
                assert an.get_sample_peak_channel(c) == m.sample_peak(c)
                assert same_peak(an.get_true_peak_channel(c), max(m.true_peak(c), m.sample_peak(c))), (k, c)
                got, want = an.filter_state(c), m.filter_state(c)
                assert np.array_equal(np.isnan(got), np.isnan(want)), (k, c, got, want)
    # the poisoned channel's state is NaN for good, the other one's is not
    assert np.isnan(an.filter_state(1)).all() and np.isfinite(an.filter_state(0)).all()
    # reset clears it (analyzer.rs:143-145)
    an.reset(); m.reset()
    an.add_samples(x[:96000]); m.add_frames(x[:96000])
    assert same_db(an.get_integrated_lufs(), m.integrated()) and np.isfinite(an.get_integrated_lufs())
    assert same_db(an.get_shortterm_lufs(), m.shortterm())


@pytest.mark.parametrize("kind", ["nan", "+inf"])
def test_file_session_ticks_over_a_nonfinite_sample(oracle, kind):
    """The tick driver (tui.rs:1482-1552: the 8x-overlapped refeed) over the sample: short-term history, statuses and the
    file analyzer's running readings against the restated App."""
    from oracle.app_driver import FileApp
    rate, frames = 48000, 48000 * 5
    x = stepped(55, frames, rate, 2 * rate)
    x[2 * (rate + 700)] = BAD[kind]
    sess = ssa.FileSession(x, 2, rate)
    app = FileApp(x, 2, rate)
    assert same_db(sess.fft_gain_compensation_db, app.fft_gain_compensation_db)
    for pos in range(2048 * 9, 2 * frames, 2048 * 2):
        res = sess.analyze_audio_file_samples(pos)
        ref = app.analyze_audio_file_samples(pos)
        for k in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status", "shortterm_status"):
            assert getattr(res, k) == ref[k], (pos, k, getattr(res, k), ref[k])
        assert same_db(res.shortterm, ref["shortterm"]), (pos, res.shortterm, ref["shortterm"])
    got, want = sess.lufs, app.lufs
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], atol=TOL_DB)
    assert same_db(sess.analyzer.get_integrated_lufs(), app.analyzer.meter.integrated())
    assert same_db(sess.analyzer.get_loudness_range(), app.analyzer.meter.loudness_range())


# ---------------------------------------------------------------- (d) a channel the crate does not filter
@pytest.mark.parametrize("kind", ["nan", "+inf"])
def test_unused_channel_carries_anything_without_effect(oracle, kind):
    """Six channels: index 3 is Channel::Unused in ebur128's default map — not filtered, not summed.  A non-finite sample there
    leaves every loudness reading as if the channel were silent (batch and handle); its PEAKS still see it (the peak scans
    run over every channel).  The same sample on channel 4 (a surround, weight 1.41) poisons the meter."""
    rate, frames, C = 48000, 48000 * 4, 6
    base = stepped(91, frames, rate, 2 * rate, channels=C)
    b = ssa.Batch(rate, C, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK)
    xs = []
    for ch in (3, 4):
        x = base.copy(); x[C * (rate + 99) + ch] = BAD[kind]; xs.append(x)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    clean = oracle.Meter(C, rate); clean.add_frames(base)
    for s, ch in enumerate((3, 4)):
        m = oracle.Meter(C, rate); m.add_frames(xs[s])
        assert same_db(res[s].integrated_lufs, m.integrated()), (ch, res[s].integrated_lufs, m.integrated())
        assert same_db(res[s].loudness_range, m.loudness_range())
        tp, sp = b.peaks(s)
        for c in range(C):
            assert sp[c] == m.sample_peak(c)
            assert same_peak(tp[c], max(m.true_peak(c), m.sample_peak(c))), (ch, c)
    assert same_db(res[0].integrated_lufs, clean.integrated()) and abs(res[1].integrated_lufs - clean.integrated()) > 5.0
    # the handle, streaming
    for s, ch in enumerate((3, 4)):
        an = ssa.Analyzer(); an.create_loudness_meter(C, rate)
        m = oracle.Meter(C, rate)
        for off in range(0, xs[s].size, 6 * 8000):
            sl = xs[s][off:off + 6 * 8000]
            an.add_samples(sl); m.add_frames(sl)
        assert same_db(an.get_integrated_lufs(), m.integrated())
        assert same_db(an.get_shortterm_lufs(), m.shortterm()), (ch, an.get_shortterm_lufs(), m.shortterm())
        assert same_db(an.get_momentary_lufs(), m.momentary())
        assert same_db(an.get_loudness_range(), m.loudness_range())
        for c in range(C):
            got, want = an.filter_state(c), m.filter_state(c)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (ch, c, got, want)
        assert np.array_equal(an.filter_state(3), np.zeros(4))                      # never filtered


# ---------------------------------------------------------------- the spectrum beside a refused window
@pytest.mark.parametrize("rate,channels,fft_n", [(48000, 1, 4096), (48000, 6, 4096), (48000, 2, 4096), (96000, 8, 16384), (48000, 2, 16384), (44100, 3, 2048)])
@pytest.mark.parametrize("kind", ["nan", "+inf"])
def test_spectrum_rows_next_to_a_refused_window(oracle, rate, channels, fft_n, kind):
    """get_fft refuses a window that holds a NaN or an infinite sample (spectrum-analyzer 1.7.0: NaNValuesNotSupported /
    InfinityValuesNotSupported, analyzer.rs:60-65) and the reference's driver then draws nothing for that tick (tui.rs:1500-1520); every
    OTHER window is analysed as ever.  In a batch: the rows of refused windows are non-finite in every bin, every other row equals
    the oracle's — also in the kernels that pack two windows into one transform (mono / multichannel N = 4096 at hop 1024: windows
    w and w + 1 ride together, and a sample in the last hop of w + 1 must not reach window w; found by tools/fuzz_batch.py once it
    injected non-finite samples, round 6)."""
    from conftest import db_close
    hop = 1024
    frames = fft_n + hop * 24 + 500
    x = make_multich(900 + channels, frames, channels, rate, level=0.5) if channels != 2 else make_stereo(900, frames, rate, level=0.5)
    holes = [(fft_n + hop * 5 + 17, 0), (fft_n + hop * 14 - 1, channels - 1), (fft_n + hop * 20 + hop // 2, 0)]
    for f, c in holes:
        x[channels * f + c] = BAD[kind]
    b = ssa.Batch(rate, channels, 1, frames, fft_n, hop, flags=L.SS_BATCH_FFT)
    b.upload(0, x); b.run(); b.sync()
    fft = b.fft(0)
    nw = frames // hop - fft_n // hop
    assert fft.shape[0] == nw
    sig = oracle.mid_side(x) if channels == 2 else [np.ascontiguousarray(x.reshape(frames, channels)[:, c]) for c in range(channels)]
    n_refused = 0
    for w in range(nw):
        start = (w + fft_n // hop + 1) * hop - fft_n
        for r in range(len(sig)):
            s = sig[r][start:start + fft_n]
            if np.isfinite(s).all():
                ref = oracle.get_fft(rate, s)[:, 1]
                assert np.isfinite(fft[w, r]).all(), (w, r)
                assert db_close(fft[w, r], ref, TOL_DB), (w, r)
            else:
                n_refused += 1
                assert not np.isfinite(fft[w, r]).any(), (w, r, fft[w, r][:8])
    assert n_refused >= 3 * min(fft_n // hop, 8) - 2             # (every hole refuses fft_n / hop windows of its row, as far as the stream reaches)
