"""Parity at the shapes bench.py times (BASELINE configs 3 and 5) — the exact launch geometry the benchmark runs,
not a reduced one — plus the low-level and special-value behaviour of the f16-split true peak.

Everything goes through the C ABI; the oracle (oracle/ss_oracle.c) is the checker.  Bars: decimation bit-exact,
spectrum / LUFS / LRA within 0.01 dB, true peak within 1e-4 relative (BASELINE.json north_star).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import db_close, make_stereo

pytestmark = pytest.mark.gpu

TOL_DB = 0.01


def lufs_close(a, b, tol=TOL_DB):
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= tol


def rel_close(a, b, rel=1e-4):
    return abs(a - b) <= rel * max(abs(b), 1e-30)


def _threads():
    return max(1, min(64, len(os.sched_getaffinity(0))))


F32, F16X3 = L.SS_TP_ARITH_F32, L.SS_TP_ARITH_F16X3


AUTO, RUN_IN, WHOLE = L.SS_TD_AUTO, L.SS_TD_RUN_IN, L.SS_TD_WHOLE_STREAMS


@pytest.mark.parametrize("overlap,arith,td_mode", [(0, None, AUTO), (1, None, AUTO), (2, None, AUTO), (0, F32, AUTO), (0, F16X3, AUTO),
                                                    (0, None, RUN_IN), (0, None, WHOLE), (0, F16X3, WHOLE)])
def test_config3_bench_geometry_matches_oracle(oracle, overlap, arith, td_mode):
    """BASELINE config 3 exactly as bench.py runs it: 1024 synthetic streams x 10 s x 48 kHz stereo, N = 4096,
    hop 1024.  The geometry the shape selects (116 windows per spectrum workgroup, 4 time segments per stream) is asserted,
    then eight streams — first, last and the ones either side of every quarter — are compared in
    full with the oracle (every window of both rows, LUFS, LRA, both peaks, every decimation bin), and the corpus
    histograms with the sum of all 1024 per-stream oracle histograms.
    arith: the 4x true peak at the reference's f32 width (the default, None, which bench.py's headline times:
    v_pk_fma_f32 on the stereo tile path), set explicitly, and as the opt-in f16x3 split.
    td_mode: how the time-domain kernel walks a stream (ss_batch_set_time_domain_mode) — the default (time segments with the
    exact state hand-over: a fix-up launch over the first two sub-blocks of segments 1..3), the run-in form of earlier rounds,
    and whole-stream workgroups."""
    rate, frames, ns = 48000, 480000, 1024
    b = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    b.set_overlap(overlap)
    b.set_time_domain_mode(td_mode)
    assert b.true_peak_arith == F32                              # the default is the reference's width
    if arith is not None:
        b.set_true_peak_arith(arith)
    assert b.true_peak_arith == (F32 if arith is None else arith)
    b.run(); b.sync()
    g, lay = b.geometry, b.layout
    assert (lay.n_windows, lay.n_bins) == (464, 1705)
    assert g.fft_windows_per_block == 116 and g.fft_blocks == 4096
    if td_mode == WHOLE:
        assert (g.td_split, g.td_segments, g.td_warm_subblocks, g.td_fixup_subblocks) == (1, 1, 0, 0)
    else:
        assert g.td_split == 0 and g.td_segments == 4 and g.td_segment_subblocks == 25
        assert (g.td_warm_subblocks, g.td_fixup_subblocks) == ((1, 0) if td_mode == RUN_IN else (0, 2))
    assert g.waveform_fused == 1 and g.td_true_peak_factor == 4 and g.overlap == int(overlap)
    res = b.results()
    picks = [0, 1, 255, 256, 511, 512, 1022, 1023]
    xs = {i: b.download_input(i) for i in picks}
    with ThreadPoolExecutor(_threads()) as ex:
        refs = dict(zip(picks, ex.map(lambda i: oracle.analyze_stream(rate, xs[i], 4096, 1024), picks)))
    for i in picks:
        ref = refs[i]
        assert ref["n_windows"] == 464
        fft = b.fft(i)
        for w in range(464):
            for c in range(2):
                assert db_close(fft[w, c], ref["fft"][w, c], TOL_DB, survey=True), (i, w, c)
        assert lufs_close(res[i].integrated_lufs, ref["integrated"]), (i, res[i].integrated_lufs, ref["integrated"])
        assert abs(res[i].loudness_range - ref["lra"]) <= TOL_DB
        tp, sp = b.peaks(i)
        for c in range(2):
            assert rel_close(res[i].true_peak[c], ref["true_peak"][c]), (i, c)
            assert res[i].sample_peak[c] == ref["sample_peak"][c]
            assert tp[c] == res[i].true_peak[c] and sp[c] == res[i].sample_peak[c]
        assert np.array_equal(b.waveform(i).reshape(-1), ref["wave"][:, 1].astype(np.float32)), i

    # corpus gate input: the device's summed histograms == the sum of every stream's oracle histograms, exactly
    ids = list(range(ns))
    inputs = [b.download_input(i) for i in ids]                      # downloads stay on this thread (one handle, one thread)

    def hist_of_x(x):
        m = oracle.Meter(2, rate)
        m.add_frames(x)
        return m.block_hist(), m.st_hist(), m.integrated()
    with ThreadPoolExecutor(_threads()) as ex:
        hs = list(ex.map(hist_of_x, inputs))
    hb, hst = b.histograms()
    assert np.array_equal(hb, sum(h[0] for h in hs))
    assert np.array_equal(hst, sum(h[1] for h in hs))
    assert ssa.corpus_integrated_lufs(hb) == oracle.gated_loudness_hist(hb)
    # and every stream's integrated loudness
    for i in ids:
        assert lufs_close(res[i].integrated_lufs, hs[i][2]), i


@pytest.mark.parametrize("tp_factor,overlap,arith,td_mode", [(4, 0, None, AUTO), (0, 0, None, AUTO), (4, 1, None, AUTO), (4, 2, None, AUTO), (4, 0, F32, AUTO),
                                                             (0, 0, F32, AUTO), (4, 0, F16X3, AUTO), (4, 0, None, RUN_IN), (4, 0, None, WHOLE), (0, 0, None, WHOLE)])
def test_config5_bench_shape_all_channels(oracle, tp_factor, overlap, arith, td_mode):
    """BASELINE config 5 as bench.py times it: 64 streams x 10 s x 96 kHz x 8 channels, N = 16384 per channel at hop
    1024, true peak forced to 4x (the benchmark) and at the crate's rule (2x at 96 kHz).  Four streams are checked in
    full on the meter side (LUFS, LRA, all EIGHT channels' true and sample peaks through ss_batch_peaks) and on a
    strided sample of windows on all eight channels."""
    rate, ch, frames, ns = 96000, 8, 960000, 64
    b = ssa.Batch(rate, ch, ns, frames, 16384, 1024, flags=L.SS_BATCH_ALL, true_peak_factor=tp_factor)
    b.synthesize(0x5EED0000, 0)
    b.set_overlap(overlap)                                      # the 16384-point run kernel beside the 8-channel time-domain kernel
    if arith is not None:                                       # None: the default = SS_TP_ARITH_F32 (the 8-channel f32 tile path)
        b.set_true_peak_arith(arith)
    assert b.true_peak_arith == (F32 if arith is None else arith)
    b.set_time_domain_mode(td_mode)
    b.run(); b.sync()
    g, lay = b.geometry, b.layout
    assert (lay.n_windows, lay.fft_channels, lay.n_bins) == (921, 8, 3410)
    assert g.td_true_peak_factor == (4 if tp_factor == 4 else 2)
    if td_mode == WHOLE:
        assert g.td_split == 1 and g.td_segments == 1
    else:
        assert g.td_segments > 1                                # the segmented path is what the bench times
        assert (g.td_warm_subblocks, g.td_fixup_subblocks) == ((1, 0) if td_mode == RUN_IN else (0, 2))
    res = b.results()
    picks = [0, 21, 42, 63]
    xs = {i: b.download_input(i) for i in picks}

    def meter_of(i):
        m = oracle.Meter(ch, rate, force_tp_factor=tp_factor)
        m.add_frames(xs[i])
        return (m.integrated(), m.loudness_range(), [m.true_peak(c) for c in range(ch)], [m.sample_peak(c) for c in range(ch)])
    with ThreadPoolExecutor(4) as ex:
        ms = dict(zip(picks, ex.map(meter_of, picks)))
    for i in picks:
        integ, lra, tps, sps = ms[i]
        assert lufs_close(res[i].integrated_lufs, integ), (i, res[i].integrated_lufs, integ)
        assert abs(res[i].loudness_range - lra) <= TOL_DB
        tp, sp = b.peaks(i)
        for c in range(ch):
            assert rel_close(tp[c], max(tps[c], sps[c])), (i, c, tp[c], tps[c])
            assert sp[c] == sps[c], (i, c)
        fft = b.fft(i)
        xm = xs[i].reshape(frames, ch)
        for w in list(range(0, lay.n_windows, 97)) + [lay.n_windows - 1]:
            start = (w + 1) * 1024
            for c in range(ch):
                ref = oracle.get_fft(rate, xm[start:start + 16384, c])
                assert db_close(fft[w, c], ref[:, 1], TOL_DB), (i, w, c)


@pytest.mark.parametrize("tp_factor", [4, 0])
def test_config5_every_window_of_two_streams(oracle, tp_factor):
    """Config 5's spectrum to the standard of config 3's: EVERY window (921) of all EIGHT channels of two streams — the first and the
    last of the batch — against the oracle, with the benchmark's forced 4x true peak beside it and with the crate's rule (2x at
    96 kHz).  Every row meets SURVEY section 7's wording of the bar (0.01 dB at >= -90 dBFS, 1e-4 of the row's largest amplitude
    below).  The stricter row-peak metric (0.01 dB down to 70 dB under the row's own loudest bin, wherever that is) is met by all
    but a handful of the 14736 rows: at N = 16384 a bin 70 dB under the peak sits where two f32 transforms of different radix are
    each ~0.007 dB from the f64 result (stream 63, window 647, channel 3: a -42 dBFS row, device 0.0134 dB from the oracle at a
    -112 dBFS bin; tools/probe_cfg5_rows.py).  Such a row must be within 0.015 dB of the oracle AND within 0.01 dB of the f64
    transform of the same windowed samples (conftest.f64_spectrum_row) — i.e. the excess is the oracle's rounding, not the device's."""
    from conftest import db_close_survey, f64_spectrum_row
    rate, ch, frames, ns = 96000, 8, 960000, 64
    b = ssa.Batch(rate, ch, ns, frames, 16384, 1024, flags=L.SS_BATCH_ALL, true_peak_factor=tp_factor)
    b.synthesize(0x5EED0000, 0)
    b.run(); b.sync()
    lay, g = b.layout, b.geometry
    assert (lay.n_windows, lay.fft_channels, lay.n_bins) == (921, 8, 3410)
    assert g.fft_windows_per_block == 116 and g.td_true_peak_factor == (4 if tp_factor == 4 else 2)
    n_edge = 0
    for i in (0, ns - 1):
        x = b.download_input(i).reshape(frames, ch)
        cols = [np.ascontiguousarray(x[:, c]) for c in range(ch)]
        fft = b.fft(i)
        jobs = [(w, c) for w in range(lay.n_windows) for c in range(ch)]

        def row_state(job):
            w, c = job
            start = (w + 1) * 1024                                  # window [p - N, p) at p = (w + 17) hop (tui.rs:1489)
            s = cols[c][start:start + 16384]
            ref = oracle.get_fft(rate, s)[:, 1]
            if not db_close_survey(fft[w, c], ref, TOL_DB):
                return 2
            if db_close(fft[w, c], ref, TOL_DB):
                return 0
            return 1 if db_close(fft[w, c], ref, 0.015) and db_close(fft[w, c], f64_spectrum_row(oracle, rate, s, 16384), TOL_DB) else 2
        with ThreadPoolExecutor(_threads()) as ex:
            st = list(ex.map(row_state, jobs, chunksize=64))
        bad = [jobs[k] for k, v in enumerate(st) if v == 2]
        assert not bad, (i, len(bad), bad[:8])
        n_edge += sum(1 for v in st if v == 1)
    assert n_edge <= 4, n_edge                                      # (one, on the synthetic corpus of the benchmark)


def test_batch_peaks_api(oracle):
    from conftest import make_multich
    rate, ch, frames = 48000, 6, 48000 * 2
    x = make_multich(77, frames, ch, rate)
    b = ssa.Batch(rate, ch, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK)
    b.upload(0, np.concatenate([x, (x * np.float32(0.5)).astype(np.float32)]))
    b.run(); b.sync()
    for s, scale in ((0, 1.0), (1, 0.5)):
        m = oracle.Meter(ch, rate); m.add_frames((x * np.float32(scale)).astype(np.float32))
        tp, sp = b.peaks(s)
        assert tp.shape == (ch,)
        for c in range(ch):
            assert rel_close(tp[c], max(m.true_peak(c), m.sample_peak(c))) and sp[c] == m.sample_peak(c)
    # capacity and argument errors
    buf = (np.empty(3, np.float64)).ctypes.data_as(L.C.POINTER(L.C.c_double))
    assert L.lib().ss_batch_peaks(b._h, 0, buf, None, 3) == L.SS_ERR_CAPACITY
    assert L.lib().ss_batch_peaks(b._h, 2, buf, None, 8) == L.SS_ERR_INVALID_ARG
    b2 = ssa.Batch(rate, 2, 1, frames, 4096, 1024, flags=L.SS_BATCH_FFT)
    assert L.lib().ss_batch_peaks(b2._h, 0, buf, None, 8) == L.SS_ERR_INVALID_MODE


@pytest.mark.parametrize("arith", [F32, F16X3])
@pytest.mark.parametrize("peak", [1e-5, 1e-6, 1e-7, 2.0 ** -23, 3e-9, 0.0])
def test_true_peak_quiet_streams(oracle, peak, arith):
    """Both true-peak arithmetics (the default f32 product and the opt-in f16 split) at very low levels.  The f16-split true-peak product scales every tile by a power of two taken from its own sample peak, so its
    relative accuracy does not depend on level: streams peaking at -100, -120, -140 dBFS, at one LSB of 24-bit PCM,
    far below that, and digital silence all stay within 1e-4 of the oracle (a fixed x256 scale lost this below
    about -120 dBFS, where the low f16 halves went sub-normal)."""
    rate, frames = 48000, 48000 * 3
    base = make_stereo(123, frames, rate, level=1.0)
    x = (base * np.float32(peak / max(np.abs(base).max(), 1e-30))).astype(np.float32) if peak else np.zeros_like(base)
    b = ssa.Batch(rate, 2, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK)
    # second stream: the same quiet programme with one loud burst, so tiles of very different scale sit side by side
    y = x.copy()
    y[2 * 20000:2 * 20400] = base[2 * 20000:2 * 20400] * np.float32(0.9)
    b.set_true_peak_arith(arith)
    b.upload(0, np.concatenate([x, y])); b.run(); b.sync()
    res = b.results()
    for i, s in enumerate((x, y)):
        m = oracle.Meter(2, rate); m.add_frames(s)
        for c in range(2):
            ref = max(m.true_peak(c), m.sample_peak(c))
            assert rel_close(res[i].true_peak[c], ref) or (ref == 0.0 and res[i].true_peak[c] == 0.0), (i, c, res[i].true_peak[c], ref)
            assert res[i].sample_peak[c] == m.sample_peak(c)
    # streaming handle, 16384-sample slices (the tick driver's feed)
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
    an.set_true_peak_arith(arith)
    m = oracle.Meter(2, rate)
    for off in range(0, y.size - 16384, 16384):
        an.add_samples(y[off:off + 16384]); m.add_frames(y[off:off + 16384])
    l, r = an.get_true_peak()
    assert rel_close(l, max(m.true_peak(0), m.sample_peak(0))) and rel_close(r, max(m.true_peak(1), m.sample_peak(1)))


def test_nan_samples_next_to_the_interpolator_match_the_crate(oracle):
    """Until round 5 this pinned DESIGN section 6's one deliberate deviation (the matrix form lost the whole 16-sample window of a
    NaN, the crate the 12 outputs per phase whose taps touch it).  A tile that holds a non-finite sample — or has one within the
    interpolator's reach in front of it — now runs the crate's own loop instead of the matrix product, so this is a plain parity
    test: peaks AND the loudness readings (the NaN poisons every later gating block: tests/test_gpu_nonfinite.py)."""
    rate, frames = 48000, 48000 * 4
    x = make_stereo(321, frames, rate, level=0.4)
    xn = x.copy()
    for f in (1000, 50001, 123456, 150000):
        xn[2 * f] = np.nan
    b = ssa.Batch(rate, 2, 2, frames, 4096, 1024, flags=L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate([x, xn])); b.run(); b.sync()
    res = b.results()
    for i, sig in enumerate((x, xn)):
        m = oracle.Meter(2, rate); m.add_frames(sig)
        for c in range(2):
            assert res[i].sample_peak[c] == m.sample_peak(c)
            assert rel_close(res[i].true_peak[c], max(m.true_peak(c), m.sample_peak(c))), (i, c)
        assert lufs_close(res[i].integrated_lufs, m.integrated()), (i, res[i].integrated_lufs, m.integrated())
        assert abs(res[i].loudness_range - m.loudness_range()) <= TOL_DB
    assert np.isfinite(res[1].true_peak[0]) and res[1].true_peak[0] <= res[0].true_peak[0] * (1 + 1e-6)     # a NaN can only hide outputs


@pytest.mark.parametrize("rate,slice_len", [(48000, 16384), (44100, 2 * 4410), (96000, 16384)])
def test_subnormal_filter_state_is_flushed_like_the_crate(oracle, rate, slice_len):
    """ebur128 flushes a sub-normal filter state to zero at the end of every internal filter call (oracle ss_oracle.c:570-571,
    SURVEY A5).  An impulse followed by silence: the carried DF-II state decays at e^-240 per second, becomes sub-normal after
    about three seconds and must then read EXACTLY zero — in the same streaming call as the oracle's — while before that it
    follows the oracle's state (the chunk-parallel recurrence is not the sequential one bit for bit, so the bar there is
    relative).  Loudness readings stay equal throughout.
    Scope: this is parity with the ORACLE's restatement — the crate's portable path, which flushes the carried state at the end
    of a filter call.  On x86-64 the crate is believed (recalled; its source is not in the image) to set the SSE flush-to-zero
    bit around the loop instead, which flushes every intermediate result: the two differ only below 2.2e-308, and the claim
    here is not pinned to that platform."""
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
    mm = oracle.Meter(2, rate)
    imp = np.zeros(2 * rate * 5, np.float32); imp[0] = 1.0; imp[1] = -0.5
    zero_at = {"gpu": None, "oracle": None}
    for k, off in enumerate(range(0, imp.size, slice_len)):
        an.add_samples(imp[off:off + slice_len]); mm.add_frames(imp[off:off + slice_len])
        for c in range(2):
            g, o = an.filter_state(c), mm.filter_state(c)
            assert np.array_equal(g == 0.0, o == 0.0), (k, c, g, o)                     # the same components are (flushed to) zero
            nz = o != 0.0
            big = nz & (np.abs(o) > 1e-290)                                            # above the sub-normal range: full precision
            # Relative to the state's largest component, 1e-5: the chunk-parallel recurrence and the oracle's sequential one are
            # two roundings of a cancellation-prone decay (near-double pole: each step subtracts numbers 1e4 .. 1e5 times its
            # result).  tools/probe_state_drift.py, call by call: 1e-9 .. 2e-7 as a rule, 2e-6 in single calls that cross a
            # gating-block boundary deep in the decay (96 kHz).  At 192 kHz (not a case here) the two recurrences are 4e-5 apart
            # after 0.7 s and 1e-4 after 1.5 s of decay, at 1e-75 .. 1e-147 of full scale, whichever way the tiles are walked.
            # No reading depends on those digits: the loudness values below are compared exactly / to 0.01 LU.
            if big.any():
                assert np.abs(g[big] - o[big]).max() <= 1e-5 * np.abs(o).max(), (k, c, g, o)
            assert not np.any((np.abs(g) < 2.2250738585072014e-308) & (g != 0.0)), (k, c, g)   # no sub-normal survives a call
        if zero_at["gpu"] is None and not an.filter_state(0).any():
            zero_at["gpu"] = k
        if zero_at["oracle"] is None and not mm.filter_state(0).any():
            zero_at["oracle"] = k
    assert zero_at["oracle"] is not None and zero_at["gpu"] == zero_at["oracle"], zero_at
    assert an.get_momentary_lufs() == mm.momentary() == -np.inf
    assert lufs_close(an.get_integrated_lufs(), mm.integrated())
    assert lufs_close(an.get_shortterm_lufs(), mm.shortterm())


def test_segment_handover_is_exact(oracle):
    """The K-weighting recurrence has a long memory; a stream cut into time segments must hand the filter state on.
    Default (SS_TD_AUTO): every segment starts from a zero state AT its boundary and a second launch re-runs the first two
    sub-blocks of every segment > 0 from the state the segment in front of it left.  Against the ONE-segment path (a
    4096-stream batch walks every stream with one wave, one segment) over all 1024 streams of the bench corpus:
      * segment 0 and the re-run head of segment 1 are BIT-EQUAL (same state, same tiles, same arithmetic; the later segments'
        heads start from a state that is itself one rounding history apart from the one-segment path's);
      * behind them nothing of the recurrence is missing — what remains is the rounding noise of the recurrence itself, the level
        at which the exact-by-construction whole-stream path differs from the one-segment path too (measured 3.0e-10 and 4.1e-10
        at the same near-silent sub-block; tools/probe_handover.py) — while the run-in form of earlier rounds is 9.4e-8 off at the
        first sub-block of a segment (its truncation)."""
    rate, frames, ns = 48000, 480000, 1024
    FL = L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM
    one = ssa.Batch(rate, 2, 4096, frames, 4096, 1024, flags=FL)
    one.synthesize(0x5EED0000, 0); one.run(); one.sync()
    assert (one.geometry.td_segments, one.geometry.td_split) == (1, 0)
    ref = np.stack([one.subblocks(i) for i in range(ns)]).reshape(ns, 100, 2)
    one.close()
    b = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=FL)
    b.synthesize(0x5EED0000, 0)
    worst = {}
    for mode in (AUTO, WHOLE, RUN_IN):
        b.set_time_domain_mode(mode); b.run(); b.sync()
        d = np.stack([b.subblocks(i) for i in range(ns)]).reshape(ns, 100, 2)
        rel = np.abs(d - ref) / np.maximum(np.abs(ref), 1e-300)
        worst[mode] = float(rel.max())
        if mode == AUTO:
            g = b.geometry
            assert (g.td_segments, g.td_segment_subblocks, g.td_fixup_subblocks) == (4, 25, 2)
            assert np.array_equal(d[:, :27], ref[:, :27]), "segment 0 and the re-run head of segment 1 equal the one-segment path bit for bit"
    assert worst[AUTO] <= 2e-9 and worst[WHOLE] <= 2e-9, worst      # rounding noise of the recurrence (3e-10 / 4e-10 measured)
    assert worst[AUTO] <= 0.1 * worst[RUN_IN], worst                 # ... and no longer the run-in's truncation (9.4e-8)
    b.close()


def test_single_file_segments_run_in_over_the_segment_in_front(oracle):
    """One file (BASELINE config 2): the stream is cut into 0.2 s segments whose tiles are dealt to the eight waves of a workgroup,
    and — the pass being a latency chain — every segment runs the FILTER over the 0.2 s in front of it, from zero, inside the one
    launch instead of leaving that to a second one (84 -> 74 us per pass).  The state a segment then starts its own frames with is
    what the second launch would have started from: against the exact-by-construction whole-stream walk the sub-block energies differ
    by the recurrence's rounding noise, on the bench material and on DC-offset material (the hand-over's worst case); results equal
    the oracle's."""
    rate, frames = 48000, 480000
    FL = L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM
    rng = np.random.default_rng(5)
    dc = np.empty(2 * frames, np.float32)
    dc[0::2] = (0.5 + 0.01 * rng.standard_normal(frames)).astype(np.float32)
    dc[1::2] = (-0.3 + 0.2 * np.sin(2 * np.pi * 5 * np.arange(frames) / rate)).astype(np.float32)
    for material in ("bench", "dc"):
        b = ssa.Batch(rate, 2, 1, frames, 4096, 1024, flags=FL)
        if material == "bench": b.synthesize(0x5EED0000, 0)
        else: b.upload(0, dc)
        b.run(); b.sync()
        g = b.geometry
        assert (g.td_split, g.td_segments, g.td_segment_subblocks, g.td_fixup_subblocks, g.td_warm_subblocks) == (2, 50, 2, 0, 2)
        got = b.subblocks(0).copy()
        r = b.results()[0]
        x = b.download_input(0)
        b.set_time_domain_mode(WHOLE); b.run(); b.sync()
        assert b.geometry.td_segments == 1
        ref = b.subblocks(0)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
        assert rel.max() <= 2e-9, (material, float(rel.max()))
        m = oracle.Meter(2, rate); m.add_frames(x)
        assert abs(r.integrated_lufs - m.integrated()) <= 1e-9 and abs(r.loudness_range - m.loudness_range()) <= 1e-9
        b.close()


def test_segmented_run_in_on_dc_offset_material(oracle):
    """DC-offset material is the worst case of the segment hand-over (the high-pass section's near-double pole: the state is
    4e4 times the offset and decays like n r^n).  The segmented batch path (default: exact hand-over by the fix-up launch) and
    the single-segment streaming path must land every gating block in the same 0.1 LU histogram bin."""
    rate, frames = 48000, 48000 * 10
    rng = np.random.default_rng(5)
    x = np.empty(2 * frames, np.float32)
    x[0::2] = (0.5 + 0.01 * rng.standard_normal(frames)).astype(np.float32)      # large DC offset
    x[1::2] = (-0.3 + 0.2 * np.sin(2 * np.pi * 100 * np.arange(frames) / rate)).astype(np.float32)
    ns = 600                                                                       # enough streams that segments are used
    b = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    for i in range(0, ns, 100):
        b.upload(i, np.tile(x, 100))
    b.run(); b.sync()
    assert b.geometry.td_segments > 1
    m = oracle.Meter(2, rate); m.add_frames(x)
    hb, hst = b.histograms()
    assert np.array_equal(hb, m.block_hist() * np.uint64(ns))
    assert np.array_equal(hst, m.st_hist() * np.uint64(ns))
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate); an.add_samples(x)       # one segment, true carried state
    assert lufs_close(an.get_integrated_lufs(), m.integrated())
    assert lufs_close(b.results()[ns // 2].integrated_lufs, m.integrated())


def test_rccl_communicator_single_rank_on_device(oracle):
    """The library's own RCCL binding on real hardware: librccl is opened, a one-rank communicator is created
    (ncclGetUniqueId + ncclCommInitRank), ncclCommCount answers, and the corpus all-reduce runs as a real
    ncclAllReduce(…, 2000, ncclUint64, ncclSum) on the batch's stream — with one rank the sum is the batch's own
    histograms.  (Two ranks need two GPUs: RCCL refuses duplicate devices; the N > 1 rank logic is covered on CPU by
    tests/test_distributed_comm.py and on 8 GPUs by the driver's scaling run.)"""
    from soundscope_amd.distributed import Comm, corpus_gate
    comm = Comm(0, 1, None, transport="rccl")
    assert comm.transport == "rccl" and comm.size == 1 and comm.rank == 0
    comm.barrier()
    assert list(comm.allreduce_sum_u64(np.array([3, 2 ** 40 + 1], np.uint64))) == [3, 2 ** 40 + 1]
    assert list(comm.allreduce_max_f64(np.array([1.5, -2.0]))) == [1.5, -2.0]
    rate, frames = 48000, 48000 * 4
    xs = [make_stereo(40 + i, frames, rate, level=0.1 + 0.2 * i, gap=(i == 1)) for i in range(3)]
    b = ssa.Batch(rate, 2, 3, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run()
    hb, hs = b.allreduce_histograms(comm)                   # queued behind the run on the batch's stream
    hb2, hs2 = b.histograms()
    assert np.array_equal(hb, hb2) and np.array_equal(hs, hs2)
    ms = []
    for x in xs:
        m = oracle.Meter(2, rate); m.add_frames(x); ms.append(m)
    assert np.array_equal(hb, sum(m.block_hist() for m in ms))
    gi, gr = corpus_gate(np.concatenate([hb, hs]))
    assert gi == oracle.gated_loudness_hist(hb)
    # the same gate queued on the device (what bench.py's step does: run + all-reduce + gate, no host sync in between)
    for c in (None, comm):
        b.run(); b.corpus_gate_enqueue(c)
        di, dr = b.corpus_gate_read()
        assert abs(di - gi) < 1e-9 and abs(dr - gr) < 1e-9, (di, gi, dr, gr)
    comm.close()


def test_corpus_allreduce_is_idempotent_per_pass(oracle, tmp_path):
    """The corpus all-reduce is in place, so it may happen only once per pass: two ranks (threads sharing this GPU, the
    host-staged transport), each calls ss_batch_allreduce_histograms TWICE and then queues the corpus gate with the
    communicator — the histograms must be the two ranks' sum, not a multiple of it; the next ss_batch_run starts over."""
    import threading
    from soundscope_amd.distributed import Comm
    rate, frames = 48000, 48000 * 4
    xs = [make_stereo(70 + i, frames, rate, level=0.1 + 0.15 * i) for i in range(4)]
    want = np.zeros(2000, np.uint64)
    for x in xs:
        m = oracle.Meter(2, rate); m.add_frames(x)
        want[:1000] += m.block_hist(); want[1000:] += m.st_hist()
    f, errs = str(tmp_path / "rdzv"), []

    def rank(r):
        try:
            comm = Comm(r, 2, f, transport="host-tcp")
            b = ssa.Batch(rate, 2, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
            b.upload(0, np.concatenate(xs[2 * r:2 * r + 2]))
            for _ in range(2):                                  # the flag is per pass: the second pass reduces again
                b.run()
                h1 = np.concatenate(b.allreduce_histograms(comm))
                h2 = np.concatenate(b.allreduce_histograms(comm))
                b.corpus_gate_enqueue(comm)
                gi, _ = b.corpus_gate_read()
                h3 = np.concatenate(b.histograms())
                assert np.array_equal(h1, want) and np.array_equal(h2, want) and np.array_equal(h3, want)
                assert abs(gi - oracle.gated_loudness_hist(want[:1000])) < 1e-9
            comm.close(); b.close()
        except Exception as e:       # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=rank, args=(r,)) for r in (1, 0)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs


def test_traffic_floor_utility(oracle):
    """ss_batch_traffic_floor times the spectrum kernel's loads and stores alone; it clobbers the spectra (documented)
    and the next pass recomputes them; shapes that do not take the N = 4096 stereo kernel are refused."""
    rate, frames = 48000, 48000 * 2
    xs = [make_stereo(70 + i, frames, rate) for i in range(4)]
    b = ssa.Batch(rate, 2, 4, frames, 4096, 1024)
    b.upload(0, np.concatenate(xs))
    b.run(); b.sync()
    ref = b.fft(2).copy()
    ms = b.traffic_floor(3)
    assert 0.0 < ms < 50.0
    assert not np.array_equal(b.fft(2), ref)                  # overwritten by the utility
    b.run(); b.sync()
    assert np.array_equal(b.fft(2), ref)                      # and recomputed, bit for bit
    b16 = ssa.Batch(rate, 2, 1, frames, 16384, 1024)
    out = L.C.c_double()
    assert L.lib().ss_batch_traffic_floor(b16._h, 1, L.C.byref(out)) == L.SS_ERR_UNSUPPORTED
    assert L.lib().ss_batch_traffic_floor(b._h, 0, L.C.byref(out)) == L.SS_ERR_INVALID_ARG


@pytest.mark.parametrize("channels,rate,frames", [(4, 48000, 48000 * 3 + 77), (8, 44100, 44100 * 2), (16, 48000, 48000), (1, 44100, 44100 * 3)])
def test_batch_true_peak_other_channel_counts(oracle, channels, rate, frames):
    """The planar f16 true peak for every channel count that takes it in a batch (1, 2, 4, 8: blocks of 64 / C frames) and
    the f32 product for 16; rates whose tiles are not whole column groups (44.1 kHz) mix both inside a tile."""
    from conftest import make_multich
    xs = [make_multich(300 + i, frames, channels, rate, level=0.2 + 0.5 * i) for i in range(2)]
    b = ssa.Batch(rate, channels, 2, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    for i, x in enumerate(xs):
        m = oracle.Meter(channels, rate); m.add_frames(x)
        tp, sp = b.peaks(i)
        for c in range(channels):
            assert rel_close(tp[c], max(m.true_peak(c), m.sample_peak(c))), (i, c, tp[c], m.true_peak(c))
            assert sp[c] == m.sample_peak(c)
        assert lufs_close(res[i].integrated_lufs, m.integrated())


@pytest.mark.parametrize("rate,channels", [(48000, 2), (96000, 2), (44100, 2), (48000, 1)])
def test_fused_decimation_nan_semantics(oracle, rate, channels):
    """Min-max decimation fused into the time-domain kernel (the batch path; integer samples-per-bin fast paths at
    48 / 96 kHz stereo, the general path otherwise) keeps f32::min / f32::max semantics (analyzer.rs:107-137): NaN
    samples are ignored, a bin of nothing but NaN stays NaN, +-inf and signed zeros pass through — bit for bit."""
    frames = rate * 2
    rng = np.random.default_rng(77)
    xs = []
    for i in range(3):
        x = (rng.standard_normal(frames * channels) * 0.1).astype(np.float32)
        spp = frames * channels // 2000
        x[spp * 10: spp * 11] = np.nan                    # exactly bin 10 when spp is an integer
        x[spp * 50 + 3: spp * 53 + 1] = np.nan            # a run across three bins
        x[spp * 100 + 1] = np.nan
        x[spp * 200] = np.inf
        x[spp * 300 + 5] = -np.inf
        x[spp * 400: spp * 401] = -0.0
        x[-5:] = np.nan                                   # the tail of the last bin
        xs.append(x)
    b = ssa.Batch(rate, channels, 3, frames, 4096, 1024, flags=L.SS_BATCH_WAVEFORM | L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    for i, x in enumerate(xs):
        ref = oracle.get_waveform(x, frames / rate)[:, 1].astype(np.float32)
        got = b.waveform(i).reshape(-1)
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)]), i
        assert np.array_equal(np.isnan(got), np.isnan(ref)), i
        assert np.isnan(ref).any()


def test_config4_full_size_equals_its_eight_shards(oracle):
    """BASELINE config 4 at its full size on ONE GPU (8192 synthetic streams x 10 s x 48 kHz stereo: 31.5 GB of input,
    52 GB of spectra) against the way eight ranks would compute it: eight 1024-stream shards with the stream ids of
    `shard_streams(8192, r, 8)`.  Size-independent properties of the sharding: the corpus histograms of the big batch
    are exactly the sum of the shards' histograms (the all-reduce's result), so the corpus gate and LRA agree exactly;
    every stream's loudness, range and peaks are the same in either batch although the launch geometry differs
    (spectrum workgroups per stream, time segments per stream: asserted to differ); three streams of the big batch
    — first, one in the middle of a shard, last — are checked against the oracle in full."""
    from soundscope_amd.distributed import shard_streams, corpus_gate
    rate, frames, total, world = 48000, 480000, 8192, 8
    try:
        big = ssa.Batch(rate, 2, total, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    except Exception as e:                              # a GPU with less free memory than the 85 GB this shape needs
        pytest.skip(f"config 4 does not fit this device: {e}")
    big.synthesize(0x5EED0000, 0)
    big.run(); big.sync()
    gb = big.geometry
    rb = big.results()
    big_res = [(r.integrated_lufs, r.loudness_range, r.true_peak[0], r.true_peak[1], r.sample_peak[0], r.sample_peak[1]) for r in rb]
    hb_big, hs_big = big.histograms()
    picks = [0, 4097, total - 1]
    xs = {i: big.download_input(i) for i in picks}
    ffts = {i: big.fft(i) for i in picks}
    waves = {i: big.waveform(i).reshape(-1).copy() for i in picks}
    big.close()

    hb_sum, hs_sum = np.zeros(1000, np.uint64), np.zeros(1000, np.uint64)
    shard_geo = None
    for r in range(world):
        first, count = shard_streams(total, r, world)
        assert (first, count) == (1024 * r, 1024)
        b = ssa.Batch(rate, 2, count, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
        b.synthesize(0x5EED0000, first)
        b.run(); b.sync()
        shard_geo = b.geometry
        hb, hs = b.histograms()
        hb_sum += hb; hs_sum += hs
        res = b.results()
        for i in range(count):
            got = (res[i].integrated_lufs, res[i].loudness_range, res[i].true_peak[0], res[i].true_peak[1],
                   res[i].sample_peak[0], res[i].sample_peak[1])
            want = big_res[first + i]
            assert lufs_close(got[0], want[0], 1e-9) and abs(got[1] - want[1]) <= 1e-9, (r, i, got, want)
            assert got[4:] == want[4:] and rel_close(got[2], want[2], 1e-6) and rel_close(got[3], want[3], 1e-6), (r, i, got, want)
        if r == 0:
            assert np.array_equal(b.download_input(0), xs[0])           # same stream ids -> same synthetic samples
        b.close()
    assert (gb.td_segments, gb.fft_windows_per_block) != (shard_geo.td_segments, shard_geo.fft_windows_per_block)
    assert np.array_equal(hb_big, hb_sum) and np.array_equal(hs_big, hs_sum)
    assert corpus_gate(np.concatenate([hb_big, hs_big])) == corpus_gate(np.concatenate([hb_sum, hs_sum]))
    assert int(hb_big.sum()) > 0

    with ThreadPoolExecutor(_threads()) as ex:
        refs = dict(zip(picks, ex.map(lambda i: oracle.analyze_stream(rate, xs[i], 4096, 1024), picks)))
    for i in picks:
        ref = refs[i]
        for w in range(464):
            for c in range(2):
                assert db_close(ffts[i][w, c], ref["fft"][w, c], TOL_DB, survey=True), (i, w, c)
        assert lufs_close(big_res[i][0], ref["integrated"]) and abs(big_res[i][1] - ref["lra"]) <= TOL_DB
        for c in range(2):
            assert rel_close(big_res[i][2 + c], ref["true_peak"][c]) and big_res[i][4 + c] == ref["sample_peak"][c]
        assert np.array_equal(waves[i], ref["wave"][:, 1].astype(np.float32)), i


_SWEEP = [(r, c) for r in (8000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000) for c in (1, 2, 3, 4, 6, 8)]


@pytest.mark.parametrize("rate,channels", _SWEEP)
def test_time_domain_chunk_lengths_across_rates_and_channels(oracle, rate, channels):
    """The time-domain kernel's tile geometry (chunk length, tiles per sub-block, lanes per chunk row, planar or f32
    true peak, fused or standalone decimation) is chosen per (rate, channel count) by a cost model; this walks the
    model's whole decision table — nine rates x six channel counts, chunk lengths 20 ... 65, exact and inexact tilings —
    with a batch whose length is not a whole number of sub-blocks or segments, and holds every loudness figure, every
    channel's peaks and every decimation bin to the oracle."""
    from conftest import make_multich
    frames = int(rate * 2.37) + 5
    xs = [make_multich(900 + 7 * i + channels, frames, channels, rate, level=0.15 + 0.6 * i) for i in range(3)]
    b = ssa.Batch(rate, channels, 3, frames, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    for i, x in enumerate(xs):
        m = oracle.Meter(channels, rate); m.add_frames(x)
        tp, sp = b.peaks(i)
        for c in range(channels):
            assert rel_close(tp[c], max(m.true_peak(c), m.sample_peak(c))), (i, c, tp[c], m.true_peak(c))
            assert sp[c] == m.sample_peak(c), (i, c)
        assert lufs_close(res[i].integrated_lufs, m.integrated()), (i, res[i].integrated_lufs, m.integrated())
        assert abs(res[i].loudness_range - m.loudness_range()) <= TOL_DB
        ref = oracle.get_waveform(x, frames / rate)[:, 1].astype(np.float32)
        assert np.array_equal(b.waveform(i).reshape(-1)[:ref.size], ref), i


@pytest.mark.parametrize("rate,channels,bound", [(48000, 2, 1e-10), (96000, 2, 1e-9), (192000, 2, 2e-8), (192000, 1, 2e-8), (44100, 8, 1e-10), (384000, 2, 1e-6)])
def test_kweighting_is_f64_accurate_at_every_rate(oracle, rate, channels, bound):
    """The K-weighted 100 ms sub-block energies of the parallel (two-pass, in-wave scan) filter against a sequential f64
    filter (scipy lfilter, transposed direct form) with the meter's own coefficients.  The poles move towards z = 1 with the
    sample rate and the DF-II state grows to 1e5 ... 1e8 times the input; the chunk scan therefore runs in backward-
    difference coordinates of the state (ss_tables.cpp kweight_transition_pow) — with plain powers of the companion
    matrix the same comparison read 3e-8 at 48 kHz, 5e-6 at 96 kHz and 2e-3 at 192 kHz."""
    from scipy.signal import lfilter
    from conftest import make_multich
    frames = int(rate * 2.37) + 5
    xs = [make_multich(900 + 7 * i + channels, frames, channels, rate, level=0.15 + 0.6 * i) for i in range(3)]
    b = ssa.Batch(rate, channels, 3, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    assert b.geometry.td_segments > 1                       # segments with a run-in, like the benchmark
    bb, aa = oracle.Meter(channels, rate).coeffs()
    S = (rate + 5) // 10
    n = frames // S
    for i, x in enumerate(xs):
        y = lfilter(bb, aa, x.reshape(-1, channels).astype(np.float64), axis=0)
        ref = np.array([[np.sum(y[k * S:(k + 1) * S, c] ** 2) for c in range(channels)] for k in range(n)])
        got = b.subblocks(i)[:n]
        assert (np.abs(got - ref) / ref).max() <= bound, (i, (np.abs(got - ref) / ref).max())


@pytest.mark.parametrize("n_streams", [1, 3, 17, 100, 257, 700])
def test_launch_geometry_across_batch_sizes(oracle, n_streams):
    """The launch geometry (windows per spectrum workgroup, time segments per stream and their run-in) is a function of
    the batch size; between the handful-of-streams shapes of the parity tests and the 1024-stream benchmark shape this
    walks the sizes in between — first, middle and last stream of each batch in full against the oracle."""
    rate, frames = 48000, int(48000 * 3.3) + 11
    b = ssa.Batch(rate, 2, n_streams, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0xC0FFEE, 5)
    b.run(); b.sync()
    res = b.results()
    lay = b.layout
    picks = sorted({0, n_streams // 2, n_streams - 1})
    for i in picks:
        x = b.download_input(i)
        ref = oracle.analyze_stream(rate, x, 4096, 1024)
        assert ref["n_windows"] == lay.n_windows
        fft = b.fft(i)
        for w in range(lay.n_windows):
            for c in range(2):
                assert db_close(fft[w, c], ref["fft"][w, c], TOL_DB, survey=True), (i, w, c)
        assert lufs_close(res[i].integrated_lufs, ref["integrated"]) and abs(res[i].loudness_range - ref["lra"]) <= TOL_DB
        for c in range(2):
            assert rel_close(res[i].true_peak[c], ref["true_peak"][c]) and res[i].sample_peak[c] == ref["sample_peak"][c]
        assert np.array_equal(b.waveform(i).reshape(-1), ref["wave"][:, 1].astype(np.float32)), i


@pytest.mark.parametrize("frames", [1, 100, 4095, 4096, 4097, 5120, 5121, 19199, 19200, 19201, 19679, 19680, 24000, 143999, 144000, 144001, 148800, 148801])
def test_stream_length_edges(oracle, frames):
    """Lengths at every threshold of the path at 48 kHz: no window / the first window (the reference skips the window whose
    left edge is sample 0, so N + hop frames are needed), the first 400 ms gating block and the first 100 ms step behind
    it, the first 3 s short-term block and its first 1 s step — five streams per batch, each against the oracle."""
    rate = 48000
    xs = [make_stereo(600 + i, frames, rate, level=0.1 + 0.2 * i) for i in range(5)]
    b = ssa.Batch(rate, 2, 5, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.upload(0, np.concatenate(xs)); b.run(); b.sync()
    res = b.results()
    lay = b.layout
    for i, x in enumerate(xs):
        ref = oracle.analyze_stream(rate, x, 4096, 1024)
        assert ref["n_windows"] == lay.n_windows
        if lay.n_windows:
            fft = b.fft(i)
            for w in range(lay.n_windows):
                for c in range(2):
                    assert db_close(fft[w, c], ref["fft"][w, c], TOL_DB, survey=True), (i, w, c)
        assert lufs_close(res[i].integrated_lufs, ref["integrated"]), (i, res[i].integrated_lufs, ref["integrated"])
        assert abs(res[i].loudness_range - ref["lra"]) <= TOL_DB
        for c in range(2):
            assert rel_close(res[i].true_peak[c], ref["true_peak"][c]) and res[i].sample_peak[c] == ref["sample_peak"][c]
        got = b.waveform(i).reshape(-1)
        want = ref["wave"][:, 1].astype(np.float32)
        assert np.array_equal(got[:want.size], want), i
