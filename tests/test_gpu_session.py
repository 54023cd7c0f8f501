"""Tick drivers through the C ABI against the CPU restatement of the reference App
(oracle/app_driver.py follows tui.rs:1207-1241, :1427-1552).  Same bars as test_gpu_parity.py:
decimation bit-exact, spectrum / loudness within +-0.01 dB."""
import numpy as np
import pytest

import soundscope_amd as ssa
from conftest import db_close, make_stereo

pytestmark = pytest.mark.gpu

TOL_DB = 0.01


def lufs_close(a, b, tol=TOL_DB):
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= tol


def check_tick(res, ref, sess, app):
    for k in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status", "shortterm_status"):
        assert getattr(res, k) == ref[k], (k, getattr(res, k), ref[k])
    if ref["fft_ran"]:
        for got, want in ((sess.mid_fft, app.mid_fft), (sess.side_fft, app.side_fft)):
            assert got.shape == want.shape
            if want.shape[0] > 1:
                assert np.array_equal(got[:, 0], want[:, 0])
                assert db_close(got[:, 1], want[:, 1], TOL_DB)
            else:
                assert np.array_equal(got, want)
    assert lufs_close(res.shortterm, ref["shortterm"])


@pytest.mark.parametrize("rate,seconds", [(48000, 6.0), (44100, 4.3), (96000, 2.6), (32000, 3.1), (192000, 3.4), (88200, 1.1)])
def test_file_session_matches_reference_driver(oracle, rate, seconds):
    from oracle.app_driver import FileApp
    frames = int(rate * seconds)
    x = make_stereo(7 + rate, frames, rate=rate, level=0.6)
    sess = ssa.FileSession(x, 2, rate)
    app = FileApp(x, 2, rate)
    # receive_audio_file
    assert sess.duration_ms == app.duration_ms
    assert sess.audio_file_chart.shape == app.audio_file_chart.shape
    assert np.array_equal(sess.audio_file_chart, app.audio_file_chart)          # bit-exact decimation
    assert abs(sess.fft_gain_compensation_db - app.fft_gain_compensation_db) <= TOL_DB
    # playback: the player reports positions at hop 1024 frames (interleaved samples = 2048 per tick);
    # include the early ticks the reference skips (left bound 0) and positions past the end
    positions = list(range(2048, 2 * frames + 3 * 2048, 2048))
    for pos in positions[:40] + positions[40::7]:
        res = sess.analyze_audio_file_samples(pos)
        ref = app.analyze_audio_file_samples(pos)
        assert res.playhead == ref["playhead"]
        check_tick(res, ref, sess, app)
    assert np.allclose(sess.lufs, app.lufs, atol=TOL_DB)
    # the file_analyzer's running state after the 8x-overlapped feed
    assert lufs_close(sess.analyzer.get_integrated_lufs(), app.analyzer.meter.integrated())
    tl, tr = sess.analyzer.get_true_peak()
    for got, c in ((tl, 0), (tr, 1)):
        want = max(app.analyzer.meter.true_peak(c), app.analyzer.meter.sample_peak(c))
        assert abs(got - want) <= 1e-4 * want
    # seek: lufs history and meter are cleared
    sess.restart()
    app.restart()
    assert np.array_equal(sess.lufs, app.lufs)
    pos = 2048 * 30
    check_tick(sess.analyze_audio_file_samples(pos), app.analyze_audio_file_samples(pos), sess, app)
    assert np.allclose(sess.lufs, app.lufs, atol=TOL_DB)


def test_file_session_odd_cases(oracle):
    """Mono-flagged file (pos / 1), a NaN pair and an infinite pair inside some windows, odd sample count."""
    from oracle.app_driver import FileApp
    rate = 48000
    x = make_stereo(99, rate * 2, rate=rate, level=0.5)[:-1].copy()      # odd length: last sample unpaired
    x[2 * 30000] = np.nan                                                # mid and side NaN at pair 30000
    x[2 * 60000 + 1] = np.inf                                            # mid +inf, side -inf at pair 60000
    for channels in (2, 1):
        sess = ssa.FileSession(x, channels, rate)
        app = FileApp(x, channels, rate)
        assert sess.duration_ms == app.duration_ms
        assert np.array_equal(sess.audio_file_chart, app.audio_file_chart, equal_nan=True)
        assert (sess.fft_gain_compensation_db == app.fft_gain_compensation_db
                or abs(sess.fft_gain_compensation_db - app.fft_gain_compensation_db) <= TOL_DB
                or (np.isnan(sess.fft_gain_compensation_db) and np.isnan(app.fft_gain_compensation_db)))
        step = 2048 if channels == 2 else 1024
        for pos in range(step * 14, x.size + 4 * step, step * 3):
            res = sess.analyze_audio_file_samples(pos)
            ref = app.analyze_audio_file_samples(pos)
            for k in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status"):
                assert getattr(res, k) == ref[k], (channels, pos, k, getattr(res, k), ref[k])
            if ref["fft_ran"] and ref["mid_status"] == 0:
                assert db_close(sess.mid_fft[:, 1], app.mid_fft[:, 1], TOL_DB)
            if ref["fft_ran"] and ref["mid_status"] != 0:
                assert np.array_equal(sess.mid_fft, app.mid_fft)
        sess.close()


def test_file_session_short_file(oracle):
    """A file shorter than one window: every tick is skipped or falls back, like the reference."""
    from oracle.app_driver import FileApp
    x = make_stereo(5, 9000, level=0.4)
    sess = ssa.FileSession(x, 2, 48000)
    app = FileApp(x, 2, 48000)
    assert np.array_equal(sess.audio_file_chart, app.audio_file_chart)
    for pos in (0, 2048, 16384, 18000, 32768, 32770, 40000):
        res = sess.analyze_audio_file_samples(pos)
        ref = app.analyze_audio_file_samples(pos)
        check_tick(res, ref, sess, app)
    assert np.allclose(sess.lufs, app.lufs, atol=TOL_DB)


@pytest.mark.parametrize("rate,channels", [(48000, 2), (44100, 2), (48000, 1), (96000, 2), (32000, 2), (88200, 1)])
def test_capture_session_matches_reference_driver(oracle, rate, channels):
    from oracle.app_driver import CaptureApp
    sess = ssa.CaptureSession(channels, rate)
    app = CaptureApp(channels, rate)
    # a 30*rate-sample capture ring advancing by one callback's worth of samples per tick
    total = make_stereo(3 + rate, 15 * rate + 8 * 4096, rate=rate, level=0.5)
    n = 30 * rate
    for k in range(6):
        off = 2 * 4096 * k
        ring = total[off:off + n]
        res = sess.analyze_microphone_input(ring)
        ref = app.analyze_microphone_input(ring)
        for key in ("mid_status", "side_status", "add_status", "shortterm_status"):
            assert getattr(res, key) == ref[key], (key, getattr(res, key), ref[key])
        assert np.array_equal(sess.microphone_input_chart, app.microphone_input_chart)   # bit-exact
        assert db_close(sess.mid_fft[:, 1], app.mid_fft[:, 1], TOL_DB)
        assert db_close(sess.side_fft[:, 1], app.side_fft[:, 1], TOL_DB)
        assert lufs_close(res.shortterm, ref["shortterm"])
    assert np.allclose(sess.lufs, app.lufs, atol=TOL_DB)


def test_session_argument_errors():
    with pytest.raises(ssa.AnalyzerError):
        ssa.CaptureSession(2, 1000)                       # 15*rate < 16384: the reference's subtraction underflows
    sess = ssa.CaptureSession(2, 48000)
    with pytest.raises(ssa.AnalyzerError):
        sess.analyze_microphone_input(np.zeros(1000, np.float32))      # not the 30*rate ring


def test_tick_launch_soak_against_the_separate_calls():
    """Three thousand ticks of the one-launch file tick (k_tick: spectrum + loudness call + short-term reading, rows flagged early)
    against the same ticks through the Analyzer's own methods (separate kernels, separate reductions): the spectra are the same
    arithmetic on the same samples, the short-term loudness the same sum in another order."""
    rate = 48000
    x = make_stereo(4242, rate * 64, rate=rate, level=0.5)
    mid, side = ssa.get_mid_and_side_samples(x)
    sess = ssa.FileSession(x, 2, rate)
    an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
    worst_st = 0.0
    n = 0
    for pos in range(16384 * 2 + 2048, x.size + 1, 2048):
        res = sess.analyze_audio_file_samples(pos)
        pf = pos // 2
        an.add_samples(x[pos - 16384:pos])
        st = an.get_shortterm_lufs()
        assert res.fft_ran and res.fed and res.mid_status == 0 and res.side_status == 0
        if n % 7 == 0:                                           # (the spectra of every seventh tick: they cost the most here)
            m = an.get_fft(mid[pf - 16384:pf]); sd = an.get_fft(side[pf - 16384:pf])
            assert np.array_equal(sess.mid_fft, m), pos
            assert np.array_equal(sess.side_fft, sd), pos
        if np.isinf(st) or np.isinf(res.shortterm):
            assert st == res.shortterm
        else:
            worst_st = max(worst_st, abs(st - res.shortterm))
        n += 1
    assert n > 2900
    assert worst_st <= 1e-9, worst_st
    sess.close()


@pytest.mark.parametrize("seed", [0, 7, 19, 42, 228, 247, 273])
def test_randomised_tick_programme_against_the_restated_app(oracle, seed):
    """tools/fuzz_ticks.py's programmes (random file with level jumps, silences, NaN / infinite pairs; playback, seeks, positions
    past both ends, restarts), a few seeds of the 3100 it has been run on."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "fuzz_ticks", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_ticks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.programme(seed)
    assert r is None, r


def test_capture_session_non_finite_samples(oracle):
    """A NaN pair in the chart's part of the ring, then an infinite pair inside the newest window: the chart keeps the crate's
    min / max semantics, the spectra fall back like the reference's, the meter's reading goes NaN with it."""
    from oracle.app_driver import CaptureApp
    rate = 48000
    sess = ssa.CaptureSession(2, rate)
    app = CaptureApp(2, rate)
    base = make_stereo(77, 15 * rate, rate=rate, level=0.4)
    for case in range(3):
        ring = base.copy()
        if case >= 1: ring[2 * 100000] = np.nan                         # far from the newest window: only the chart sees it
        if case >= 2: ring[ring.size - 2 * 5000 + 1] = np.inf           # inside the newest 16384 pairs
        res = sess.analyze_microphone_input(ring)
        ref = app.analyze_microphone_input(ring)
        for key in ("mid_status", "side_status", "add_status", "shortterm_status"):
            assert getattr(res, key) == ref[key], (case, key, getattr(res, key), ref[key])
        assert np.array_equal(sess.microphone_input_chart, app.microphone_input_chart, equal_nan=True), case
        for got, want in ((sess.mid_fft, app.mid_fft), (sess.side_fft, app.side_fft)):
            assert got.shape == want.shape, case
            if want.shape[0] > 1:
                assert db_close(got[:, 1], want[:, 1], TOL_DB), case
            else:
                assert np.array_equal(got, want), case
        a, b = res.shortterm, ref["shortterm"]
        assert (np.isnan(a) and np.isnan(b)) or lufs_close(a, b), (case, a, b)


def test_file_session_many_non_finite_pairs_and_empty_file(oracle):
    """More non-finite pairs than the device-side index holds (the open falls back to the host's own scan), and a file without a
    single sample: statuses and fallbacks like the restated App's."""
    from oracle.app_driver import FileApp
    rate = 48000
    x = make_stereo(11, rate * 2, rate=rate, level=0.3)
    x[2 * 20000:2 * 20000 + 2 * 6000:2] = np.nan              # 6000 pairs with a NaN left sample
    x[2 * 50000 + 1:2 * 50000 + 2 * 3000:2] = np.inf          # 3000 pairs with an infinite right sample
    sess = ssa.FileSession(x, 2, rate); app = FileApp(x, 2, rate)
    assert np.array_equal(sess.audio_file_chart, app.audio_file_chart, equal_nan=True)
    for pos in range(2048 * 14, x.size + 2 * 2048, 2048 * 5):
        res = sess.analyze_audio_file_samples(pos); ref = app.analyze_audio_file_samples(pos)
        for key in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status"):
            assert getattr(res, key) == ref[key], (pos, key, getattr(res, key), ref[key])
        if ref["fft_ran"]:
            for got, want in ((sess.mid_fft, app.mid_fft), (sess.side_fft, app.side_fft)):
                assert got.shape == want.shape, pos
                if want.shape[0] > 1:
                    assert db_close(got[:, 1], want[:, 1], TOL_DB), pos
                else:
                    assert np.array_equal(got, want), pos
    sess.close()
    empty = np.zeros(0, np.float32)
    sess = ssa.FileSession(empty, 2, rate); app = FileApp(empty, 2, rate)
    assert sess.duration_ms == app.duration_ms
    assert sess.audio_file_chart.shape == app.audio_file_chart.shape
    for pos in (0, 2048, 40000):
        res = sess.analyze_audio_file_samples(pos); ref = app.analyze_audio_file_samples(pos)
        for key in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status"):
            assert getattr(res, key) == ref[key], (pos, key, getattr(res, key), ref[key])
    sess.close()


@pytest.mark.parametrize("rate,channels", [(48000, 2), (44100, 2), (48000, 1), (96000, 2)])
def test_capture_ring_resident_on_the_device(oracle, rate, channels):
    """The capture ring kept on the device: pushes of the sizes a capture callback delivers (and one longer than the whole ring, one
    after a long pause without a tick), a tick after every few pushes — against the restated App on the ring the reference would
    hold (zeros, then everything pushed, the oldest dropped), and against the snapshot form of the same tick."""
    from oracle.app_driver import CaptureApp
    rng = np.random.default_rng(5 + rate + channels)
    n = 30 * rate
    sess = ssa.CaptureSession(channels, rate)
    snap = ssa.CaptureSession(channels, rate)
    app = CaptureApp(channels, rate)
    ring = np.zeros(n, np.float32)                          # the reference's ring: starts full of zeros (tui.rs:1783-1784)
    src = make_stereo(9 + rate, 40 * rate, rate=rate, level=0.5)
    at = 0

    def push(k):
        nonlocal ring, at
        x = src[at:at + k]; at += k
        sess.push(x)
        ring = x[-n:].copy() if k >= n else np.concatenate([ring[k:], x])

    plan = [[960, 960, 964], [4096], [2, 1022, 3000], [n + 1234], [512] * 5, [7 * rate, 9 * rate, 3 * rate], [480, 480]]
    if rate == 96000: plan = plan[:5]
    for pushes in plan:
        for k in pushes: push(k)
        res = sess.analyze_resident()
        ref = app.analyze_microphone_input(ring)
        res2 = snap.analyze_microphone_input(ring)
        for key in ("mid_status", "side_status", "add_status", "shortterm_status"):
            assert getattr(res, key) == ref[key] == getattr(res2, key), (key, getattr(res, key), ref[key])
        assert np.array_equal(sess.microphone_input_chart, app.microphone_input_chart)       # bit-exact
        assert np.array_equal(sess.microphone_input_chart, snap.microphone_input_chart)
        assert np.array_equal(sess.mid_fft, snap.mid_fft) and np.array_equal(sess.side_fft, snap.side_fft)
        assert db_close(sess.mid_fft[:, 1], app.mid_fft[:, 1], TOL_DB)
        assert db_close(sess.side_fft[:, 1], app.side_fft[:, 1], TOL_DB)
        assert lufs_close(res.shortterm, ref["shortterm"])
        assert res.shortterm == res2.shortterm or abs(res.shortterm - res2.shortterm) <= 1e-9
    assert np.allclose(sess.lufs, app.lufs, atol=TOL_DB)
    # mixing the forms: a snapshot tick replaces the ring, pushes go on from there
    other = make_stereo(77, 15 * rate, rate=rate, level=0.3)
    sess.analyze_microphone_input(other); app.analyze_microphone_input(other); ring = other.copy()
    push(2048)
    res = sess.analyze_resident(); ref = app.analyze_microphone_input(ring)
    assert np.array_equal(sess.microphone_input_chart, app.microphone_input_chart)
    assert db_close(sess.mid_fft[:, 1], app.mid_fft[:, 1], TOL_DB)
    assert lufs_close(res.shortterm, ref["shortterm"])


@pytest.mark.parametrize("seed", [1, 5, 12, 33])
def test_randomised_capture_programme_against_the_restated_app(oracle, seed):
    """tools/fuzz_capture.py's programmes (a device-resident ring fed by pushes of random sizes, snapshot ticks, restarts, NaNs)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "fuzz_capture", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_capture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.programme(seed)
    assert r is None, r


def test_two_sessions_ticking_from_two_threads():
    """Two file sessions driven from two threads at once (ctypes releases the GIL during the calls): each reproduces, bit for bit,
    the readings it gives when it runs alone — the table caches, the stream pool and the kernels' one-time set-up are shared."""
    import threading
    rate = 48000
    files = [make_stereo(100 + i, rate * 6, rate=rate, level=0.3 + 0.2 * i) for i in range(2)]
    positions = list(range(16384 * 2 + 2048, files[0].size, 2048))

    def run(x, out):
        sess = ssa.FileSession(x, 2, rate)
        for pos in positions:
            res = sess.analyze_audio_file_samples(pos)
            out.append((res.shortterm, float(sess.mid_fft[:, 1].sum()), float(sess.side_fft[:, 1].sum())))
        out.append((sess.analyzer.get_integrated_lufs(),) + tuple(sess.analyzer.get_true_peak()))
        sess.close()

    alone = [[], []]
    for i in range(2): run(files[i], alone[i])
    both = [[], []]
    th = [threading.Thread(target=run, args=(files[i], both[i])) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(2):
        assert len(both[i]) == len(alone[i])
        assert both[i] == alone[i], next((k, a, b) for k, (a, b) in enumerate(zip(alone[i], both[i])) if a != b)
