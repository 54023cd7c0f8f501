#!/usr/bin/env python3
"""Randomised render-side reductions (tui.rs:49-51, :664-681, :801-821) against the restatement oracle/render.py: spectrum
rows -> chart columns (1 .. 5000 columns, the reference's gain or a number, N = 4096 / 16384, three rates), the decimated waveform ->
columns over random x ranges (inside, across and beyond the data), the Player-mode view bounds.
python tools/fuzz_render.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from soundscope_amd.batch import waveform_view
from oracle import render as R
from conftest import make_stereo


def programme(seed):
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([44100, 48000, 96000]))
    fft_n = int(rng.choice([4096, 4096, 16384]))
    ns = int(rng.integers(1, 5))
    frames = int(rate * float(np.exp(rng.uniform(np.log(0.45), np.log(5.0))))) + int(rng.integers(0, 999))
    cols = int(rng.choice([1, 2, 80, 160, 512, 1000, 4000, int(rng.integers(1, 5001))]))
    gain = None if rng.random() < 0.5 else float(rng.uniform(-80.0, 40.0))
    what = f"seed {seed}: {rate} Hz, N {fft_n}, {ns} streams x {frames} frames, {cols} columns, gain {gain}"
    xs = [make_stereo(seed * 3 + s, frames, rate=rate, level=float(rng.uniform(0.001, 0.9)), gap=bool(rng.random() < 0.3)) for s in range(ns)]
    if rng.random() < 0.2: xs[0][1::2] = xs[0][0::2]
    b = ssa.Batch(rate, 2, ns, frames, fft_n, 1024, flags=L.SS_BATCH_ALL)
    b.upload(0, np.concatenate(xs)); b.run()
    notes = []
    if b.layout.n_windows:
        b.render_spectrum(cols, gain)
        res = b.results()
        chart_x, _, _ = b.bin_tables()
        for s in range(ns):
            g = R.gain_db(res[s].integrated_lufs) if gain is None else gain
            if not np.isfinite(g): continue                  # (an all-silent stream: -13 - (-inf))
            rows, got = b.fft(s), b.spectrum_columns(s)
            for w in sorted(set([0, rows.shape[0] // 2, rows.shape[0] - 1])):
                for ch in (0, 1):
                    want = R.spectrum_columns(np.stack([chart_x, rows[w, ch].astype(np.float64)], 1), g, cols)
                    if not np.array_equal(np.isnan(got[w, ch]), np.isnan(want)): notes.append(f"stream {s} window {w} row {ch}: empty columns differ"); continue
                    ok = ~np.isnan(want)
                    if ok.any() and not np.abs(got[w, ch][ok] - want[ok]).max() <= 1e-3: notes.append(f"stream {s} window {w} row {ch}: {np.abs(got[w, ch][ok] - want[ok]).max()}")
    pts = b.layout.n_wave_points
    for _ in range(4):
        playhead, window_s = float(rng.uniform(-500, frames / rate * 1000 + 500)), float(rng.choice([0.001, 0.5, 1.5, 15.0, float(rng.uniform(0.01, 20))]))
        if waveform_view(playhead, window_s, pts) != R.waveform_view(playhead, window_s, pts): notes.append(f"waveform_view({playhead}, {window_s}, {pts})")
        lo, hi = R.waveform_view(playhead, window_s, pts)
        x_min, x_max = (int(lo), int(np.ceil(hi))) if rng.random() < 0.5 else (int(rng.integers(0, pts // 2 + 50)), int(rng.integers(0, pts // 2 + 200)))
        wc = int(rng.choice([1, 7, 80, 500, int(rng.integers(1, 3000))]))
        try:
            b.render_waveform(wc, x_min, x_max)
        except ssa.AnalyzerError as e:
            if not (x_max <= x_min and e.code == L.SS_ERR_INVALID_ARG): notes.append(f"render_waveform({wc}, {x_min}, {x_max}) status {e.code}")
            continue
        if x_max <= x_min: notes.append(f"render_waveform({wc}, {x_min}, {x_max}) accepted"); continue
        for s in range(ns):
            mm = b.waveform(s)
            chart = np.stack([np.repeat(np.arange(mm.shape[0]), 2), mm.reshape(-1).astype(np.float64)], 1)
            want = R.waveform_columns(chart, x_min, x_max, wc)
            if not np.array_equal(b.waveform_columns(s).astype(np.float64), want, equal_nan=True): notes.append(f"stream {s}: waveform columns ({wc}, {x_min}, {x_max}) of {mm.shape[0]} bins")
    b.close()
    return not notes, what + ("" if not notes else " -> " + "; ".join(notes[:5]))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failed = 0
    for seed in range(first, first + n):
        try:
            ok, msg = programme(seed)
        except Exception as e:                               # noqa: BLE001
            import traceback
            ok, msg = False, f"seed {seed}: exception {type(e).__name__}: {e} @ {traceback.extract_tb(e.__traceback__)[-1].lineno}"
        failed += 0 if ok else 1
        if not ok or "-v" in sys.argv:
            print(("ok   " if ok else "FAIL ") + msg, flush=True)
    print(f"{n} render programmes, {failed} failed")
    sys.exit(1 if failed else 0)
