#!/usr/bin/env python3
"""Randomised columns-only spectra (SS_BATCH_FFT_COLUMNS, the render-side reduction fused into the spectrum kernel) against the
two-pass result (full rows + ss_batch_render_spectrum), bit for bit: random column count (1 .. 512), gain (the reference's from the
integrated loudness, or a number), rate (the bin -> column tables change with it), stream count and length, level jumps, a few NaN /
infinite samples.      python tools/fuzz_columns.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo


def programme(seed):
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([44100, 48000, 48000, 88200, 96000, 192000]))
    cols = int(rng.choice([1, 2, 3, 7, 64, 100, 160, 255, 256, 511, 512, int(rng.integers(1, 513)), int(rng.integers(1, 513))]))
    gain = None if rng.random() < 0.4 else float(rng.uniform(-60.0, 30.0))
    ns = int(rng.choice([1, 2, 5, 33, 70]))
    frames = int(rate * float(np.exp(rng.uniform(np.log(0.12), np.log(6.0))))) + int(rng.integers(0, 1024))
    what = f"seed {seed}: {rate} Hz, {cols} columns, gain {gain}, {ns} streams x {frames} frames"
    kinds = min(ns, 3)
    xs = []
    for k in range(kinds):
        x = make_stereo(seed * 5 + k, frames, rate=rate, level=float(rng.uniform(0.001, 0.9)), gap=bool(rng.random() < 0.3))
        r = rng.random()
        if r < 0.15: x[1::2] = x[0::2]                      # dual mono: the side row is the -150 dB floor
        elif r < 0.25: x[2 * int(rng.integers(0, frames))] = np.nan
        elif r < 0.35: x[2 * int(rng.integers(0, frames)) + 1] = np.inf
        elif r < 0.45: x[: 2 * (frames // 2)] = 0.0         # leading digital silence
        xs.append(x)
    buf = np.concatenate([xs[i % kinds] for i in range(ns)])
    try:
        two = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
        one = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
    except ssa.AnalyzerError as e:
        return "device" not in str(e).lower(), what + f" -> refused at create ({e})"
    two.upload(0, buf); two.run(); two.render_spectrum(cols, gain)
    one.upload(0, buf); one.set_columns_gain(gain)
    notes = []
    for rep in range(2):
        one.run(); one.sync()
        for s in sorted(set([0, ns - 1, int(rng.integers(0, ns))])):
            a, b = one.spectrum_columns(s), two.spectrum_columns(s)
            if a.shape != b.shape: notes.append(f"stream {s}: shape {a.shape} vs {b.shape}"); continue
            neq = ~((a == b) | (np.isnan(a) & np.isnan(b)))
            if neq.any():
                w, c, k = [int(v[0]) for v in np.nonzero(neq)]
                notes.append(f"pass {rep} stream {s}: {int(neq.sum())} of {a.size} differ, first at window {w} row {c} column {k}: {a[w, c, k]} vs {b[w, c, k]}")
    one.close(); two.close()
    return not notes, what + ("" if not notes else " -> " + "; ".join(notes[:4]))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failed = 0
    for seed in range(first, first + n):
        try:
            ok, msg = programme(seed)
        except Exception as e:                               # noqa: BLE001
            ok, msg = False, f"seed {seed}: exception {type(e).__name__}: {e}"
        failed += 0 if ok else 1
        if not ok or "-v" in sys.argv or "refused" in msg:
            print(("ok   " if ok else "FAIL ") + msg, flush=True)
    print(f"{n} columns programmes, {failed} failed")
    sys.exit(1 if failed else 0)
