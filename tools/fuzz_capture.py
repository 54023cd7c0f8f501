#!/usr/bin/env python3
"""Randomised capture programmes: a device-resident capture ring fed by pushes of random sizes (one sample to more than the whole
ring), ticks in between, now and then a snapshot tick or a restart — against the restated App on the ring the reference would hold.
python tools/fuzz_capture.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from conftest import db_close

def programme(seed):
    from oracle.app_driver import CaptureApp
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([32000, 44100, 48000, 48000, 96000]))
    ch = int(rng.choice([1, 2, 2]))
    n = 30 * rate
    sess = ssa.CaptureSession(ch, rate); app = CaptureApp(ch, rate)
    ring = np.zeros(n, np.float32)
    t0 = 0
    for k in range(int(rng.integers(4, 14))):
        for _ in range(int(rng.integers(0, 5))):
            kind = rng.integers(0, 8)
            m = int(rng.integers(1, 64)) if kind == 0 else int(rng.integers(64, 8192)) if kind < 6 else int(rng.integers(8192, 3 * rate)) if kind == 6 else n + int(rng.integers(0, 5000))
            t = (np.arange(m) + t0) / rate; t0 += m
            x = (10.0 ** (rng.uniform(-50, -3) / 20.0) * (np.sin(2 * np.pi * rng.uniform(60, 5000) * t) + 0.2 * rng.standard_normal(m))).astype(np.float32)
            if rng.integers(0, 25) == 0: x[int(rng.integers(0, m))] = np.nan
            sess.push(x)
            ring = x[-n:].copy() if m >= n else np.concatenate([ring[m:], x])
        mode = rng.integers(0, 10)
        if mode == 0:
            sess.restart(); app.restart()
        if mode == 1:
            res = sess.analyze_microphone_input(ring)
        else:
            res = sess.analyze_resident()
        ref = app.analyze_microphone_input(ring)
        for key in ("mid_status", "side_status", "add_status", "shortterm_status"):
            if getattr(res, key) != ref[key]: return f"seed {seed} ({rate} Hz, {ch} ch) tick {k}: {key} {getattr(res, key)} vs {ref[key]}"
        if not np.array_equal(sess.microphone_input_chart, app.microphone_input_chart, equal_nan=True): return f"seed {seed} tick {k}: chart differs"
        for got, want, nm in ((sess.mid_fft, app.mid_fft, "mid"), (sess.side_fft, app.side_fft, "side")):
            if got.shape != want.shape: return f"seed {seed} tick {k}: {nm} shape"
            if want.shape[0] > 1:
                if not db_close(got[:, 1], want[:, 1], 0.01): return f"seed {seed} ({rate} Hz) tick {k}: {nm} spectrum differs"
            elif not np.array_equal(got, want): return f"seed {seed} tick {k}: {nm} fallback differs"
        a, b = res.shortterm, ref["shortterm"]
        if not ((np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= 1e-6 + 1e-8 * abs(b)): return f"seed {seed} ({rate} Hz) tick {k}: short-term {a} vs {b}"
    sess.close()
    return None

if __name__ == "__main__":
    cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + cnt):
        r = programme(seed)
        if r: print("FAIL", r, flush=True); bad += 1
    print(f"{cnt} capture programmes, {bad} failed")
