#!/usr/bin/env python3
"""Randomised tick programmes against the restated App (oracle/app_driver.py): a file of random rate, length and content (level
jumps, a silence, now and then a NaN or an infinite pair), then a few dozen ticks at random positions — forward playback at the
player's hop, seeks backwards and forwards, positions at and past both ends, a restart in between; statuses, the two spectra,
the short-term loudness and the 300-entry history after every tick.      python tools/fuzz_ticks.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from conftest import db_close

def programme(seed):
    from oracle.app_driver import FileApp
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([32000, 44100, 48000, 48000, 88200, 96000, 192000]))
    secs = float(rng.uniform(0.2, 3.5))
    frames = int(rate * secs)
    t = np.arange(frames) / rate
    x = np.empty((frames, 2), np.float32)
    lev = 10.0 ** (rng.uniform(-50, -3) / 20.0)
    env = np.where(t > rng.uniform(0, secs), 10.0 ** (rng.uniform(-40, 0) / 20.0), 1.0)
    for c in range(2):
        x[:, c] = lev * env * (np.sin(2 * np.pi * rng.uniform(50, 4000) * t + c) + 0.2 * rng.standard_normal(frames))
    if rng.integers(0, 4) == 0:
        a = int(rng.integers(0, frames)); x[a:a + int(rng.integers(1, rate // 2))] = 0.0
    x = np.ascontiguousarray(x.reshape(-1))
    if rng.integers(0, 5) == 0 and x.size > 10: x[int(rng.integers(0, x.size))] = np.nan
    if rng.integers(0, 5) == 0 and x.size > 10: x[int(rng.integers(0, x.size))] = np.inf
    if rng.integers(0, 6) == 0: x = x[:-1].copy()
    sess = ssa.FileSession(x, 2, rate); app = FileApp(x, 2, rate)
    if not np.array_equal(sess.audio_file_chart, app.audio_file_chart, equal_nan=True): return f"seed {seed}: chart differs"
    pos = int(rng.integers(0, 4)) * 2048
    for k in range(int(rng.integers(10, 60))):
        mode = rng.integers(0, 10)
        if mode < 6: pos += 2048
        elif mode == 6: pos = int(rng.integers(0, x.size + 3 * 2048))
        elif mode == 7: pos = max(0, pos - int(rng.integers(1, 40)) * 2048)
        elif mode == 8: pos = x.size + int(rng.integers(-2, 3)) * 2048
        else:
            sess.restart(); app.restart()
        pos = max(0, pos)
        res = sess.analyze_audio_file_samples(pos); ref = app.analyze_audio_file_samples(pos)
        for key in ("fft_ran", "mid_status", "side_status", "lufs_ran", "fed", "add_status", "shortterm_status"):
            if getattr(res, key) != ref[key]: return f"seed {seed} ({rate} Hz, {secs:.2f} s) tick {k} pos {pos}: {key} {getattr(res, key)} vs {ref[key]}"
        if ref["fft_ran"]:
            for got, want, nm in ((sess.mid_fft, app.mid_fft, "mid"), (sess.side_fft, app.side_fft, "side")):
                if got.shape != want.shape: return f"seed {seed} tick {k} pos {pos}: {nm} shape {got.shape} vs {want.shape}"
                if want.shape[0] > 1:
                    if not (np.array_equal(got[:, 0], want[:, 0]) and db_close(got[:, 1], want[:, 1], 0.01)):
                        return f"seed {seed} ({rate} Hz) tick {k} pos {pos}: {nm} spectrum differs (max {np.nanmax(np.abs(got[:, 1] - want[:, 1])):.4f} dB)"
                elif not np.array_equal(got, want): return f"seed {seed} tick {k} pos {pos}: {nm} fallback differs"
        a, b = res.shortterm, ref["shortterm"]
        same = (np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= 1e-6 + 1e-8 * abs(b)
        if not same:
            return f"seed {seed} ({rate} Hz) tick {k} pos {pos}: short-term {a} vs {b}"
        if not np.allclose(sess.lufs, app.lufs, atol=1e-6, rtol=1e-8, equal_nan=True): return f"seed {seed} tick {k}: history differs"
    sess.close()
    return None

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n):
        r = programme(seed)
        if r: print("FAIL", r, flush=True); bad += 1
    print(f"{n} tick programmes, {bad} failed")
