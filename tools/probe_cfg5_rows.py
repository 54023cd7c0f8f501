#!/usr/bin/env python3
"""Config 5 (96 kHz, 8 channels, N = 16384): every window of every channel of one stream against the oracle at both spectrum metrics
(conftest.db_close: relative to the row's own peak; db_close_survey: SURVEY section 7's absolute wording), and — for the rows that miss
one of them — against an f64 transform of the same Hann-windowed samples: whose rounding is it.   python tools/probe_cfg5_rows.py [stream]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
from conftest import db_close, db_close_survey, db_report

stream = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rate, ch, frames, ns, N = 96000, 8, 960000, 64, 16384
b = ssa.Batch(rate, ch, ns, frames, N, 1024, flags=L.SS_BATCH_ALL, true_peak_factor=4)
b.synthesize(0x5EED0000, 0)
b.run(); b.sync()
lay = b.layout
x = b.download_input(stream).reshape(frames, ch)
cols = [np.ascontiguousarray(x[:, c]) for c in range(ch)]
fft = b.fft(stream)
jobs = [(w, c) for w in range(lay.n_windows) for c in range(ch)]


def truth_row(s):
    hw = po.hann_window(s).astype(np.float64)
    X = np.fft.rfft(hw)
    k = np.arange(X.size); fr = k * (np.float32(rate) / np.float32(N))
    keep = (fr >= 20) & (fr <= 20000)
    mag = np.abs(X[keep]); f = fr[keep].astype(np.float64)
    with np.errstate(divide="ignore"):
        return np.where(mag == 0, -150.0, 20 * np.log10(mag * 4 / N)) + 10 * np.log10(f / 1000.0)


def row(job):
    w, c = job
    start = (w + 1) * 1024
    s = cols[c][start:start + N]
    ref = po.get_fft(rate, s)[:, 1]
    got = fft[w, c]
    a = db_close(got, ref, 0.01)
    sv = db_close_survey(got, ref, 0.01)
    if a and sv:
        return None
    t = truth_row(s)
    g = got.astype(np.float64)
    loud = ref >= -90.0
    d = np.where(loud, np.abs(g - ref), 0)
    k = int(np.argmax(d))
    return (w, c, a, sv, float(ref.max()), k, float(ref[k]), float(g[k]), float(t[k]), db_report(got, ref))


with ThreadPoolExecutor(max(1, min(64, len(os.sched_getaffinity(0))))) as ex:
    res = [r for r in ex.map(row, jobs, chunksize=64) if r is not None]
print(f"stream {stream}: {len(jobs)} rows, {len(res)} miss a metric ({sum(1 for r in res if not r[2])} the row-peak one, {sum(1 for r in res if not r[3])} the survey one)")
for r in res[:40]:
    w, c, a, sv, peak, k, rk, gk, tk, rep = r
    print(f"  w {w:3d} ch {c}: row-peak {'ok' if a else 'MISS'} survey {'ok' if sv else 'MISS'}; peak {peak:7.2f} dB; worst loud bin {k}: oracle {rk:9.4f} device {gk:9.4f} f64 {tk:9.4f}"
          f"  (device-oracle {abs(gk - rk):.4f}, device-f64 {abs(gk - tk):.4f}, oracle-f64 {abs(rk - tk):.4f}); db_report {rep[0]:.4f} dB / {rep[1]:.2e}")
