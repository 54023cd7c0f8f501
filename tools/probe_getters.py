#!/usr/bin/env python3
"""Latency of the handle's readings behind add_samples (analyzer.rs:139-164 as the render loop calls them, tui.rs:917-969):
add_samples(2048 samples) then get_integrated_lufs + get_loudness_range + get_true_peak + get_shortterm_lufs, wall clock per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from conftest import make_stereo
rate = 48000
x = make_stereo(5, rate * 14, rate)
an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
an.add_samples(x[:rate * 2 * 4])
t = {k: [] for k in ("add_samples", "integrated", "range", "true_peak", "shortterm", "all")}
pos = rate * 2 * 4
for it in range(400):
    c = x[pos:pos + 2048]; pos += 2048
    t0 = time.perf_counter(); an.add_samples(c)
    t1 = time.perf_counter(); i = an.get_integrated_lufs()
    t2 = time.perf_counter(); r = an.get_loudness_range()
    t3 = time.perf_counter(); p = an.get_true_peak()
    t4 = time.perf_counter(); s = an.get_shortterm_lufs()
    t5 = time.perf_counter()
    if it >= 20:
        for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)): t[k].append(v)
for k, v in t.items():
    print(f"{k:12s} median {np.median(v) * 1e6:7.1f} us   p90 {np.percentile(v, 90) * 1e6:7.1f} us")
print("readings:", i, r, p, s)
# calculate_integrated_lufs on a 10 s file (one kept loudness-only batch: upload + k_time_domain + k_finalize + read-back)
y = x[:rate * 2 * 10]
for _ in range(3): v = an.calculate_integrated_lufs(2, y)
t0 = time.perf_counter()
for _ in range(20): v = an.calculate_integrated_lufs(2, y)
print(f"calculate_integrated_lufs(10 s stereo): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms  -> {v}")
