#!/usr/bin/env python3
"""Dual-mono material (L == R: the side signal is exactly zero; L == -R: the mid signal is): what do the packed mid/side
spectrum kernels return for the empty row?  The reference transforms an all-zero buffer there: -150 dB + pink everywhere."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
rate, frames = 48000, 48000 * 2
rng = np.random.default_rng(1)
t = np.arange(frames) / rate
m = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.uniform(-1, 1, frames)).astype(np.float32)
for n in (4096, 16384):
    for name, l, r in (("L == R", m, m), ("L == -R", m, -m)):
        x = np.empty(2 * frames, np.float32); x[0::2] = l; x[1::2] = r
        b = ssa.Batch(rate, 2, 1, frames, n, 1024, flags=L.SS_BATCH_FFT)
        b.upload(0, x); b.run(); b.sync()
        got = b.fft(0)
        ref = po.analyze_stream(rate, x, n, 1024)["fft"]
        empty = 1 if name == "L == R" else 0
        print(f"N={n} {name}: empty row ref [{ref[:, empty].min():.2f}, {ref[:, empty].max():.2f}] dB, device [{got[:, empty].min():.2f}, {got[:, empty].max():.2f}] dB;"
              f" full row max |d| {np.abs(got[:, 1 - empty] - ref[:, 1 - empty])[ref[:, 1 - empty] > ref[:, 1 - empty].max() - 70].max():.4f} dB", flush=True)
        b.close()
    a = ssa.Analyzer(); a.create_loudness_meter(2, rate)
    z = a.get_fft(np.zeros(n, np.float32))
    print(f"N={n} ss_get_fft(zeros): dB range [{z[:, 1].min():.2f}, {z[:, 1].max():.2f}]")
