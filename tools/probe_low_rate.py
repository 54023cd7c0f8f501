#!/usr/bin/env python3
"""Loudness at absurdly low rates (the crate accepts 16 Hz and up: a 100 ms sub-block is (rate + 5) / 10 = 2 frames at 16 Hz): the batch
path, the handle path and the oracle on the same mono noise; the batch's sub-block energies against scipy's f64 filter."""
import os, sys
import numpy as np
import scipy.signal as sg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po


def design(rate):
    f0, G, Q = 1681.974450955533, 3.999843853973347, 0.7071752369554196
    K = np.tan(np.pi * f0 / rate); Vh = 10 ** (G / 20); Vb = Vh ** 0.4996667741545416
    a0 = 1 + K / Q + K * K
    pb = [(Vh + Vb * K / Q + K * K) / a0, 2 * (K * K - Vh) / a0, (Vh - Vb * K / Q + K * K) / a0]; pa = [1, 2 * (K * K - 1) / a0, (1 - K / Q + K * K) / a0]
    f0, Q = 38.13547087602444, 0.5003270373238773; K = np.tan(np.pi * f0 / rate)
    rb = [1, -2, 1]; ra = [1, 2 * (K * K - 1) / (1 + K / Q + K * K), (1 - K / Q + K * K) / (1 + K / Q + K * K)]
    return np.convolve(pb, rb), np.convolve(pa, ra)


for rate, frames in ((16, 4610), (16, 490), (20, 3000), (31, 3000), (64, 5000), (199, 5000), (8000, 20000)):
    rng = np.random.default_rng(rate)
    x = (0.3 * rng.uniform(-1, 1, frames)).astype(np.float32)
    an = ssa.Analyzer(); an.create_loudness_meter(1, rate)
    one = an.calculate_integrated_lufs(1, x)
    an.add_samples(x); h = an.get_integrated_lufs()
    ref = po.calculate_integrated_lufs(rate, 1, x)
    S = (rate + 5) // 10
    b = ssa.Batch(rate, 1, 1, frames, 4096, 1024, flags=L.SS_BATCH_LUFS)
    b.upload(0, x); b.run(); b.sync()
    g = b.geometry
    sub = b.subblocks(0)[:, 0]
    bb, aa = design(rate)
    y = sg.lfilter(bb, aa, x.astype(np.float64))
    nsub = frames // S
    e = (y[:nsub * S].reshape(nsub, S) ** 2).sum(1)
    n = min(nsub, sub.size)
    rel = np.abs(sub[:n] - e[:n]) / np.maximum(e[:n], 1e-300)
    badi = np.nonzero(rel > 1e-6)[0]
    print(f"rate {rate} frames {frames} S {S}: one-shot {one} handle {h} oracle {ref} batch {b.results()[0].integrated_lufs}; segments {g.td_segments} x {g.td_segment_subblocks} split {g.td_split};"
          f" sub-blocks {n}, wrong {badi.size}, first {badi[:5]}, max rel {rel.max() if n else 0:.2e}")
    an.close(); b.close()
