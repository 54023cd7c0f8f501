#!/usr/bin/env python3
"""A spectrum row a fuzz_batch programme flagged, looked at from three sides: the device, the oracle's f32 radix-2 transform and an f64 rfft
(numpy) of the same Hann-windowed samples.  Who is how far from the truth, bin by bin.   python tools/probe_fuzz_row.py <seed> <kind> <window> <row>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
sys.argv_saved = sys.argv[:]
seed, kind, w, row = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sys.argv = [sys.argv[0]] + [a for a in sys.argv[5:]]
import fuzz_batch as fb
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
P = fb.plan(seed)
rate, ch, slot, fft_n, hop = P["rate"], P["ch"], P["slot"], P["fft_n"], P["hop"]
x = P["content"][kind]
b = ssa.Batch(rate, ch, 1, slot, fft_n, hop, flags=L.SS_BATCH_FFT)
b.upload(0, x); b.run(); b.sync()
fft = b.fft(0)
sig = po.mid_side(x) if ch == 2 else [np.ascontiguousarray(x.reshape(slot, ch)[:, c]) for c in range(ch)]
first, nb = None, None
nbins, _ = po.fft_bins(rate, fft_n)
for ww in ([w] if w >= 0 else range(fft.shape[0])):
    start = (ww + fft_n // hop + 1) * hop - fft_n
    s = sig[row][start:start + fft_n]
    try: ref = po.get_fft(rate, s)
    except po.OracleError: continue
    freq = None
    hw = po.hann_window(s).astype(np.float64)
    X = np.fft.rfft(hw)
    k = np.arange(X.size); fr = k * (np.float32(rate) / np.float32(fft_n))
    keep = (fr >= 20) & (fr <= 20000)
    mag = np.abs(X[keep])
    f = fr[keep].astype(np.float64)
    with np.errstate(divide="ignore"):
        truth = np.where(mag == 0, -150.0, 20 * np.log10(mag * 4 / fft_n)) + 10 * np.log10(f / 1000.0)
    got, orc = fft[ww, row].astype(np.float64), ref[:, 1]
    peak = truth.max(); strong = truth >= peak - 70
    dg, do, dd = np.abs(got - truth)[strong].max(), np.abs(orc - truth)[strong].max(), np.abs(got - orc)[strong].max()
    if w >= 0 or dd > 0.01:
        kk = int(np.argmax(np.where(strong, np.abs(got - orc), 0)))
        print(f"window {ww} row {row}: peak {peak:.1f} dB; within 70 dB of it: device-f64 {dg:.4f} dB, oracle-f64 {do:.4f} dB, device-oracle {dd:.4f} dB at bin {kk} ({truth[kk]:.1f} dB: device {got[kk]:.4f} oracle {orc[kk]:.4f} f64 {truth[kk]:.4f})")
