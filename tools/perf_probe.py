#!/usr/bin/env python3
"""Quick per-kernel timing probe on the bench workload (no torch): python tools/perf_probe.py [streams] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
flags = L.SS_BATCH_ALL
for a in sys.argv:
    if a.startswith('--flags='):
        flags = int(a.split('=')[1])
cols = 0
for a in sys.argv:
    if a.startswith('--cols='):
        cols = int(a.split('=')[1])
b = ssa.Batch(48000, 2, streams, 480000, 4096, 1024, flags=flags, spectrum_columns=cols)
b.synthesize(0x5EED0000, 0)
if "--tp-f16" in sys.argv:          # true peak as the opt-in f16x3 split (the default is the reference's f32 width)
    b.set_true_peak_arith(L.SS_TP_ARITH_F16X3)
for _ in range(2):
    b.run(); b.sync()
b.timing_enable(True)
for _ in range(steps):
    b.run(); b.sync()
tot = 0.0
for k in range(L.SS_KERNEL_COUNT):
    ms, n = b.timing_read(k)
    tot += ms / max(n, 1)
    print(f"{L.lib().ss_batch_kernel_name(b._h, k).decode():16s} {ms / max(n, 1):9.4f} ms")
lay = b.layout
alg = streams * (480000 * 2 * 4 + lay.n_windows * 2 * (cols if cols else lay.n_bins) * 4)
fft_ms = b.timing_read(0)[0] / steps
print(f"sum {tot:.4f} ms  -> {streams * 960000 / tot / 1e6:.1f} Gsamples/s;  FFT roofline {alg / fft_ms / 1e6 / 8000:.3f}")
if "--check" in sys.argv and not cols:
    from oracle import pyoracle as po
    x = b.download_input(1)
    ref = po.analyze_stream(48000, x, 4096, 1024)
    r = b.results()[1]
    f = b.fft(1)
    strong = ref["fft"] > -90
    print("fft max err dB", np.abs(f[strong] - ref["fft"][strong]).max(), "I", r.integrated_lufs, ref["integrated"],
          "TP", r.true_peak[0], ref["true_peak"][0], "wave", np.array_equal(b.waveform(1).reshape(-1), ref["wave"][:, 1].astype(np.float32)))
