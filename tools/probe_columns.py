#!/usr/bin/env python3
"""Columns-only spectrum (SS_BATCH_FFT_COLUMNS) at the bench shape: per-kernel times beside the ordinary pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
for cols in (0, 160, 512):
    b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
    b.synthesize(0x5EED0000, 0)
    for _ in range(2):
        b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(10):
        b.run(); b.sync()
    ks = [b.timing_read(k) for k in range(L.SS_KERNEL_COUNT)]
    ms = [m / max(n, 1) for m, n in ks]
    flops = 5.0 * 4096 * 12 * 1024 * 464
    print(f"cols={cols:3d}: spectrum {ms[0]:.4f} ms ({flops / ms[0] / 1e9 / 157.3e3:.3f} of the fp32 vector peak by 5 N log2 N), time domain {ms[1]:.4f}, finalize {ms[2]:.4f}; "
          f"pass {sum(ms):.4f} ms -> {1024 * 960000 / sum(ms) / 1e6:.1f} Gsamples/s", flush=True)
    b.close()
