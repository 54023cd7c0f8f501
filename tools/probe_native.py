#!/usr/bin/env python3
"""The reference's native cadence in a batch: stereo, N = 16384, hop 1024 (tui.rs:1488), spectrum only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for rate in (48000, 44100):
    b = ssa.Batch(rate, 2, streams, rate * 10, 16384, 1024, flags=L.SS_BATCH_FFT)
    b.synthesize(7, 0)
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(3):
        b.run(); b.sync()
    ms, n = b.timing_read(L.SS_KERNEL_FFT)
    lay = b.layout
    alg = streams * (rate * 10 * 2 * 4 + lay.n_windows * 2 * lay.n_bins * 4)
    print(f"{rate} Hz x {streams} streams: spectrum {ms / n:.3f} ms, windows {lay.n_windows} bins {lay.n_bins}, "
          f"{streams * rate * 10 * 2 / (ms / n) / 1e6:.1f} Gsamples/s, algorithmic {alg / (ms / n) / 1e6:.0f} GB/s")
    b.close()
