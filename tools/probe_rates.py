#!/usr/bin/env python3
"""The bench shape at other sample rates (1024 stereo streams x 10 s, N = 4096, hop 1024): per-kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for rate in (44100, 48000, 88200, 96000):
    frames = rate * 10
    b = ssa.Batch(rate, 2, streams, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(3):
        b.run(); b.sync()
    t = [b.timing_read(k) for k in range(L.SS_KERNEL_COUNT)]
    ms = [a / max(n, 1) for a, n in t]
    lay = b.layout
    print(f"{rate:6d} Hz: spectrum {ms[0]:.3f} ms ({lay.n_windows} windows x {lay.n_bins} bins), time domain {ms[1]:.3f}, finalize {ms[2]:.3f}, "
          f"waveform {ms[3]:.3f} -> {streams * frames * 2 / sum(ms) / 1e6:.1f} Gsamples/s")
    b.close()
