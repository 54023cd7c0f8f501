#!/usr/bin/env python3
"""One long stream (config 2: 1 x 600 s, 48 kHz stereo) and a few other small batches: per-kernel times of a pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
for streams, seconds in ((1, 600), (1, 3600), (8, 600), (1, 10), (64, 60)):
    b = ssa.Batch(48000, 2, streams, 48000 * seconds, 4096, 1024, flags=L.SS_BATCH_ALL)
    b.synthesize(0x5EED0000, 0)
    for _ in range(2):
        b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(10):
        b.run(); b.sync()
    ms = [b.timing_read(k)[0] / max(b.timing_read(k)[1], 1) for k in range(L.SS_KERNEL_COUNT)]
    r = b.results()[0]
    print(f"{streams:3d} x {seconds:5d} s: spectrum {ms[0]:.4f} ms, time domain {ms[1]:.4f}, finalize {ms[2]:.4f}, waveform {ms[3]:.4f} "
          f"-> {streams * seconds * 96000 / sum(ms) / 1e6:.1f} Gsamples/s;  I = {r.integrated_lufs:.9f} LRA = {r.loudness_range:.9f}")
    b.close()
