#!/usr/bin/env python3
"""The reference's native window on stereo material: 256 streams x 10 s x 48 kHz stereo, N = 16384, hop 1024 (mid/side)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
b = ssa.Batch(48000, 2, 256, 480000, 16384, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(3, 0)
b.run(); b.sync()
b.timing_enable(True)
for _ in range(5):
    b.run(); b.sync()
ms = [b.timing_read(k)[0] / 5 for k in range(4)]
print(f"stereo N=16384 x 256 streams: {L.lib().ss_batch_kernel_name(b._h, 0).decode()} {ms[0]:.3f} ms, time domain {ms[1]:.3f} ms -> {256 * 960000 / sum(ms) / 1e6:.1f} Gsamples/s")
