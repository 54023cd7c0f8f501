#!/usr/bin/env python3
"""BASELINE config 5 probe: 96 kHz, 8 ch, N=16384 per channel, true peak 4x (forced) and 2x (ebur128 rule)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for tp in (4, 0):
    b = ssa.Batch(96000, 8, streams, 960000, 16384, 1024, true_peak_factor=tp)
    b.synthesize(7, 0)
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(3):
        b.run(); b.sync()
    tot = 0
    for k in range(L.SS_KERNEL_COUNT):
        ms, n = b.timing_read(k); tot += ms / max(n, 1)
        print(f"tp={tp or 2}x {L.lib().ss_batch_kernel_name(b._h, k).decode():16s} {ms / max(n, 1):9.3f} ms")
    lay = b.layout
    print(f"tp={tp or 2}x total {tot:.3f} ms -> {streams * 960000 * 8 / tot / 1e6:.2f} Gsamples/s; windows {lay.n_windows} bins {lay.n_bins}")
    b.close()
