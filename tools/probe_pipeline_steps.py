#!/usr/bin/env python3
"""K passes enqueued back to back with ONE synchronisation at the end (overlap mode): ms per pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
b.set_overlap(True)
for _ in range(3):
    b.run()
L.lib().ss_device_synchronize()
for mode in ("sync each", "sync at end"):
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        b.run()
        if mode == "sync each":
            b.histograms()
    L.lib().ss_device_synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:12s} {dt / K * 1e3:.3f} ms per pass")
