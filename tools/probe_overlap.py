#!/usr/bin/env python3
"""Wall time of one batch pass with the spectrum kernel beside the time-domain chain (SS_BATCH_OVERLAP=1) or behind it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = ssa.Batch(48000, 2, streams, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
for _ in range(3):
    b.run(); b.sync()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    b.run()
b.sync()
dt = (time.perf_counter() - t0) / n
r = b.results()[1]
print(f"SS_BATCH_OVERLAP={os.environ.get('SS_BATCH_OVERLAP', '0')}: {dt * 1e3:.3f} ms per pass -> {streams * 960000 / dt / 1e9:.1f} Gsamples/s  (I={r.integrated_lufs:.3f})")
