#!/usr/bin/env python3
"""Does the overlapped pass return exactly what the sequential pass returns?  (bench shape, several repetitions)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
b.set_overlap(False)
b.run(); b.sync()
picks = list(range(0, ns, max(1, ns // 64)))
ref = {i: b.fft(i).copy() for i in picks}
refr = [(r.integrated_lufs, r.true_peak[0], r.true_peak[1]) for r in b.results()]
for rep in range(6):
    b.set_overlap(rep % 2 == 0)
    b.run(); b.sync()
    bad = []
    for i in picks:
        f = b.fft(i)
        d = np.argwhere((f != ref[i]).any(axis=2))
        if d.size:
            bad.append((i, [tuple(x) for x in d[:4]], int(d.shape[0])))
    rr = [(r.integrated_lufs, r.true_peak[0], r.true_peak[1]) for r in b.results()]
    print("rep", rep, "overlap", rep % 2 == 0, "fft mismatching streams", len(bad), bad[:6], "results equal", rr == refr, flush=True)
