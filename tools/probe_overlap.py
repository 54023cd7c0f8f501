#!/usr/bin/env python3
"""Wall time of one batch pass by launch mode (ss_batch_set_overlap): 0 sequential (spectrum kernel, then the time-domain chain),
1 the spectrum kernel on a second stream beside the whole chain, 2 beside the chain's tail only (the main time-domain launch first and
alone; the hand-over's second launch and the gating run beside the spectrum kernel).  Modes interleaved, three repetitions.
python tools/probe_overlap.py [streams] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
b = ssa.Batch(48000, 2, streams, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
ref = None
for rep in range(3):
    for mode in (0, 1, 2):
        b.set_overlap(mode)
        for _ in range(3):
            b.run(); b.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            b.run()
        b.sync()
        dt = (time.perf_counter() - t0) / n
        r = b.results()
        sig = (r[1].integrated_lufs, r[streams - 1].loudness_range, r[7].true_peak[0], float(b.fft(3)[17, 1, 100]))
        ref = ref or sig
        print(f"rep {rep} mode {mode}: {dt * 1e3:.3f} ms per pass -> {streams * 960000 / dt / 1e9:.1f} Gsamples/s  same results {sig == ref}", flush=True)
