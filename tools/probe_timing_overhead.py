#!/usr/bin/env python3
"""What the per-kernel event pairs cost a timed region: config 3's step (pass + corpus gate) with ss_batch_timing_enable on and off, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
lib = L.lib()
b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
for _ in range(5):
    b.run(); b.corpus_gate_enqueue(None)
lib.ss_device_synchronize()
for rep in range(4):
    for on in (False, True):
        b.timing_enable(on)
        t0 = time.perf_counter()
        for _ in range(20):
            b.run(); b.corpus_gate_enqueue(None)
        lib.ss_device_synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        b.sync()
        print(f"rep {rep} events {'on ' if on else 'off'}: {dt:.4f} ms per step", flush=True)
