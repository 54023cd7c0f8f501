#!/usr/bin/env python3
"""The session tick before and after a large batch has been created and destroyed in the same process: the FIRST session
opened after gigabytes of device memory were released ticks ~45 us slower (133 against 88 us on MI355X / ROCm 7.2); the next one
is back to normal, with or without a hipDeviceSynchronize in between.  bench.py therefore times the tick before it closes its batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, soundscope_amd as ssa
from soundscope_amd import _lib as L
def tick(tag):
    r = bench.reference_tick_workload(ssa, L)
    print(f"{tag}: tick median {r['gpu_tick_us']['median']:.1f} us", flush=True)
tick("fresh process")
b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_FFT); b.synthesize(1, 0); b.run(); b.sync(); b.close()
tick("after a 1024-stream batch was closed")
tick("a second session after that")
