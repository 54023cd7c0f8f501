#!/usr/bin/env python3
"""The session tick before and after a large batch has been created and destroyed in the same process.  Through most of
round 4 "the first session after a big batch" ticked 133 instead of 88 us: not the release of memory — the session's two
streams had been dealt hardware queues on one pipe of the command processor (tools/probe_tick_queues.sh), which the stream
churn of a batch made likely.  Since the tick is one launch on one stream (k_tick) the three numbers agree."""
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
