#!/usr/bin/env python3
"""Tick medians of file sessions opened one after another in one process (does a session's speed depend on what was created
and destroyed before it — which HSA queues its two streams landed on?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 8, rate)
def med(sess, n=150):
    t = []
    for k, pos in enumerate(range(16384 * 2 + 2048, x.size, 2048)):
        t0 = time.perf_counter(); sess.analyze_audio_file_samples(pos); t1 = time.perf_counter()
        if k >= 10: t.append(t1 - t0)
        if k >= n: break
    return np.median(t) * 1e6
keep = []
for i in range(6):
    s = ssa.FileSession(x, 2, rate); keep.append(s)
    print(f"session {i} (all earlier ones alive): {med(s):.1f} us", flush=True)
for s in keep: s.close()
for i in range(4):
    s = ssa.FileSession(x, 2, rate)
    print(f"session {i} (opened after the earlier one was closed): {med(s):.1f} us", flush=True)
    s.close()
an = ssa.Analyzer(); an.create_loudness_meter(2, rate); an.calculate_integrated_lufs(2, x)
s = ssa.FileSession(x, 2, rate); print(f"after calculate_integrated_lufs on another handle: {med(s):.1f} us", flush=True); s.close()
b = ssa.Batch(48000, 2, 64, 480000, 4096, 1024, flags=L.SS_BATCH_FFT); b.synthesize(1, 0); b.run(); b.sync()
s = ssa.FileSession(x, 2, rate); print(f"a 64-stream batch alive: {med(s):.1f} us", flush=True); s.close()
b.close()
for i in range(3):
    s = ssa.FileSession(x, 2, rate); print(f"after the batch was closed, session {i}: {med(s):.1f} us", flush=True); s.close()
