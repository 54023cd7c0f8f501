#!/usr/bin/env python3
"""The file tick at different cadences: back to back, 0.3 ms, 2 ms, 8 ms (the TUI's loop) and 21.3 ms (the reference's tick period
at 48 kHz) between calls — idle clocks and cold caches are part of what a real tick costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 12, rate)
sess = ssa.FileSession(x, 2, rate)
pos = list(range(16384 * 2 + 2048, x.size, 2048))
for p in pos[:30]: sess.analyze_audio_file_samples(p)
for pause in (0.0, 0.0003, 0.002, 0.008, 0.0213):
    t = []
    n = 200 if pause < 0.005 else 80
    for p in pos[30:30 + n]:
        t0 = time.perf_counter(); sess.analyze_audio_file_samples(p); t1 = time.perf_counter()
        t.append(t1 - t0)
        if pause: time.sleep(pause)
    print(f"pause {pause * 1e3:5.1f} ms: tick median {np.median(t) * 1e6:.1f} us, p90 {np.percentile(t, 90) * 1e6:.1f} us")
