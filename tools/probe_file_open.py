#!/usr/bin/env python3
"""receive_audio_file on the device (ss_session_open_file): open times of files of different lengths one after another (the
one-shot loudness batch is kept and re-used through the ragged path), and the gain against the oracle's meter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
from oracle import pyoracle as po
for secs in (12, 7, 12, 30, 3):
    x = make_stereo(secs, 48000 * secs, 48000)
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); s = ssa.FileSession(x, 2, 48000); ts.append((time.perf_counter() - t0) * 1e3)
        g = s.fft_gain_compensation_db; s.close()
    m = po.Meter(2, 48000); m.add_frames(x)
    print(f"{secs:3d} s: open ms {[round(t, 2) for t in ts]}  gain {g:.6f}  oracle {np.float32(-13.0) - np.float32(m.integrated()):.6f}")
an = ssa.Analyzer(); an.create_loudness_meter(2, 48000)
x = make_stereo(5, 48000 * 9, 48000)
m = po.Meter(2, 48000); m.add_frames(x)
print("calculate_integrated_lufs", an.calculate_integrated_lufs(2, x), m.integrated())
