#!/usr/bin/env python3
"""Where a session tick's wall time goes on the host side (needs a -DSS_TUNING build: ss_debug_tick_prof).
[0] host-side checks, [1] tick launch (loudness call + spectrum), [2] separate spectrum launch (if any), [3] short-term enqueue + event + gating enqueue,
[4] x halves of the two charts (the device working), [5] the rows' flags + y halves (the loudness call still running),
[6] wait for the event"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 12, rate)
sess = ssa.FileSession(x, 2, rate)
f = C.CDLL(L.LIB_PATH).ss_debug_tick_prof
f.argtypes = [C.POINTER(C.c_double), C.c_int]
out = (C.c_double * 8)()
pos = list(range(16384 * 2 + 2048, x.size, 2048))
for p in pos[:20]:
    sess.analyze_audio_file_samples(p)
f(out, 1)
pause = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.0          # optional: milliseconds between ticks
if pause: pos = pos[:20 + 100]
wall = 0.0
for p in pos[20:]:
    t0 = time.perf_counter()
    sess.analyze_audio_file_samples(p)
    wall += time.perf_counter() - t0
    if pause: time.sleep(pause)
wall = wall / len(pos[20:]) * 1e6
f(out, 1)
n = len(pos[20:])
names = ["host-side checks", "tick launch", "separate spectrum launch", "short-term enqueue + event + gating", "x halves (host)",
         "row flags + y halves", "wait: event"]
print(f"tick wall (python loop) {wall:.1f} us")
for i, nm in enumerate(names):
    print(f"  {out[i] / n:7.1f} us  {nm}")
