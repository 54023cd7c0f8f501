#!/usr/bin/env python3
"""Timeline of the loudness call inside one tick (needs a -DSS_TD_TRACE build: ss_debug_td_trace): for every tile of the call the
time (us, from the first wave's start) at which its wave took it up, had it staged, had run the zero-state pass, received the
state of the tile in front, finished the scan, the second pass, received the energy shares, finished the true-peak product."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
x = make_stereo(1, rate * 6, rate)
sess = ssa.FileSession(x, 2, rate)
f = C.CDLL(L.LIB_PATH).ss_debug_td_trace
f.argtypes = [C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 512)()
pos = list(range(16384 * 2 + 2048, x.size, 2048))
for n, p in enumerate(pos[:60]):
    sess.analyze_audio_file_samples(p)
    if n in (40, 41, 42):
        f(out)
        t = np.array(out, dtype=np.float64).reshape(32, 16)
        t0 = min(v for v in t[:8, 11] if v > 0)
        print(f"--- tick {n}: waves start {[round((v - t0) / 100, 2) for v in t[:8, 11]]}  end {[round((v - t0) / 100, 2) for v in t[:8, 12]]}")
        print(f"    spectrum workgroups: start {[round((t[30 + q, 0] - t0) / 100, 2) for q in (0, 1)]} end {[round((t[30 + q, 1] - t0) / 100, 2) for q in (0, 1)]};"
              f" first ring workgroup: {round((t[29, 0] - t0) / 100, 2)} .. {round((t[29, 1] - t0) / 100, 2)}")
        print("tile  taken staged decim  pass1 state>  scan  pass2  e>    conv   tp    tail")
        for ti in range(12):
            if t[ti, 8] < t0: continue
            r = lambda s: (t[ti, s] - t0) / 100 if t[ti, s] >= t0 else float('nan')
            print(f"{ti:4d} {r(8):6.2f} {r(0):6.2f} {r(1):6.2f} {r(2):6.2f} {r(9):6.2f} {r(3):6.2f} {r(4):6.2f} {r(10):6.2f} {r(5):6.2f} {r(6):6.2f} {r(7):6.2f}")
