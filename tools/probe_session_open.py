import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 12, rate)
for i in range(6):
    t0 = time.perf_counter(); s = ssa.FileSession(x, 2, rate); t1 = time.perf_counter()
    s.analyze_audio_file_samples(16384 * 2 + 2048)
    t2 = time.perf_counter(); s.close(); t3 = time.perf_counter()
    print(f"open {1e3*(t1-t0):.2f} ms, first tick {1e3*(t2-t1):.2f} ms, close {1e3*(t3-t2):.2f} ms", flush=True)
