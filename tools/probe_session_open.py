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
# the same after a 1024-stream batch (10 GB) lived and died in the process
from soundscope_amd import _lib as L
b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_FFT); b.synthesize(1, 0); b.run(); b.sync()
t0 = time.perf_counter(); s = ssa.FileSession(x, 2, rate); t1 = time.perf_counter(); s.close()
print(f"with a 10 GB batch alive: open {1e3*(t1-t0):.2f} ms", flush=True)
b.close()
for i in range(4):
    t0 = time.perf_counter(); s = ssa.FileSession(x, 2, rate); t1 = time.perf_counter()
    t2 = time.perf_counter(); s.close(); t3 = time.perf_counter()
    print(f"after it was closed: open {1e3*(t1-t0):.2f} ms, close {1e3*(t3-t2):.2f} ms", flush=True)
