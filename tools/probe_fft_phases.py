#!/usr/bin/env python3
"""Per-phase clock profile of k_fft4096_ms1 (needs a -DSS_FFT_PROF build of the library:
   make -C soundscope_amd/csrc OBJDIR=build_fprof OUT=$PWD/tools/bin/fftprof.so EXTRA=-DSS_FFT_PROF
   SOUNDSCOPE_HIP_LIB=tools/bin/fftprof.so python tools/probe_fft_phases.py [streams])"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names = ["window multiply, radix-16 #1, twiddles, LDS write", "barrier 1", "LDS read (transposed)", "barrier 2",
         "radix-16 #2, twiddles, LDS write", "barrier 3", "LDS read", "barrier 4", "radix-16 #3, publish", "barrier 5",
         "epilogue (mirror reads, |.|^2, log2, stores) + register slide", "barrier 6"]
f = L.lib().ss_debug_fft_prof
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
b = ssa.Batch(48000, 2, streams, 480000, 4096, 1024, flags=L.SS_BATCH_FFT)
b.synthesize(0x5EED0000, 0)
b.set_overlap(False)
b.run(); b.sync()
f(out, 1)
b.timing_enable(True)
n = 5
for _ in range(n):
    b.run(); b.sync()
f(out, 1)
ms, cnt = b.timing_read(L.SS_KERNEL_FFT)
tot = sum(out[i] for i in range(12))
print(f"k_fft4096_ms1 {ms / max(cnt, 1):.3f} ms (instrumented build), {out[15] // n} waves, {tot / out[15]:.0f} clocks per wave in the window loop")
for i, nm in enumerate(names):
    print(f"  {100.0 * out[i] / tot:5.1f} %  {out[i] / out[15]:10.0f} clk/wave  {nm}")
