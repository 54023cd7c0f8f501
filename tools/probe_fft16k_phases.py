#!/usr/bin/env python3
"""Per-phase clock profile of k_fft16k_run at BASELINE config 5 (needs a -DSS_FFT_PROF build of the library:
   make -C soundscope_amd/csrc OBJDIR=build_fprof OUT=$PWD/tools/bin/fftprof.so EXTRA=-DSS_FFT_PROF
   SOUNDSCOPE_HIP_LIB=tools/bin/fftprof.so python tools/probe_fft16k_phases.py [streams] [rate] [channels])
s_memtime counts at the constant 100 MHz reference clock: 10 ns per count; a CU holds two workgroups (16 waves), so the wall time of a
window per CU is HALF a workgroup's loop time per window."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rate = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
names = ["Hann rebuild + window multiply, radix-16 #1, twiddles, LDS write, next hop's load issued", "barrier 1", "LDS read (transposed)", "barrier 2",
         "radix-16 #2, twiddles from LDS, LDS write", "barrier 3", "LDS read", "barrier 4", "radix-16 #3, publish, epilogue twiddles requested",
         "barrier 5", "epilogue (mirror reads, Horner recombination per retained bin, |.|^2, log2, staged 16-byte stores) + register slide",
         "barrier 6"]
f = L.lib().ss_debug_fft_prof
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
b = ssa.Batch(rate, ch, streams, rate * 10, 16384, 1024, flags=L.SS_BATCH_FFT)
b.synthesize(0x5EED0000, 0)
b.run(); b.sync()
f(out, 1)
b.timing_enable(True)
n = 5
for _ in range(n):
    b.run(); b.sync()
f(out, 1)
ms, cnt = b.timing_read(L.SS_KERNEL_FFT)
lay, g = b.layout, b.geometry
tot = sum(out[i] for i in range(12))
waves = out[15] // n
windows = streams * lay.n_windows * lay.fft_channels
print(f"k_fft16k_run {ms / max(cnt, 1):.3f} ms (instrumented build), {waves} waves, {g.fft_blocks} workgroups x {g.fft_windows_per_block} windows, {lay.n_bins} bins")
per_win = tot / out[15] / g.fft_windows_per_block
print(f"  {per_win:.1f} counts of the clock register per wave and window in the loop")
for i, nm in enumerate(names):
    print(f"  {100.0 * out[i] / tot:5.1f} %  {out[i] / out[15] / g.fft_windows_per_block:8.2f} /window  {nm}")
