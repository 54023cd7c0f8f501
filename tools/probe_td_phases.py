#!/usr/bin/env python3
"""Per-phase clock profile of k_time_domain (needs a -DSS_TD_PROF build of the library:
   make -C soundscope_amd/csrc OBJDIR=build_prof OUT=$PWD/tools/bin/tdprof.so EXTRA=-DSS_TD_PROF
   SOUNDSCOPE_HIP_LIB=tools/bin/tdprof.so python tools/probe_td_phases.py [streams] [channels] [rate])"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rate = int(sys.argv[3]) if len(sys.argv) > 3 else 48000
names = ["stage tile (prefetch regs -> LDS, next prefetch issue)", "min-max decimation", "K-weight pass 1", "scan",
         "K-weight pass 2 (+ energy, sample peak)", "true peak: halo save, f32 remainder, f16 conversion", "true peak: MFMA product",
         "tile tail (carry-out, sub-block energy, halo copy)"]
f = L.lib().ss_debug_td_prof
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
for flags, label in ((L.SS_BATCH_ALL, "all"), (L.SS_BATCH_LUFS, "loudness only")):
    b = ssa.Batch(rate, ch, streams, rate * 10, 4096, 1024, flags=flags)
    b.synthesize(0x5EED0000, 0)
    b.set_overlap(False)
    b.run(); b.sync()
    f(out, 1)
    b.timing_enable(True)
    n = 5
    for _ in range(n):
        b.run(); b.sync()
    f(out, 1)
    ms, cnt = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    tot = sum(out[i] for i in range(8))
    print(f"--- {label}: k_time_domain {ms / max(cnt, 1):.3f} ms (instrumented build), {out[15] // n} waves, {tot / out[15]:.0f} clocks per wave in the tile loop")
    for i, nm in enumerate(names):
        print(f"  {100.0 * out[i] / tot:5.1f} %  {out[i] / out[15]:10.0f} clk/wave  {nm}")
    b.close()
