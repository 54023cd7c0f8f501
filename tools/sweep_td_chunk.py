#!/usr/bin/env python3
"""Chunk-length sweep of k_time_domain (needs a -DSS_TUNING build: SS_TD_L forces the length).
usage: SOUNDSCOPE_HIP_LIB=tools/bin/tune.so python tools/sweep_td_chunk.py <rate> <channels> <streams> L1 L2 ..."""
import os, subprocess, sys
rate, ch, streams = sys.argv[1:4]
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import soundscope_amd as ssa
from soundscope_amd import _lib as L
rate, ch, streams = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = ssa.Batch(rate, ch, streams, rate * 10, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM, true_peak_factor=4)
b.synthesize(7, 0)
b.run(); b.sync()
b.timing_enable(True)
for _ in range(5):
    b.run(); b.sync()
ms, n = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
g = b.geometry
r = b.results()[0]
print(f"L={os.environ.get('SS_TD_L','auto'):>4}  k_time_domain {ms / n:.3f} ms  segments {g.td_segments} x {g.td_segment_subblocks} sub-blocks  I={r.integrated_lufs:.6f} TP={r.true_peak[0]:.7f}")
'''
for l in sys.argv[4:]:
    env = dict(os.environ)
    if l != "auto":
        env["SS_TD_L"] = l
    subprocess.run([sys.executable, "-c", code, rate, ch, streams], env=env)
