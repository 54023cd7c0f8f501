#!/usr/bin/env python3
"""k_time_domain's time at the shapes that take its different forms (whole samples per decimation bin / any samples per bin, the
channel counts, the true-peak factors), for A/B runs of two library builds:   [SOUNDSCOPE_HIP_LIB=...] python tools/probe_td_shapes.py [steps]
Prints per shape: the time-domain kernels' time per pass (main + hand-over launch), integrated loudness, true peak and a checksum of
the decimation bins of stream 0 (two builds must agree on them)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L

SHAPES = [  # label, rate, channels, streams, frames
    ("48 kHz stereo x 1024 x 10 s (96 samples per bin)", 48000, 2, 1024, 480000),
    ("44.1 kHz stereo x 1024 x 10 s (88.2 per bin)", 44100, 2, 1024, 441000),
    ("48 kHz stereo x 1024, odd length (any per bin)", 48000, 2, 1024, 470001),
    ("96 kHz stereo x 512 x 10 s", 96000, 2, 512, 960000),
    ("88.2 kHz stereo x 512 x 10 s", 88200, 2, 512, 882000),
    ("48 kHz 5.1 x 512 x 10 s", 48000, 6, 512, 480000),
    ("44.1 kHz 5.1 x 512 x 10 s", 44100, 6, 512, 441000),
    ("96 kHz 8 ch x 64 x 10 s, 4x (config 5)", 96000, 8, 64, 960000),
    ("96 kHz 8 ch x 64 x 10 s, the crate's 2x", 96000, 8, 64, 960000),
    ("96 kHz stereo x 1024 x 5 s", 96000, 2, 1024, 480000),
    ("96 kHz 5.1 x 128 x 10 s", 96000, 6, 128, 960000),
    ("48 kHz mono x 2048 x 10 s", 48000, 1, 2048, 480000),
    ("44.1 kHz mono x 2048 x 10 s", 44100, 1, 2048, 441000),
]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for label, rate, ch, ns, frames in SHAPES:
    tpf = 4 if "4x" in label else 0
    b = ssa.Batch(rate, ch, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL & ~L.SS_BATCH_FFT, true_peak_factor=tpf)
    b.synthesize(7, 0)
    for _ in range(2):
        b.run(); b.sync()
    b.timing_enable(True)
    ms0, n0 = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    for _ in range(steps):
        b.run(); b.sync()
    ms1, n1 = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    b.timing_enable(False)
    td = (ms1 - ms0) / steps
    r = b.results()[0]
    w = b.waveform(0)
    print(f"{label:<52} k_time_domain {td:7.4f} ms   I {r.integrated_lufs:.4f}  TP {max(r.true_peak[:ch]):.7f}  wave crc {zlib.crc32(np.ascontiguousarray(w).tobytes()):08x}", flush=True)
    b.close()
