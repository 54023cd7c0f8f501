#!/usr/bin/env python3
"""Which register build of k_time_domain a big grid should take: the four-waves-per-SIMD build (128 VGPRs; some instantiations spill
16-104 B per lane, profiles/r05_td_kernel_resources.txt) against the three-waves build (168 VGPRs, nothing spilled, the grid in more
rounds).  Needs a -DSS_TUNING build (SS_TD_WPS forces the build, SS_TD_VERBOSE names the instantiation):
    SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so python tools/probe_td_wps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

SHAPES = [  # (label, rate, channels, streams, frames, flags, true-peak factor (0 = the rate's rule), time-domain mode)
    ("48k stereo, bench shape (int4 decimation)", 48000, 2, 1024, 480000, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("48k stereo, odd length (general decimation)", 48000, 2, 1024, 470001, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("48k stereo, bench shape, whole streams", 48000, 2, 1024, 480000, L.SS_BATCH_ALL, 0, L.SS_TD_WHOLE_STREAMS),
    ("48k stereo, odd length, whole streams", 48000, 2, 1024, 470001, L.SS_BATCH_ALL, 0, L.SS_TD_WHOLE_STREAMS),
    ("96k stereo, odd length", 96000, 2, 1024, 470001, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("96k 8 ch x 256 streams (4 x config 5), 2x true peak", 96000, 8, 256, 960000, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("96k 8 ch x 256 streams, forced 4x true peak", 96000, 8, 256, 960000, L.SS_BATCH_ALL, 4, L.SS_TD_AUTO),
    ("48k 5.1 x 512 streams", 48000, 6, 512, 480000, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("48k 4 ch x 512 streams (general channel count)", 48000, 4, 512, 480000, L.SS_BATCH_ALL, 0, L.SS_TD_AUTO),
    ("48k 4 ch x 512 streams, no decimation", 48000, 4, 512, 480000, L.SS_BATCH_ALL & ~L.SS_BATCH_WAVEFORM, 0, L.SS_TD_AUTO),
    ("48k stereo, loudness only", 48000, 2, 1024, 480000, L.SS_BATCH_LUFS, 0, L.SS_TD_AUTO),
]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for label, rate, ch, ns, frames, flags, tpf, mode in SHAPES:
    fft_n = 4096
    b = ssa.Batch(rate, ch, ns, frames, fft_n, 1024, flags=flags & ~L.SS_BATCH_FFT, true_peak_factor=tpf)
    b.set_time_domain_mode(mode)
    b.synthesize(7, 0)
    line = []
    for wps in ("3", "4"):
        os.environ["SS_TD_WPS"] = wps
        os.environ["SS_TD_VERBOSE"] = "1"
        sys.stderr.write(f"## {label}: SS_TD_WPS={wps}\n"); sys.stderr.flush()
        b.run(); b.sync()
        del os.environ["SS_TD_VERBOSE"]
        b.run(); b.sync()
        b.timing_enable(True)
        ms0, n0 = b.timing_read(1)
        for _ in range(steps):
            b.run(); b.sync()
        ms, n = b.timing_read(1)
        line.append((ms - ms0) / max(n - n0, 1))
    r = b.results()[0]
    print(f"{label:58s} k_time_domain  3 waves {line[0]:8.4f} ms   4 waves {line[1]:8.4f} ms   ({line[1] / line[0]:.3f})   I {r.integrated_lufs:.4f}")
    sys.stdout.flush()
    b.close()
