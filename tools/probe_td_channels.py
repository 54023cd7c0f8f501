import os, sys, zlib
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
SHAPES = [("48 kHz 3 ch x 683 x 10 s", 48000, 3, 683, 480000), ("48 kHz 4 ch x 512 x 10 s", 48000, 4, 512, 480000), ("48 kHz 5 ch x 410 x 10 s", 48000, 5, 410, 480000),
          ("48 kHz 7 ch x 293 x 10 s", 48000, 7, 293, 480000), ("48 kHz mono x 2048 x 10 s", 48000, 1, 2048, 480000), ("48 kHz stereo x 1024 x 10 s", 48000, 2, 1024, 480000),
          ("96 kHz stereo x 1024 x 10 s", 96000, 2, 1024, 960000), ("96 kHz 3 ch x 341 x 10 s", 96000, 3, 341, 960000), ("48 kHz 16 ch x 128 x 10 s", 48000, 16, 128, 480000)]
for label, rate, ch, ns, frames in SHAPES:
    b = ssa.Batch(rate, ch, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL & ~L.SS_BATCH_FFT)
    b.synthesize(7, 0)
    for _ in range(2): b.run(); b.sync()
    b.timing_enable(True)
    ms0, n0 = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    for _ in range(6): b.run(); b.sync()
    ms1, n1 = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    b.timing_enable(False)
    r = b.results()[0]
    print(f"{label:<36} {ns*frames*ch/1e9:.2f} G samples  k_time_domain {(ms1-ms0)/6:7.4f} ms  -> {ns*frames*ch/((ms1-ms0)/6*1e-3)/1e9:7.1f} G samples/s   TP {max(r.true_peak[:ch]):.7f}", flush=True)
    b.close()
