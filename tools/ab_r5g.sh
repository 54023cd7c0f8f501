#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
python - > $out/modes2.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
ns = 1024
FL = L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM
b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=FL)
b.synthesize(0x5EED0000, 0)
res = {}
for mode in (0, 1, 2):
    b.set_time_domain_mode(mode)
    b.run(); b.sync()
    res[mode] = np.stack([b.subblocks(i) for i in range(ns)])
one = ssa.Batch(48000, 2, 4096, 480000, 4096, 1024, flags=FL)
one.synthesize(0x5EED0000, 0)
one.run(); one.sync()
g = one.geometry
print("reference batch: streams 4096 segments", g.td_segments, "split", g.td_split)
ref = np.stack([one.subblocks(i) for i in range(ns)])
x0 = b.download_input(0); x1 = one.download_input(0)
print("same input:", np.array_equal(x0, x1))
for mode in (0, 1, 2):
    d = res[mode].reshape(ns, 100, 2)
    r = ref.reshape(ns, 100, 2)
    neq = (d != r)
    print("mode", mode, "mismatching sub-blocks by index (sum over streams, both channels):")
    print("   ", neq.sum(axis=(0, 2)).tolist())
    rel = np.abs(d - r) / np.maximum(np.abs(r), 1e-300)
    print("    max rel by sub-block index:", ["%.1e" % v for v in rel.max(axis=(0, 2))][:30])
# independent f64 reference for stream 0: scipy lfilter with the library's coefficients
from scipy.signal import lfilter
from oracle import pyoracle as po
m = po.Meter(2, 48000)
bb, aa = m.filter_coeffs()
x = b.download_input(0).astype(np.float64).reshape(-1, 2)
for c in range(2):
    y = lfilter(bb, aa, x[:, c])
    e = (y * y).reshape(100, 4800).sum(axis=1)
    for mode in (0, 1, 2):
        d = res[mode].reshape(ns, 100, 2)[0, :, c]
        print("stream 0 ch", c, "mode", mode, "max rel vs scipy f64:", float(np.max(np.abs(d - e) / e)))
    print("stream 0 ch", c, "one-segment max rel vs scipy f64:", float(np.max(np.abs(ref.reshape(ns, 100, 2)[0, :, c] - e) / e)))
PY
cat $out/modes2.log
