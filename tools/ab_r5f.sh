#!/bin/bash
# round 5, call F: whole-stream workgroups (exact state hand-over) against time segments at the bench shape
out=gpurun_out/r5f; mkdir -p $out
python - > $out/split.log 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
ns = 1024
b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
res = {}
for mode in (1, 2, 1, 2):
    b.set_time_domain_mode(mode)
    g = b.geometry
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(10):
        b.run(); b.sync()
    ms, n = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    fms, fn = b.timing_read(L.SS_KERNEL_FFT)
    b.timing_enable(False)
    print("mode", mode, "td_split", g.td_split, "segments", g.td_segments, "k_time_domain %.4f ms" % (ms / n), "fft %.4f" % (fms / fn), flush=True)
    res[mode] = (b.results(), [b.subblocks(i).copy() for i in (0, 1, 511, 1023)], [b.waveform(i).copy() for i in (0, 1023)], [b.peaks(i) for i in (0, 1023)])
r1, r2 = res[1], res[2]
print("integrated max diff", max(abs(a.integrated_lufs - c.integrated_lufs) for a, c in zip(r1[0], r2[0])))
print("subblock rel diff", max(float(np.max(np.abs(a - c) / np.maximum(np.abs(c), 1e-300))) for a, c in zip(r1[1], r2[1])))
print("waveform equal", all(np.array_equal(a, c) for a, c in zip(r1[2], r2[2])))
print("peaks", r1[3], r2[3])
for i in (0, 1023):
    x = b.download_input(i)
    ref = po.analyze_stream(48000, x, 4096, 1024)
    r = r2[0][i]
    print("stream", i, "vs oracle: I", r.integrated_lufs - ref["integrated"], "LRA", r.loudness_range - ref["lra"], "TP rel", (r.true_peak[0] - ref["true_peak"][0]) / ref["true_peak"][0],
          "wave", np.array_equal(b.waveform(i).reshape(-1), ref["wave"][:, 1].astype(np.float32)))
    m = po.Meter(2, 48000); m.add_frames(x)
PY
cat $out/split.log
