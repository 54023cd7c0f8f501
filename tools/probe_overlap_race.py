#!/usr/bin/env python3
"""Does an overlapped pass return exactly what the sequential pass returns?  Every stream, many repetitions
(device-side checksums), and where it does not: which windows / rows / bins differ, and by how much.
usage: probe_overlap_race.py [streams] [reps] [--split]   (--split: spectrum and time-domain chain as two batches)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L

args = [a for a in sys.argv[1:] if not a.startswith("--")]
ns = int(args[0]) if len(args) > 0 else 1024
reps = int(args[1]) if len(args) > 1 else 20
split = "--split" in sys.argv
print("lib", L.LIB_PATH, "streams", ns, "reps", reps, "split", split, flush=True)

if split:
    # two batches = two HIP streams with no fork/join between them: the spectrum kernel of A beside the time-domain chain of B
    A = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_FFT)
    B = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM)
    A.synthesize(0x5EED0000, 0); B.synthesize(0x5EED0000, 0)
    A.run(); A.sync(); B.run(); B.sync()
    ra, rb = A.checksums(), B.checksums()
    for rep in range(reps):
        B.run(); A.run(); B.run(); A.sync(); B.sync()
        ca, cb = A.checksums(), B.checksums()
        print("rep", rep, "A (spectrum) streams differing", int((ca != ra).any(axis=1).sum()), "B (time domain) streams differing", int((cb != rb).any(axis=1).sum()), flush=True)
    sys.exit(0)

b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
b.set_overlap(0)
b.run(); b.sync()
ref = b.checksums()
refr = [(r.integrated_lufs, r.true_peak[0], r.true_peak[1]) for r in b.results()]
total_bad = 0
for rep in range(reps):
    mode = (0, 1, 2)[rep % 3]
    b.set_overlap(mode)
    b.run(); b.sync()
    c = b.checksums()
    bad = np.nonzero((c != ref).any(axis=1))[0]
    rr = [(r.integrated_lufs, r.true_peak[0], r.true_peak[1]) for r in b.results()]
    print("rep", rep, "mode", mode, "streams differing", len(bad), "cols", sorted(set(np.nonzero(c != ref)[1].tolist())), "results equal", rr == refr, flush=True)
    if len(bad) and total_bad < 12:
        got = {int(i): b.fft(int(i)).copy() for i in bad[:4]}
        b.set_overlap(0); b.run(); b.sync()
        for i, g in got.items():
            f = b.fft(i)
            d = np.argwhere(g != f)
            wins = sorted(set((int(w), int(r)) for w, r, _ in d))
            w0, r0 = wins[0]
            bins = np.nonzero(g[w0, r0] != f[w0, r0])[0]
            print("   stream", i, "wrong (window,row)", wins[:6], "n", len(wins), "| first: bins", bins[:4], "..", bins[-1], "count", len(bins),
                  "max |d| dB", float(np.abs(g[w0, r0] - f[w0, r0]).max()), flush=True)
            total_bad += 1
print("done")
