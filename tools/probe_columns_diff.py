#!/usr/bin/env python3
"""Where the fused columns differ from the two-pass result (debug): python tools/probe_columns_diff.py [cols]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from conftest import make_stereo
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rate, frames, ns = 48000, 48000 * 3 + 333, 5
xs = [make_stereo(300 + s, frames, rate=rate, level=0.05 + 0.2 * s, gap=(s == 2)) for s in range(ns)]
xs[4][1::2] = xs[4][0::2]
xs[3][2 * 70000] = np.nan
xs[1][2 * 30000 + 1] = np.inf
two = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL)
two.upload(0, np.concatenate(xs)); two.run(); two.render_spectrum(cols, None)
one = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=L.SS_BATCH_ALL, spectrum_columns=cols)
one.upload(0, np.concatenate(xs)); one.set_columns_gain(None); one.run(); one.sync()
print("integrated:", [r.integrated_lufs for r in one.results()])
for s in range(ns):
    a, b = one.spectrum_columns(s), two.spectrum_columns(s)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    print("stream", s, "mismatches", int(bad.sum()))
    f = two.fft(s)
    for w, r, c in list(zip(*np.nonzero(bad)))[:12]:
        row = f[w, r]
        print("   window", w, "row", r, "col", c, "fused", a[w, r, c], "two-pass", b[w, r, c], "| row: nan", int(np.isnan(row).sum()), "+inf", int(np.isposinf(row).sum()), "-inf", int(np.isneginf(row).sum()), "finite max", np.nanmax(np.where(np.isfinite(row), row, -1e30)))
