#!/usr/bin/env python3
"""Relative difference of the carried K-weighting state between the handle's streaming path and the oracle, call by call, on an
impulse followed by silence (the decay of test_subnormal_filter_state_is_flushed_like_the_crate): python tools/probe_state_drift.py [rate]
(with a -DSS_TUNING build, SS_TD_SPLIT=0 keeps streaming calls on one wave)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from oracle import pyoracle as po
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
slice_len = 16384
an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
mm = po.Meter(2, rate)
imp = np.zeros(2 * rate * 3, np.float32); imp[0] = 1.0; imp[1] = -0.5
out = []
for k, off in enumerate(range(0, imp.size, slice_len)):
    an.add_samples(imp[off:off + slice_len]); mm.add_frames(imp[off:off + slice_len])
    g, o = an.filter_state(0), mm.filter_state(0)
    nz = np.abs(o) > 1e-290
    if nz.any():
        out.append(float(np.max(np.abs(g[nz] / o[nz] - 1.0))))
print(f"rate {rate}: max relative state difference per call:", " ".join(f"{v:.1e}" for v in out[:24]))
