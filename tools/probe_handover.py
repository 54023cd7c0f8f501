#!/usr/bin/env python3
"""Segment hand-over of the batch time-domain kernel: sub-block energies of the three modes (SS_TD_AUTO: segments + fix-up launch,
SS_TD_RUN_IN, SS_TD_WHOLE_STREAMS) against the ONE-segment path (a 4096-stream batch walks each stream with one wave) and against
scipy's f64 lfilter, on the bench corpus and on DC-offset material; and the kernel times.  python tools/probe_handover.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from scipy.signal import lfilter
FL = L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM
rate, frames = 48000, 480000
def kweight():
    import ctypes as C
    b5 = (C.c_double * 5)(); a5 = (C.c_double * 5)()
    f = L.lib().ss_inspect_kweight
    f.argtypes = [C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    f(rate, b5, a5)
    return np.array(b5[:]), np.array(a5[:])
bb, aa = kweight()

def run(label, fill):
    ns = 1024
    b = ssa.Batch(rate, 2, ns, frames, 4096, 1024, flags=FL)
    fill(b, ns)
    res, t = {}, {}
    for mode in (0, 1, 2, 0, 1, 2):
        b.set_time_domain_mode(mode)
        b.run(); b.sync()
        b.timing_enable(True)
        for _ in range(10):
            b.run(); b.sync()
        ms, n = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
        b.timing_enable(False)
        t.setdefault(mode, []).append(ms / n)
        res[mode] = np.stack([b.subblocks(i) for i in range(ns)]).reshape(ns, 100, 2)
    g = b.geometry
    one = ssa.Batch(rate, 2, 4096, frames, 4096, 1024, flags=FL)
    fill(one, 4096)
    one.run(); one.sync()
    assert one.geometry.td_segments == 1 and one.geometry.td_split == 0
    ref = np.stack([one.subblocks(i) for i in range(ns)]).reshape(ns, 100, 2)
    x = b.download_input(0).astype(np.float64).reshape(-1, 2)
    sci = np.stack([(lfilter(bb, aa, x[:, c]) ** 2).reshape(100, 4800).sum(axis=1) for c in range(2)], axis=1)
    print(f"== {label}")
    print("   one-segment path vs scipy f64 (stream 0): max rel %.2e" % float(np.max(np.abs(ref[0] - sci) / sci)))
    for mode, name in ((0, "segments + fix-up (AUTO)"), (1, "segments, 0.1 s run-in"), (2, "whole-stream workgroups")):
        d = res[mode]
        rel = np.abs(d - ref) / np.maximum(np.abs(ref), 1e-300)
        neq = (d != ref).sum(axis=(0, 2))
        print(f"   {name:28s} time-domain launches {min(t[mode]):.4f} ms | vs one-segment: max rel {rel.max():.2e}, sub-blocks bit-equal in all streams: "
              f"{[int(i) for i in np.nonzero(neq == 0)[0]]} | vs scipy (stream 0) {float(np.max(np.abs(d[0] - sci) / sci)):.2e}")
    print("   worst sub-block index per mode:", {m: int(np.argmax((np.abs(res[m] - ref) / np.maximum(np.abs(ref), 1e-300)).max(axis=(0, 2)))) for m in res})

run("bench corpus (ss_batch_synthesize)", lambda b, n: b.synthesize(0x5EED0000, 0))
rng = np.random.default_rng(5)
x = np.empty(2 * frames, np.float32)
x[0::2] = (0.5 + 0.01 * rng.standard_normal(frames)).astype(np.float32)
x[1::2] = (-0.3 + 0.2 * np.sin(2 * np.pi * 100 * np.arange(frames) / rate)).astype(np.float32)
def fill_dc(b, n):
    for i in range(0, n, 128):
        b.upload(i, np.tile(x, 128))
run("DC-offset material (0.5 + noise | -0.3 + 100 Hz), every stream the same", fill_dc)
