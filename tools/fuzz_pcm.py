#!/usr/bin/env python3
"""Randomised PCM ingest (symphonia's sample conversions as audio_player.rs:169-267 sees them): random BYTES in every sample
format — for the float formats every bit pattern: NaNs with payloads, sub-normals, f64 values beyond the f32 range — and random lengths,
device conversion against the oracle's, bit for bit.      python tools/fuzz_pcm.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soundscope_amd import _lib as L
from soundscope_amd import ingest
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SB = {L.SS_PCM_U8: 1, L.SS_PCM_S16: 2, L.SS_PCM_S24: 3, L.SS_PCM_S32: 4, L.SS_PCM_F32: 4, L.SS_PCM_F64: 8}
failed = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    fmt = int(rng.choice(list(SB)))
    count = int(rng.choice([0, 1, 2, 3, 63, 64, 65, 255, 4097, int(rng.integers(0, 300000))]))
    raw = rng.integers(0, 256, count * SB[fmt], dtype=np.uint8)
    if fmt in (L.SS_PCM_F32, L.SS_PCM_F64) and rng.random() < 0.5:       # ordinary values mixed with the odd ones
        v = rng.standard_normal(count) * 10.0 ** rng.uniform(-45, 40, count)
        raw = v.astype("<f4" if fmt == L.SS_PCM_F32 else "<f8").view(np.uint8)
    raw = raw.tobytes()
    got, ref = ingest.pcm_decode(raw, fmt), po.pcm_to_f32(raw, fmt)
    same = got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    if not same and got.shape == ref.shape:                                # NaN payloads may legitimately differ in the quiet bit only
        d = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
        same = bool(np.all(np.isnan(got[d]) & np.isnan(ref[d])))
        if same: print(f"note seed {seed}: format {fmt}, {d.size} NaN payloads differ (first: {got.view(np.uint32)[d[0]]:#x} vs {ref.view(np.uint32)[d[0]]:#x})")
    if not same:
        failed += 1
        d = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0] if got.shape == ref.shape else []
        print(f"FAIL seed {seed}: format {fmt}, {count} samples, {len(d)} differ" + (f", first at {d[0]}: {got[d[0]]!r} vs {ref[d[0]]!r}" if len(d) else f" shapes {got.shape} {ref.shape}"), flush=True)
print(f"{n} pcm programmes, {failed} failed")
sys.exit(1 if failed else 0)
