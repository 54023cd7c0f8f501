#!/usr/bin/env python3
"""Packs the .npy files written by the Rust harness into tests/golden/crates_v1.npz (python tools/pin_from_crates/pack.py <dir>)."""
import glob
import os
import sys

import numpy as np

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
arrays = {os.path.splitext(os.path.basename(p))[0]: np.load(p) for p in sorted(glob.glob(os.path.join(src, "*.npy")))}
if not arrays:
    raise SystemExit(f"no .npy files in {src}")
out = os.path.join(root, "tests", "golden", "crates_v1.npz")
np.savez_compressed(out, **arrays)
print(out, os.path.getsize(out), "bytes,", len(arrays), "arrays")
