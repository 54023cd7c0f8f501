#!/usr/bin/env python3
"""PCIe-inclusive throughput of the pipelined corpus runner (host-resident corpus -> per-stream loudness results)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
rate, frames = 48000, 480000
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
base = (0.2 * rng.standard_normal(2 * frames)).astype(np.float32)
for dtype in (np.float32, np.int16):
    one = base if dtype == np.float32 else np.round(base * 20000).astype(np.int16)
    corpus = np.tile(one, ns)
    for chunk in (128, 256):
        t0 = time.perf_counter()
        res, hist = ssa.analyze_corpus(corpus, rate, 2, frames, chunk_streams=chunk)
        dt = time.perf_counter() - t0
        print(f"{np.dtype(dtype).name:8s} chunk {chunk:4d}: {dt * 1e3:8.1f} ms for {ns} streams ({corpus.nbytes / 1e9:.2f} GB) -> "
              f"{corpus.size / dt / 1e9:.2f} Gsamples/s, {corpus.nbytes / dt / 1e9:.1f} GB/s over PCIe, I[0]={res[0][0]:.2f}")
