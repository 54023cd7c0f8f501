#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz: expected outputs of the CPU oracle on seeded inputs.

The reference's arithmetic lives in un-vendored Rust crates that cannot be built or imported in
the build image (no cargo/rustc), so these vectors are produced by the oracle, which
is itself pinned by the known-answer tests in tests/test_oracle_known_answers.py.  Inputs are
regenerated from `golden_input` (pure integer arithmetic, version independent); only expected
outputs are stored.  Re-run after an intentional oracle change:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def golden_input(seed: int, n: int, scale: float = 0.5) -> np.ndarray:
    """Deterministic pseudo-audio: xorshift32 noise + two integer-phase 'tones' (no libm)."""
    s = np.uint64((seed * 2654435761 + 12345) & 0xFFFFFFFF) | np.uint64(1)
    idx = np.arange(n, dtype=np.uint64)
    x = (idx + s) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33)
    noise = ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)) * 2.0 - 1.0
    # triangle waves at two periods chosen from the seed
    p1, p2 = 37 + seed % 23, 211 + (seed * 7) % 101
    tri = lambda p: 2.0 * np.abs(2.0 * ((idx % np.uint64(p)).astype(np.float64) / p) - 1.0) - 1.0
    return (scale * (0.5 * tri(p1) + 0.3 * tri(p2) + 0.1 * noise)).astype(np.float32)


CASES_FFT = [(44100, 16384, 1), (48000, 4096, 2), (96000, 16384, 3), (48000, 256, 4)]
CASES_WAVE = [(44100, 15.0, 5), (9600, 0.1, 6), (1000, 0.3, 7)]
CASES_METER = [(2, 48000, 4.0, 8), (2, 44100, 3.5, 9), (1, 48000, 2.0, 10), (6, 48000, 2.0, 11)]
BATCH = dict(rate=48000, frames=48000 * 2 + 500, fft_n=4096, hop=1024, seeds=(21, 22))


def build():
    from oracle import pyoracle as po
    out = {}
    for rate, n, seed in CASES_FFT:
        out[f"fft_{rate}_{n}_{seed}"] = po.get_fft(rate, golden_input(seed, n))
    for n, win, seed in CASES_WAVE:
        out[f"wave_{n}_{win}_{seed}"] = po.get_waveform(golden_input(seed, n), win)
    for ch, rate, secs, seed in CASES_METER:
        x = golden_input(seed, int(rate * secs) * ch, 0.6)
        m = po.Meter(ch, rate)
        st = []
        step = 16384 - (16384 % ch)
        for off in range(0, x.size, step):
            m.add_frames(x[off:off + step])
            st.append(m.shortterm())
        out[f"meter_{ch}_{rate}_{seed}"] = np.array(
            [m.integrated(), m.loudness_range(), m.momentary()] + [m.true_peak(c) for c in range(ch)] +
            [m.sample_peak(c) for c in range(ch)], np.float64)
        out[f"meter_st_{ch}_{rate}_{seed}"] = np.array(st, np.float64)
        out[f"meter_hist_{ch}_{rate}_{seed}"] = np.nonzero(m.block_hist())[0].astype(np.int32)
    for seed in BATCH["seeds"]:
        x = golden_input(seed, BATCH["frames"] * 2, 0.7)
        r = po.analyze_stream(BATCH["rate"], x, BATCH["fft_n"], BATCH["hop"])
        # keep three windows of the spectrum (first, middle, last) to bound the fixture size
        w = r["n_windows"]
        out[f"batch_fft_{seed}"] = r["fft"][[0, w // 2, w - 1]].astype(np.float32)
        out[f"batch_scalars_{seed}"] = np.array([r["integrated"], r["lra"], *r["true_peak"], *r["sample_peak"], w, r["n_bins"]], np.float64)
        out[f"batch_wave_{seed}"] = r["wave"][:, 1].astype(np.float32)
    return out


if __name__ == "__main__":
    data = build()
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes,", len(data), "arrays")
