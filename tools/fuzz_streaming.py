#!/usr/bin/env python3
"""Randomised streaming programmes against the oracle: random rate and channel count, a few dozen add_samples calls of random
length (one frame to a third of a second: calls shorter than a tile, tiles that end inside a chunk, calls of many rounds of
tiles), material with level jumps and silences; after every call short-term and momentary loudness and the carried filter state,
at the end integrated loudness, loudness range and the peaks.      python tools/fuzz_streaming.py [programmes] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from oracle import pyoracle as po

def close_lu(a, b, tol):
    if np.isinf(a) or np.isinf(b) or np.isnan(a) or np.isnan(b):
        return (a == b) or (np.isnan(a) and np.isnan(b))
    if b < -200.0: return abs(a - b) <= 0.01            # the decay of digital silence (-700 LUFS and below, 130 dB under the absolute gate): the
                                                       # state there carries the chunk scan's 1e-4 .. 1e-3 at 192 kHz (below), its energy twice that
    return abs(a - b) <= tol + 1e-8 * abs(b)

def programme(seed):
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([22050, 32000, 44100, 48000, 88200, 96000, 192000]))
    ch = int(rng.choice([1, 2, 2, 2, 3, 6, 8]))
    an = ssa.Analyzer(); an.create_loudness_meter(ch, rate)
    mm = po.Meter(ch, rate)
    total = 0
    worst = {"st": 0.0, "mom": 0.0, "state": 0.0}
    peak_scale = 0.0
    n_calls = int(rng.integers(8, 40))
    # non-finite samples (round 6): in a quarter of the programmes ONE call carries a NaN / +Inf / -Inf on a random channel (a weighted
    # one or one the crate does not filter): readings and the carried state's NaN pattern are compared after every call behind it
    r4 = np.random.default_rng(seed + 4 * 10 ** 6)
    bad_call = int(r4.integers(0, n_calls)) if (r4.random() < 0.25 and "--finite" not in sys.argv) else -1
    for k in range(n_calls):
        kind = rng.integers(0, 6)
        if kind == 0: frames = int(rng.integers(1, 64))
        elif kind == 1: frames = int(rng.integers(64, 2000))
        elif kind == 2: frames = 16384 // ch
        else: frames = int(rng.integers(2000, rate // 3))
        level = 10.0 ** (rng.uniform(-70, -3) / 20.0)
        t = (np.arange(frames) + total) / rate
        x = np.empty((frames, ch), np.float32)
        for c in range(ch):
            f0 = rng.uniform(40, 5000)
            x[:, c] = level * (np.sin(2 * np.pi * f0 * t + c) + 0.3 * rng.standard_normal(frames))
        silent = rng.integers(0, 7) == 0
        if silent: x[:] = 0.0
        if k == bad_call: x[int(r4.integers(0, frames)), int(r4.integers(0, ch))] = [np.nan, np.inf, -np.inf][int(r4.integers(0, 3))]
        xs = np.ascontiguousarray(x.reshape(-1))
        an.add_samples(xs); mm.add_frames(xs)
        total += frames
        for name, g, o, tol in (("st", an.get_shortterm_lufs(), mm.shortterm(), 1e-6), ("mom", an.get_momentary_lufs(), mm.momentary(), 1e-6)):
            if not close_lu(g, o, tol): return f"seed {seed} ({rate} Hz, {ch} ch) call {k} ({frames} frames): {name} {g} vs {o}"
            if np.isfinite(g) and np.isfinite(o): worst[name] = max(worst[name], abs(g - o))
        if bad_call >= 0:
            for c in range(ch):
                if not np.array_equal(np.isnan(an.filter_state(c)), np.isnan(mm.filter_state(c))):
                    return f"seed {seed} ({rate} Hz, {ch} ch) call {k}: channel {c} state {an.filter_state(c)} vs {mm.filter_state(c)}"
        gs, os_ = an.filter_state(0), mm.filter_state(0)
        scale = np.max(np.abs(os_))
        if scale > 1e-280:
            d = float(np.max(np.abs(gs - os_)) / scale)
            worst["state"] = max(worst["state"], d)
            if "-v" in sys.argv: print(f"   call {k}: {frames} frames, level {20 * np.log10(level):.0f} dB{', SILENT' if silent else ''}: |state| {scale:.3e}, off by {d:.2e}", flush=True)
            # (192 kHz: the chunk scan's difference coordinates are good for 1e-4 .. 1e-3 of a state that has decayed through a
            # silence — the same on one wave and on eight; the readings above do not see it)
            peak_scale = max(peak_scale, scale)
            # ... and a state that has decayed through one silence after another (seed 70123: from 2e4 to 2e-37) is held to the
            # programme's own scale: an error 1e-24 of the largest state the filter has carried is below anything a sample can feel
            if d > (1e-5 if rate <= 96000 else 2e-3) and d * scale > 1e-24 * peak_scale: return f"seed {seed} ({rate} Hz, {ch} ch) call {k} ({frames} frames): state off by {d:.2e} of its largest component"
    gi, oi = an.get_integrated_lufs(), mm.integrated()
    if not close_lu(gi, oi, 1e-6): return f"seed {seed}: integrated {gi} vs {oi}"
    if not close_lu(an.get_loudness_range(), mm.loudness_range(), 1e-6): return f"seed {seed}: range {an.get_loudness_range()} vs {mm.loudness_range()}"
    for c in range(ch):
        gp = an.get_true_peak_channel(c) if hasattr(an, "get_true_peak_channel") else None
        if gp is not None:
            want = max(mm.true_peak(c), mm.sample_peak(c))
            if gp != want and not abs(gp - want) <= 1e-4 * max(want, 1e-30): return f"seed {seed} ch {c}: true peak {gp} vs {want}"
    return worst

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    tot = {"st": 0.0, "mom": 0.0, "state": 0.0}
    for seed in range(first, first + n):
        r = programme(seed)
        if isinstance(r, str):
            print("FAIL", r, flush=True); bad += 1
        else:
            for k in tot: tot[k] = max(tot[k], r[k])
    print(f"{n} programmes, {bad} failed; worst differences: short-term {tot['st']:.1e} LU, momentary {tot['mom']:.1e} LU, state {tot['state']:.1e}")
