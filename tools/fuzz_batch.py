#!/usr/bin/env python3
"""Randomised batch programmes against the oracle: random rate, channel count, stream count (1 .. 130), length (a few ms to 25 s, never
round), window / hop, true-peak factor and arithmetic, time-domain hand-over mode, now and then ragged lengths — the shapes the
batch's geometry rules (segments, fix-up launch, split segments, whole-stream workgroups, small-batch gating, spectrum run lengths)
switch on.  Per stream: integrated loudness, range, every channel's true and sample peak, the decimated waveform bit for bit, three
spectrum rows.      python tools/fuzz_batch.py [programmes] [first seed] [--big] [--wide] [--reuse] [-v]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
from conftest import db_close, db_report, make_multich, window_peak_db

RATES = [8000, 22050, 32000, 44100, 48000, 48000, 48000, 88200, 96000, 96000, 192000]
CHANNELS = [1, 2, 2, 2, 2, 3, 6, 8, 8]
STREAMS = [1, 1, 2, 3, 7, 20, 64, 65, 130]


def lufs_close(a, b, tol=0.01):
    if np.isinf(a) or np.isinf(b) or np.isnan(a) or np.isnan(b):
        return a == b or (np.isnan(a) and np.isnan(b))
    return abs(a - b) <= tol


_pink = {}
def pink_of(rate, n):
    """the pink-noise compensation the retained bins of (rate, n) carry: 10 log10(f / 1000), f as the reference computes it (f32)"""
    if (rate, n) not in _pink:
        cnt, first = po.fft_bins(rate, n)
        f = (np.arange(cnt) + first) * (np.float32(rate) / np.float32(n))
        _pink[(rate, n)] = 10 * np.log10(f.astype(np.float64) / 1000.0)
    return _pink[(rate, n)]


def plan(seed):
    """the programme's parameters and contents (no device needed)"""
    rng = np.random.default_rng(seed)
    rate = int(rng.choice(RATES)); ch = int(rng.choice(CHANNELS)); ns = int(rng.choice(STREAMS))
    big = "--big" in sys.argv                                # long streams: many segments, the fix-up launch, long spectrum runs
    wide = "--wide" in sys.argv                              # grids of more than 768 workgroups: the four-waves-per-SIMD kernel builds
    if wide: ns = int(rng.choice([700, 1024, 1500, 3000]))
    secs = float(np.exp(rng.uniform(np.log(2.0 if big else 0.02), np.log(40.0 if big else (1.5 if wide else 25.0)))))
    budget = 24_000_000 if big else 6_000_000                # samples of DISTINCT content per programme (the oracle's time)
    kinds = int(min(ns, rng.integers(1, 5)))
    slot = max(1, min(int(rate * secs) + int(rng.integers(0, 97)), budget // (kinds * ch)))
    while ns * slot * ch > (300_000_000 if wide else 400_000_000): slot = max(1, slot // 2) if wide else slot; ns = ns if wide else max(1, ns // 2)
    kinds = min(kinds, ns)
    fft_ok = rate >= 40000
    fft_n, hop = [(4096, 1024), (4096, 1024), (4096, 512), (4096, 1000), (16384, 1024), (2048, 512)][int(rng.integers(0, 6))]
    flags = L.SS_BATCH_ALL if fft_ok else (L.SS_BATCH_ALL & ~L.SS_BATCH_FFT)
    if rng.random() < 0.15: flags &= ~L.SS_BATCH_WAVEFORM
    if rng.random() < 0.15: flags &= ~L.SS_BATCH_TRUE_PEAK
    tpf = int(rng.choice([0, 0, 0, 2, 4]))
    mode = int(rng.choice([L.SS_TD_AUTO, L.SS_TD_AUTO, L.SS_TD_RUN_IN, L.SS_TD_WHOLE_STREAMS]))
    arith = int(rng.choice([L.SS_TP_ARITH_F32, L.SS_TP_ARITH_F16X3]))
    ragged = rng.random() < 0.3
    what = f"seed {seed}: {rate} Hz x {ch} ch x {ns} streams x {slot} frames, N {fft_n} hop {hop}, flags {flags}, tp {tpf}, mode {mode}, arith {arith}, ragged {ragged}"
    content = []
    for k in range(kinds):
        x = make_multich(seed * 7 + k, slot, ch, rate, level=float(rng.uniform(0.02, 0.9)))
        # a near-silent passage (absolute gate) OR a DC offset (what a truncated hand-over would show) — not both: under a DC term
        # 70 dB above everything else a window's retained bins ARE the transform's rounding noise (bins 0 and 1 are not retained,
        # so the row-relative metric does not see the term), in the reference's f32 FFT as in any other (seed 40 of the first run)
        if rng.random() < 0.3:
            g0 = int(rng.integers(0, slot)); x.reshape(slot, ch)[g0:g0 + slot // 3] *= 1e-4
        elif rng.random() < 0.3: x += np.float32(0.05)
        r2 = np.random.default_rng(seed * 11 + k + 10 ** 6).random()      # (its own generator: the seeds of earlier runs keep their shapes)
        if r2 < 0.12 and ch == 2: x[1::2] = x[0::2]                        # dual mono: the side row is the -150 dB floor + pink
        elif r2 < 0.24: x[: ch * (slot // 2)] = 0.0                        # digital silence in front of the programme: empty rows, the floor again
        elif r2 < 0.30 and ch > 1: x.reshape(slot, ch)[:, ch - 1] = 0.0    # one silent channel
        # non-finite samples (round 6): one to three of NaN / +Inf / -Inf anywhere, any channel (weighted or not), the programme 20 dB
        # louder behind the first of them — in the crate the channel's filter state is NaN from there on and no later block counts
        r4 = np.random.default_rng(seed * 13 + k + 4 * 10 ** 6)
        if r4.random() < 0.25 and "--finite" not in sys.argv:
            xm = x.reshape(slot, ch)
            at = sorted(int(v) for v in r4.integers(0, slot, int(r4.integers(1, 4))))
            if r4.random() < 0.7: xm[: at[0]] *= np.float32(0.1)
            for f in at: xm[f, int(r4.integers(0, ch))] = [np.nan, np.inf, -np.inf][int(r4.integers(0, 3))]
        content.append(x)
    lens = [slot] * ns
    if ragged:
        lens = [int(rng.choice([slot, slot, int(rng.integers(0, slot + 1)), slot // 2, min(slot, fft_n + hop), min(slot, 4799), 0, 1])) for _ in range(ns)]
    return dict(rng=rng, rate=rate, ch=ch, ns=ns, slot=slot, kinds=kinds, fft_n=fft_n, hop=hop, flags=flags, tpf=tpf, mode=mode, arith=arith,
                ragged=ragged, what=what, content=content, lens=lens)


def programme(seed):
    P = plan(seed)
    rng, rate, ch, ns, slot, kinds, fft_n, hop, flags, tpf, mode, arith, ragged, what, content, lens = (P[k] for k in (
        "rng", "rate", "ch", "ns", "slot", "kinds", "fft_n", "hop", "flags", "tpf", "mode", "arith", "ragged", "what", "content", "lens"))
    try:
        b = ssa.Batch(rate, ch, ns, slot, fft_n, hop, flags=flags, true_peak_factor=tpf)
    except ssa.AnalyzerError as e:
        return "device" not in str(e).lower(), what + f" -> refused at create ({e})"
    try: b.set_time_domain_mode(mode)
    except ssa.AnalyzerError: pass
    b.set_true_peak_arith(arith)
    ov = int(np.random.default_rng(seed + 3 * 10 ** 6).choice([0, 0, 1, 2]))      # the spectrum kernel beside the time-domain chain (opt-in modes)
    if ov: b.set_overlap(ov)
    lay = b.layout
    refs = {}
    ok, notes = True, []
    def bad(msg):
        nonlocal ok
        ok = False; notes.append(msg)
    # the batch is used up to three times ("--reuse"): new lengths (ragged from then on), the contents dealt differently, and the
    # hand-over mode switched — what a pipeline that keeps its batch does (the one-shot loudness call keeps one)
    r3 = np.random.default_rng(seed + 2 * 10 ** 6)
    passes = 1 + (int(r3.integers(0, 3)) if "--reuse" in sys.argv else 0)
    for pass_no in range(passes):
      rot = 0
      if pass_no:
        ragged = True
        lens = [int(r3.choice([slot, int(r3.integers(0, slot + 1)), slot // 3, 0])) for _ in range(ns)]
        rot = int(r3.integers(0, kinds))
        try: b.set_time_domain_mode(int(r3.choice([L.SS_TD_AUTO, L.SS_TD_RUN_IN, L.SS_TD_WHOLE_STREAMS])))
        except ssa.AnalyzerError: pass
      if ragged: b.set_lengths(lens)
      buf = np.full((ns, slot * ch), 7.0, np.float32)
      for i in range(ns): buf[i, :lens[i] * ch] = content[(i + rot) % kinds][:lens[i] * ch]
      b.upload(0, buf.reshape(-1))
      b.run(); b.sync()
      res = b.results()
      check = sorted(set([0, ns - 1] + [int(v) for v in rng.integers(0, ns, 6)]))
      for i_ in check:
        i = i_
        n = lens[i]; key = ((i + rot) % kinds, n)
        x = content[(i + rot) % kinds][:n * ch]
        if key not in refs:
            r = {}
            if n:
                m = po.Meter(ch, rate, force_tp_factor=tpf); m.add_frames(x)
                r["I"], r["lra"] = m.integrated(), m.loudness_range()
                r["sp"] = [m.sample_peak(c) for c in range(ch)]; r["tp"] = [max(m.true_peak(c), r["sp"][c]) for c in range(ch)]
                r["wave"] = po.get_waveform(x, n / rate)
            refs[key] = r
        r = refs[key]
        if n == 0:
            if not (res[i].integrated_lufs == -np.inf and res[i].loudness_range == 0.0): bad(f"pass {pass_no} stream {i}: empty stream reads {res[i].integrated_lufs} {res[i].loudness_range}")
            continue
        if flags & L.SS_BATCH_LUFS:
            if not lufs_close(res[i].integrated_lufs, r["I"]): bad(f"pass {pass_no} stream {i}: I {res[i].integrated_lufs} vs {r['I']}")
            if not abs(res[i].loudness_range - r["lra"]) <= 0.01: bad(f"pass {pass_no} stream {i}: LRA {res[i].loudness_range} vs {r['lra']}")
        tp, sp = b.peaks(i)
        if flags & L.SS_BATCH_TRUE_PEAK:
            for c in range(ch):
                if not (tp[c] == r["tp"][c] or abs(tp[c] - r["tp"][c]) <= 1e-4 * max(abs(r["tp"][c]), 1e-30)): bad(f"pass {pass_no} stream {i} ch {c}: true peak {tp[c]} vs {r['tp'][c]}")
                if sp[c] != r["sp"][c]: bad(f"pass {pass_no} stream {i} ch {c}: sample peak {sp[c]} vs {r['sp'][c]}")
        if flags & L.SS_BATCH_WAVEFORM:
            w = b.waveform(i).reshape(-1)
            want = r["wave"][:, 1].astype(np.float32)
            if ragged: w = w[:want.size]
            if not (w.size == want.size and np.array_equal(w, want, equal_nan=True)): bad(f"pass {pass_no} stream {i}: waveform differs ({w.size} vs {want.size} points)")
        if flags & L.SS_BATCH_FFT:
            nw = max(0, n // hop - fft_n // hop)
            got_nw = b.stream_shape(i).n_windows if ragged else lay.n_windows
            if got_nw != nw: bad(f"pass {pass_no} stream {i}: {got_nw} windows vs {nw}")
            elif nw:
                fft = b.fft(i)
                xm = x.reshape(n, ch)
                if ch == 2: sig = po.mid_side(x)
                else: sig = [np.ascontiguousarray(xm[:, c]) for c in range(ch)]
                for wdx in sorted(set([0, nw // 2, nw - 1])):
                    start = (wdx + fft_n // hop + 1) * hop - fft_n      # positions p = k hop with N < p <= F, window [p - N, p) (tui.rs:1489)
                    for c in sorted(set([0, len(sig) - 1])):
                        try: ref = po.get_fft(rate, sig[c][start:start + fft_n])[:, 1]
                        except po.OracleError: continue                 # (a non-finite sample inside the window: the crate refuses it)
                        # (0.015 dB here, 0.01 in the committed tests: at the metric's edge, 70 dB under the row's peak, the difference of two
                        # f32 transforms' rounding noise is 0.004 dB typical — over thousands of random rows the tail reaches 0.011, seed 102143)
                        # (a row that misses is looked at again with the 70 dB counted from the window's strongest component over ALL bins:
                        # a DC offset 20 dB above the retained band's peak — bins 0 ... 6 of N = 16384 are not in the row — sets the
                        # rounding noise of any f32 transform of that window; seed 620369: 0.021 dB at bins 86 dB under the DC term)
                        if not db_close(fft[wdx, c], ref, 0.015) and not db_close(fft[wdx, c], ref, 0.015, peak=window_peak_db(po, sig[c][start:start + fft_n]), pink=pink_of(rate, fft_n)): bad(f"pass {pass_no} stream {i} window {wdx} ch {c}: spectrum row differs {db_report(fft[wdx, c], ref)} row peak {float(ref.max()):.1f} dB")
    g = b.geometry
    b.close()
    return ok, what + f" [segments {g.td_segments} x {g.td_segment_subblocks}, split {g.td_split}, fixup {g.td_fixup_subblocks}]" + ("" if ok else " -> " + "; ".join(notes[:6]))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failed = 0
    for seed in range(first, first + n):
        try:
            ok, msg = programme(seed)
        except Exception as e:                               # noqa: BLE001 — a crash of one programme is a finding, not the end of the run
            ok, msg = False, f"seed {seed}: exception {type(e).__name__}: {e}"
        if not ok:
            failed += 1
        if not ok or "-v" in sys.argv or "refused" in msg:
            print(("ok   " if ok else "FAIL ") + msg, flush=True)
    print(f"{n} batch programmes, {failed} failed")
    sys.exit(1 if failed else 0)
