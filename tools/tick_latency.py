#!/usr/bin/env python3
"""Latency of one reference tick (tui.rs:1482-1552) through the C ABI: get_fft(mid) + get_fft(side)
(N = 16384) + add_samples(last 16384 interleaved samples) + get_shortterm_lufs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import soundscope_amd as ssa
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 12, rate)
mid, side = ssa.get_mid_and_side_samples(x)
an = ssa.Analyzer(); an.create_loudness_meter(2, rate)
t_fft, t_add, t_st, t_all = [], [], [], []
for tick, pos in enumerate(range(16384 * 2 + 2048, x.size, 2048)):
    fpos = pos // 2
    t0 = time.perf_counter()
    an.get_fft(mid[fpos - 16384:fpos]); an.get_fft(side[fpos - 16384:fpos])
    t1 = time.perf_counter()
    an.add_samples(x[pos - 16384:pos])
    t2 = time.perf_counter()
    an.get_shortterm_lufs()
    t3 = time.perf_counter()
    if tick >= 20:
        t_fft.append(t1 - t0); t_add.append(t2 - t1); t_st.append(t3 - t2); t_all.append(t3 - t0)
    if tick > 300:
        break
f = lambda v: f"median {np.median(v) * 1e6:.0f} us, p99 {np.percentile(v, 99) * 1e6:.0f} us"
print("2 x get_fft(16384):", f(t_fft)); print("add_samples(16384):", f(t_add)); print("get_shortterm_lufs:", f(t_st))
print("whole tick        :", f(t_all), "(budget: 8 ms TUI loop + 21 ms between ticks at 48 kHz)")
t0 = time.perf_counter(); w = ssa.Analyzer.get_waveform(x, 12.0); t1 = time.perf_counter()
v = an.calculate_integrated_lufs(2, x); t2 = time.perf_counter()
print(f"file load (12 s stereo): get_waveform {(t1 - t0) * 1e3:.2f} ms, calculate_integrated_lufs {(t2 - t1) * 1e3:.2f} ms")

# the same tick as ONE call on a file session (audio resident in HBM, nothing uploaded per tick)
t0 = time.perf_counter(); sess = ssa.FileSession(x, 2, rate); t1 = time.perf_counter()
print(f"FileSession open (upload + waveform + integrated gain): {(t1 - t0) * 1e3:.2f} ms")
t_sess = []
for tick, pos in enumerate(range(16384 * 2 + 2048, x.size, 2048)):
    t0 = time.perf_counter()
    sess.analyze_audio_file_samples(pos)
    t1 = time.perf_counter()
    time.sleep(0.0003)        # (ticks are 21 ms apart; each leaves the gating and the render loop's readings running behind its results)
    if tick >= 20:
        t_sess.append(t1 - t0)
    if tick > 300:
        break
print("session tick      :", f(t_sess))
cap = ssa.CaptureSession(2, rate)
ring = np.concatenate([x, x, x])[:30 * rate]
t_cap = []
for tick in range(60):
    t0 = time.perf_counter()
    cap.analyze_microphone_input(ring)
    t1 = time.perf_counter()
    if tick >= 10:
        t_cap.append(t1 - t0)
print("capture tick (30 s ring upload + 15 s waveform):", f(t_cap))
# the same with the snapshot in memory the caller has page-locked once (ss_host_register): the upload is a DMA, not a staged copy
import ctypes as C
from soundscope_amd import _lib as L
if L.lib().ss_host_register(ring.ctypes.data_as(C.c_void_p), ring.nbytes) == L.SS_OK:
    t_cap = []
    for tick in range(60):
        t0 = time.perf_counter()
        cap.analyze_microphone_input(ring)
        t1 = time.perf_counter()
        if tick >= 10:
            t_cap.append(t1 - t0)
    print("capture tick, page-locked snapshot             :", f(t_cap))
    L.lib().ss_host_unregister(ring.ctypes.data_as(C.c_void_p))
# the getters of the reference's render loop (tui.rs:917, :950, :969: every frame): first reading of a meter state, then repeats
t_first, t_rep = [], []
for k in range(40):
    an.add_samples(x[:16384])
    t0 = time.perf_counter(); an.get_integrated_lufs(); an.get_true_peak(); an.get_loudness_range(); t1 = time.perf_counter()
    an.get_integrated_lufs(); an.get_true_peak(); an.get_loudness_range(); t2 = time.perf_counter()
    t_first.append(t1 - t0); t_rep.append(t2 - t1)
print("render-loop getters (integrated + true peak + range): first reading of a state", f(t_first), "| again", f(t_rep))
# the same getters on a file session's analyzer right behind a tick (the tick has enqueued them)
t_g = []
for k, pos in enumerate(range(16384 * 2 + 2048, x.size, 2048)):
    sess.analyze_audio_file_samples(pos)
    time.sleep(0.0002)                                   # (the render comes a few hundred microseconds later at the earliest)
    t0 = time.perf_counter(); sess.analyzer.get_integrated_lufs(); sess.analyzer.get_true_peak(); sess.analyzer.get_loudness_range(); t1 = time.perf_counter()
    t_g.append(t1 - t0)
    if k > 200: break
print("render-loop getters on the session's analyzer behind a tick:", f(t_g))
# the capture tick on a ring that lives on the device: the callback pushes 1024 frames per tick, nothing else crosses PCIe
cap2 = ssa.CaptureSession(2, rate)
chunk = ring[:2048]
t_cap = []
for tick in range(120):
    cap2.push(chunk)
    t0 = time.perf_counter()
    cap2.analyze_resident()
    t1 = time.perf_counter()
    if tick >= 20:
        t_cap.append(t1 - t0)
print("capture tick, ring resident on the device      :", f(t_cap))
