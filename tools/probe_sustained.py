#!/usr/bin/env python3
"""The timed step back to back for half a minute (the bench line's `config.sustained` is one second): throughput and shader clock in
five-second slices — does the chip hold its clock under this path?      python tools/probe_sustained.py [seconds]"""
import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soundscope_amd as ssa
from soundscope_amd import _lib as L
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
b = ssa.Batch(48000, 2, 1024, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
for _ in range(5): b.run(); b.sync()


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showtemp", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        sclk = re.search(r"GPU\[0\].*sclk clock level.*\((\d+)Mhz\)", out)
        temp = re.search(r"GPU\[0\].*Temperature \(Sensor junction\) \(C\): ([\d.]+)", out)
        pw = re.search(r"GPU\[0\].*(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
        return (sclk.group(1) if sclk else "?"), (temp.group(1) if temp else "?"), (pw.group(1) if pw else "?")
    except Exception as e:                                   # noqa: BLE001
        return "?", "?", str(e)[:40]


t_end = time.perf_counter() + secs
slice_s = 5.0
while time.perf_counter() < t_end:
    t0 = time.perf_counter(); n = 0
    mid = None
    while time.perf_counter() - t0 < slice_s:
        for _ in range(20): b.run()
        n += 20
        if mid is None and time.perf_counter() - t0 > slice_s / 2:
            mid = clocks()                                   # (asked while the queue is full: the clock under load)
        b.sync()
    dt = time.perf_counter() - t0
    print(f"{n} steps in {dt:.2f} s: {dt / n * 1e3:.3f} ms per step, {1024 * 960000 * n / dt / 1e9:.1f} G samples/s; sclk {mid[0]} MHz, junction {mid[1]} C, power {mid[2]} W", flush=True)
