#!/bin/bash
# round 5: the same-box A/B calls behind profiles/r05_ab_*.txt, one function per GPU call (gpurun -- 'tools/ab_round5.sh <step>').
# Variant builds of the library are expected in tools/bin/ (git-ignored; make -C soundscope_amd/csrc OBJDIR=build_x OUT=$PWD/tools/bin/x.so
# EXTRA="-D..."): tw9 / tw12 (-DSS_MS1_TW=9 / 12), cols1 (-DSS_COLS_COPIES=1, an earlier form of the columns epilogue), fold
# (-DSS_COLS_FOLD=1 before it became the default), head (the previous commit), r3fft_now (round 3's ss_fft.hip against today's host
# code), tune (-DSS_TUNING).
#   a  suite + k_fft4096_ms1: TW6 / TW9 / TW12 / round-3 kernel file / previous commit; columns-only: new epilogue, 1 / 4 copies
#   b  columns-only with float LDS atomics: one per bin / folded / previous commit, PMC passes over the columns probe
#   c  columns tests + k_fft16k_run at config 5: narrower last epilogue iteration vs previous commit, run-length sweep
#   d  columns tests and timing of the shipped form
#   f  whole-stream workgroups vs time segments at the bench shape (tools/probe_handover.py supersedes it)
#   g  hand-over modes against the one-segment path, mismatches by sub-block index
#   i  small-batch gating (k_finalize<SMALL>, tables in LDS): loudness tests, then head.so vs the tree on the one-stream probes
#   h  k_time_domain's two register builds on big grids, shape by shape (tools/probe_td_wps.py, needs tools/bin/tune.so)
set -u
step=${1:?a|b|c|d|f|g|h|i}

step_a() {
# round 5, call A: suite with the f32 default, then same-box A/B of k_fft4096_ms1 builds (TW6 / TW9 / TW12 resident pass-1 twiddles,
# the round-3 kernel file, the previous commit) and of the columns-only kernel (new epilogue with 4 / 1 accumulator copies, previous)
out=gpurun_out/r5a; mkdir -p $out
python -m pytest tests -m gpu -q -x > $out/suite.log 2>&1; tail -3 $out/suite.log
for rep in 1 2; do
for lib in default tw9 tw12 r3fft_now head; do
  echo "=== full rows: $lib (rep $rep)"
  if [ $lib = default ]; then python tools/perf_probe.py 1024 20; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 20; fi
done
for lib in default cols1 head; do
  echo "=== columns only (160): $lib (rep $rep)"
  if [ $lib = default ]; then python tools/perf_probe.py 1024 20 --cols=160; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 20 --cols=160; fi
done
done > $out/ab.log 2>&1
cat $out/ab.log | grep -E "===|k_fft|k_time"
}

step_b() {
# round 5, call B: columns-only kernel with float LDS atomics (per bin / folded in the VALU first) against the previous commit's
out=gpurun_out/r5b; mkdir -p $out
python -m pytest tests/test_gpu_columns.py tests/test_gpu_dynamic_range.py -m gpu -q -x > $out/cols_tests.log 2>&1; tail -3 $out/cols_tests.log
for rep in 1 2; do
for lib in default fold head; do
  echo "=== columns only (160): $lib (rep $rep)"
  if [ $lib = default ]; then python tools/perf_probe.py 1024 20 --cols=160; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 20 --cols=160; fi
done
echo "=== full rows: default (rep $rep)"; python tools/perf_probe.py 1024 20
done > $out/ab.log 2>&1
grep -E "===|k_fft|k_time" $out/ab.log
PROBE=tools/perf_probe.py tools/pmc_passes.sh r5b/pmc_cols "1024 2 --cols=160" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" 2>&1 | grep -E "##|fft4096" | cut -c1-150
}

step_c() {
# round 5, call C: columns-only default (fold + general groups) and the 16384-point run kernel's narrower last epilogue iteration
out=gpurun_out/r5c; mkdir -p $out
python -m pytest tests/test_gpu_columns.py tests/test_gpu_bench_shapes.py tests/test_gpu_independent.py -m gpu -q -x > $out/tests1.log 2>&1; tail -3 $out/tests1.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16384 or 16k or native or rate or mono or channels" > $out/tests2.log 2>&1; tail -3 $out/tests2.log
for rep in 1 2; do
for lib in default head; do
  echo "=== config 5: $lib (rep $rep)"
  if [ $lib = default ]; then python tools/probe_cfg5.py 64; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/probe_cfg5.py 64; fi
done
done > $out/ab5.log 2>&1
grep -E "===|fft16k" $out/ab5.log
for g in 2 4 8 16; do echo "=== config 5, runs: SS_FFT16K_GROUPS=$g (tuning build)"; SS_FFT16K_GROUPS=$g SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so python tools/probe_cfg5.py 64 | grep -E "fft16k"; done > $out/groups.log 2>&1
cat $out/groups.log
echo "=== columns only, default"; python tools/perf_probe.py 1024 20 --cols=160 | grep -E "k_fft|k_time"
}

step_d() {
out=gpurun_out/r5d; mkdir -p $out
python -m pytest tests/test_gpu_columns.py tests/test_gpu_bench_shapes.py tests/test_gpu_independent.py tests/test_gpu_reference_suite.py -m gpu -q -x > $out/tests1.log 2>&1; tail -3 $out/tests1.log
for rep in 1 2; do echo "=== columns only, default (rep $rep)"; python tools/perf_probe.py 1024 20 --cols=160 | grep -E "k_fft|k_time"; done
echo "=== full rows"; python tools/perf_probe.py 1024 20 | grep -E "k_fft|k_time"
}

step_f() {
# round 5, call F: whole-stream workgroups (exact state hand-over) against time segments at the bench shape
out=gpurun_out/r5f; mkdir -p $out
python - > $out/split.log 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
from oracle import pyoracle as po
ns = 1024
b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=L.SS_BATCH_ALL)
b.synthesize(0x5EED0000, 0)
res = {}
for mode in (1, 2, 1, 2):
    b.set_time_domain_mode(mode)
    g = b.geometry
    b.run(); b.sync()
    b.timing_enable(True)
    for _ in range(10):
        b.run(); b.sync()
    ms, n = b.timing_read(L.SS_KERNEL_TIME_DOMAIN)
    fms, fn = b.timing_read(L.SS_KERNEL_FFT)
    b.timing_enable(False)
    print("mode", mode, "td_split", g.td_split, "segments", g.td_segments, "k_time_domain %.4f ms" % (ms / n), "fft %.4f" % (fms / fn), flush=True)
    res[mode] = (b.results(), [b.subblocks(i).copy() for i in (0, 1, 511, 1023)], [b.waveform(i).copy() for i in (0, 1023)], [b.peaks(i) for i in (0, 1023)])
r1, r2 = res[1], res[2]
print("integrated max diff", max(abs(a.integrated_lufs - c.integrated_lufs) for a, c in zip(r1[0], r2[0])))
print("subblock rel diff", max(float(np.max(np.abs(a - c) / np.maximum(np.abs(c), 1e-300))) for a, c in zip(r1[1], r2[1])))
print("waveform equal", all(np.array_equal(a, c) for a, c in zip(r1[2], r2[2])))
print("peaks", r1[3], r2[3])
for i in (0, 1023):
    x = b.download_input(i)
    ref = po.analyze_stream(48000, x, 4096, 1024)
    r = r2[0][i]
    print("stream", i, "vs oracle: I", r.integrated_lufs - ref["integrated"], "LRA", r.loudness_range - ref["lra"], "TP rel", (r.true_peak[0] - ref["true_peak"][0]) / ref["true_peak"][0],
          "wave", np.array_equal(b.waveform(i).reshape(-1), ref["wave"][:, 1].astype(np.float32)))
    m = po.Meter(2, 48000); m.add_frames(x)
PY
cat $out/split.log
}

step_g() {
out=gpurun_out/r5g; mkdir -p $out
python - > $out/modes2.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import soundscope_amd as ssa
from soundscope_amd import _lib as L
ns = 1024
FL = L.SS_BATCH_LUFS | L.SS_BATCH_TRUE_PEAK | L.SS_BATCH_WAVEFORM
b = ssa.Batch(48000, 2, ns, 480000, 4096, 1024, flags=FL)
b.synthesize(0x5EED0000, 0)
res = {}
for mode in (0, 1, 2):
    b.set_time_domain_mode(mode)
    b.run(); b.sync()
    res[mode] = np.stack([b.subblocks(i) for i in range(ns)])
one = ssa.Batch(48000, 2, 4096, 480000, 4096, 1024, flags=FL)
one.synthesize(0x5EED0000, 0)
one.run(); one.sync()
g = one.geometry
print("reference batch: streams 4096 segments", g.td_segments, "split", g.td_split)
ref = np.stack([one.subblocks(i) for i in range(ns)])
x0 = b.download_input(0); x1 = one.download_input(0)
print("same input:", np.array_equal(x0, x1))
for mode in (0, 1, 2):
    d = res[mode].reshape(ns, 100, 2)
    r = ref.reshape(ns, 100, 2)
    neq = (d != r)
    print("mode", mode, "mismatching sub-blocks by index (sum over streams, both channels):")
    print("   ", neq.sum(axis=(0, 2)).tolist())
    rel = np.abs(d - r) / np.maximum(np.abs(r), 1e-300)
    print("    max rel by sub-block index:", ["%.1e" % v for v in rel.max(axis=(0, 2))][:30])
# independent f64 reference for stream 0: scipy lfilter with the library's coefficients
from scipy.signal import lfilter
from oracle import pyoracle as po
m = po.Meter(2, 48000)
bb, aa = m.filter_coeffs()
x = b.download_input(0).astype(np.float64).reshape(-1, 2)
for c in range(2):
    y = lfilter(bb, aa, x[:, c])
    e = (y * y).reshape(100, 4800).sum(axis=1)
    for mode in (0, 1, 2):
        d = res[mode].reshape(ns, 100, 2)[0, :, c]
        print("stream 0 ch", c, "mode", mode, "max rel vs scipy f64:", float(np.max(np.abs(d - e) / e)))
    print("stream 0 ch", c, "one-segment max rel vs scipy f64:", float(np.max(np.abs(ref.reshape(ns, 100, 2)[0, :, c] - e) / e)))
PY
cat $out/modes2.log
}

step_h() {
out=gpurun_out/r5h; mkdir -p $out
SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so python tools/probe_td_wps.py 10 > $out/wps.log 2> $out/wps.err
cat $out/wps.log; grep -E '##|k_time_domain<' $out/wps.err | uniq
}

step_i() {
out=gpurun_out/r5i; mkdir -p $out
python -m pytest tests -m gpu -q -x -k "parity or known or golden or session or tick or capture or independent or ragged or lufs or loud or gate or hist or corpus" > $out/tests.log 2>&1; tail -3 $out/tests.log
for rep in 1 2; do
for lib in head default; do
  echo "=== $lib (rep $rep)"
  if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
  python tools/probe_single_file.py 2>&1 | grep -v "^ *$"
  python tools/probe_getters.py
  python tools/probe_cfg5.py 64 2>&1 | grep -E "k_finalize|k_time|sum|wall"
  python tools/perf_probe.py 1024 10 | grep -E "k_finalize|sum"
done
done > $out/ab.log 2>&1
unset SOUNDSCOPE_HIP_LIB
cat $out/ab.log
}

step_$step
