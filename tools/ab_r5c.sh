#!/bin/bash
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
