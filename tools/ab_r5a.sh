#!/bin/bash
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
