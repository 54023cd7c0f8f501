#!/bin/bash
# round 3, GPU call G: second-pass twiddle table [kb][tb] with batched reads (SS_FFT_TW2K) A/B, then the suite
O=gpurun_out/r3g; mkdir -p $O
for i in 1 2 3; do
for lib in default tw2k0; do
  echo "=== $lib" >> $O/ab.log
  if [ $lib = default ]; then python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; python tools/probe_cfg5.py >> $O/ab.log 2>&1
  else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/probe_cfg5.py >> $O/ab.log 2>&1; fi
done
done
python tools/probe_columns.py > $O/columns.log 2>&1
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
grep -E "===|k_fft4096|tp=4x k_fft16k|tp=4x k_time" $O/ab.log | paste - - - - ; cat $O/columns.log; grep -E "passed|failed|FAILED" $O/gpu_all.log
