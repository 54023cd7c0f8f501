#!/bin/bash
# round 3, GPU call C: skewed k_fft16k_run A/B, zero-row + LDS-table ms1 A/B against HEAD, dynamic-range tests, dual-mono probe
O=gpurun_out/r3c; mkdir -p $O
for i in 1 2; do
for lib in default noskew; do
  echo "=== $lib" >> $O/cfg5_ab.log
  if [ $lib = default ]; then python tools/probe_cfg5.py >> $O/cfg5_ab.log 2>&1; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/probe_cfg5.py >> $O/cfg5_ab.log 2>&1; fi
done
for lib in default head; do
  echo "=== $lib" >> $O/cfg3_ab.log
  if [ $lib = default ]; then python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; fi
done
done
python tools/probe_mono.py > $O/mono.log 2>&1
python -m pytest tests/test_gpu_dynamic_range.py -x -q -s > $O/dynrange.log 2>&1; echo "rc $?" >> $O/dynrange.log
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
grep -E "fft16k_run|total|===" $O/cfg5_ab.log | head -40; grep -E "===|k_fft4096|time_domain|sum" $O/cfg3_ab.log; tail -5 $O/dynrange.log; tail -5 $O/gpu_all.log; cat $O/mono.log
