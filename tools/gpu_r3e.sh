#!/bin/bash
# round 3, GPU call E: columns-only mode tests + timing, lighter zero-row detection A/B against HEAD~ build, whole suite
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests/test_gpu_columns.py tests/test_gpu_dynamic_range.py -q -x > $O/new_tests.log 2>&1; echo "rc $?" >> $O/new_tests.log
for i in 1 2 3; do
for lib in default head; do
  echo "=== $lib" >> $O/cfg3_ab.log
  if [ $lib = default ]; then python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; fi
done
done
python tools/probe_columns.py > $O/columns.log 2>&1
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
tail -4 $O/new_tests.log; grep -E "===|k_fft4096|time_domain" $O/cfg3_ab.log | paste - - - ; cat $O/columns.log; grep -E "passed|failed|FAILED" $O/gpu_all.log
