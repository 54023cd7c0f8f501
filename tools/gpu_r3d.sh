#!/bin/bash
# round 3, GPU call D: ms1 variants (LDS table / zero-row detection), time-domain b64 frame reads, dynamic-range tests
O=gpurun_out/r3d; mkdir -p $O
for i in 1 2 3; do
for lib in default head ms1_noldstab ms1_nozrow tdb32; do
  echo "=== $lib" >> $O/cfg3_ab.log
  if [ $lib = default ]; then python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 10 >> $O/cfg3_ab.log 2>&1; fi
done
done
python tools/probe_cfg5.py > $O/cfg5.log 2>&1
python -m pytest tests/test_gpu_dynamic_range.py -q -s > $O/dynrange.log 2>&1; echo "rc $?" >> $O/dynrange.log
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
grep -E "===|k_fft4096|time_domain" $O/cfg3_ab.log; grep fft16k_run $O/cfg5.log; tail -4 $O/dynrange.log; grep -E "passed|failed|FAILED" $O/gpu_all.log
