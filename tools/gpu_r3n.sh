#!/bin/bash
# round 3, GPU call N: batched remainder of the K-weighting passes (chunk lengths that are no multiple of ten: 44.1 / 88.2 kHz)
O=gpurun_out/r3n; mkdir -p $O
for i in 1 2; do
  echo "=== prev" >> $O/rates.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/prev.so python tools/probe_rates.py >> $O/rates.log 2>&1
  echo "=== default" >> $O/rates.log; python tools/probe_rates.py >> $O/rates.log 2>&1
done
python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_gpu_independent.py tests/test_gpu_session.py -m gpu -q -x > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
cat $O/rates.log; tail -5 $O/tests.log
