#!/bin/bash
# round 3, GPU call L: symmetric four-output epilogue of k_fft16k_run: correctness, then A/B against the Horner epilogue and round 2
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_gpu_dynamic_range.py tests/test_gpu_bench_shapes.py tests/test_gpu_independent.py -q -x -k "16384 or config5 or 16k or native or near_silent or scaled or numpy" > $O/tests16k.log 2>&1; echo "rc $?" >> $O/tests16k.log
for i in 1 2; do
  echo "=== r2tree" >> $O/ab.log; python tools/bin/r2tree/tools/probe_cfg5.py >> $O/ab.log 2>&1; python tools/bin/r2tree/tools/probe_native16k.py >> $O/ab.log 2>&1
  echo "=== nosym" >> $O/ab.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/nosym.so python tools/probe_cfg5.py >> $O/ab.log 2>&1; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/nosym.so python tools/probe_native16k.py >> $O/ab.log 2>&1
  echo "=== default" >> $O/ab.log; python tools/probe_cfg5.py >> $O/ab.log 2>&1; python tools/probe_native16k.py >> $O/ab.log 2>&1
done
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
tail -5 $O/tests16k.log; grep -E "===|tp=4x k_fft16k|stereo N" $O/ab.log; grep -E "passed|failed|FAILED" $O/gpu_all.log
