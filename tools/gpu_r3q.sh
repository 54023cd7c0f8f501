#!/bin/bash
# round 3, GPU call Q: k_finalize for long streams (up to 1024 threads; window sums with their loads issued together)
O=gpurun_out/r3q2; mkdir -p $O
for i in 1 2; do
  echo "=== prev" >> $O/long.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/prev.so python tools/probe_long_stream.py >> $O/long.log 2>&1
  echo "=== default" >> $O/long.log; python tools/probe_long_stream.py >> $O/long.log 2>&1
done
for i in 1 2; do
  echo "=== prev" >> $O/tick.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/prev.so python tools/tick_latency.py >> $O/tick.log 2>&1
  echo "=== default" >> $O/tick.log; python tools/tick_latency.py >> $O/tick.log 2>&1
done
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/long.log $O/tick.log; tail -3 $O/gpu_all.log
