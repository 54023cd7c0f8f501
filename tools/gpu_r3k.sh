#!/bin/bash
# round 3, GPU call K: pair kernel at four waves per SIMD; native 16k stereo and other channel counts against the round-2 tree
O=gpurun_out/r3k; mkdir -p $O
for i in 1 2; do
  echo "=== r2tree" >> $O/ab.log; python tools/bin/r2tree/tools/probe_channels.py >> $O/ab.log 2>&1; python tools/bin/r2tree/tools/probe_native16k.py >> $O/ab.log 2>&1
  echo "=== default" >> $O/ab.log; python tools/probe_channels.py >> $O/ab.log 2>&1; python tools/probe_native16k.py >> $O/ab.log 2>&1
  echo "=== pairw4" >> $O/ab.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/pairw4.so python tools/probe_channels.py >> $O/ab.log 2>&1
done
cat $O/ab.log
