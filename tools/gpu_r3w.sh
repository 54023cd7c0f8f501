#!/bin/bash
# round 3, GPU call W: k_time_domain graded wave priorities, third batch (2 bits per phase)
O=gpurun_out/r3w; mkdir -p $O
for i in 1 2 3; do
  for v in default tdpmap1516 tdpmap1532 tdpmap2540 tdpmap1772 tdpmap17901 tdpmap5612 tdpmap1452 tdpmap1464 tdpmap1260 tdpmap492 tdpmap34286; do
    if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
    echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep -E 'time_domain')" >> $O/ab.log
  done
done
cat $O/ab.log
