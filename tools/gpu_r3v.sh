#!/bin/bash
# round 3, GPU call V: k_time_domain wave priorities, second batch (combinations, levels, graded maps)
O=gpurun_out/r3v; mkdir -p $O
for i in 1 2 3; do
  for v in default tdprio10 tdprio10l1 tdprio10l3 tdprio58 tdprio26 tdprio14 tdprio62 tdprio74 tdpmap1516 tdpmap3976 tdpmap1756; do
    echo "=== $v" >> $O/ab.log
    if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
    python tools/perf_probe.py 1024 10 2>&1 | grep -E "time_domain" >> $O/ab.log
  done
done
cat $O/ab.log
