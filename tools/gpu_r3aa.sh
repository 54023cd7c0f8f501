#!/bin/bash
# round 3, GPU call AA: the conflict-free-passes bound of k_time_domain re-measured under the graded wave priorities (timing only)
O=gpurun_out/r3aa; mkdir -p $O
for i in 1 2 3; do for v in default tdnc; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep time_domain)" >> $O/ab.log
done; done
cat $O/ab.log
