#!/bin/bash
# round 3, GPU call AB: k_time_domain register-allocated for three waves per SIMD (no spills) under the wave priorities
O=gpurun_out/r3ab; mkdir -p $O
for i in 1 2 3; do for v in default tdw3; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep time_domain) | $(python tools/probe_cfg5.py 64 2>&1 | grep 'tp=4x k_time_domain')" >> $O/ab.log
done; done
cat $O/ab.log
