#!/bin/bash
# round 3, GPU call AK: k_time_domain pass reads 4 deep (the kernel) / exactly 2 deep / conflict-free (timing-only probes)
O=gpurun_out/r3ak; mkdir -p $O
for i in 1 2 3; do for v in default td2way tdnc; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep time_domain)" >> $O/ab.log
done; done
cat $O/ab.log
