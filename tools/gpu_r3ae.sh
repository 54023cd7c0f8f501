#!/bin/bash
# round 3, GPU call AE: STATIC wave priorities by hardware wave slot: k_time_domain (slot & 3, no phase priorities);
# k_fft4096_ms1 / pairw (2 - min(slot, 2) outside the exchanges, 3 inside)
O=gpurun_out/r3ae; mkdir -p $O
for i in 1 2 3; do for v in default tdstatic fftstatic; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep -E 'k_fft4096|time_domain' | tr '\n' ' ')" >> $O/ab.log
done; done
cat $O/ab.log
