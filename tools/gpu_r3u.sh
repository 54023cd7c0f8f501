#!/bin/bash
# round 3, GPU call U: k_time_domain with raised wave priority during one group of phases (SS_TD_PRIO bit i = the phase behind
# phase mark i): 10 = the two K-weighting passes, 4 = scan, 48 = true-peak conversion + MFMA loop, 32 = MFMA loop, 129 = staging + decimation
O=gpurun_out/r3u; mkdir -p $O
for i in 1 2; do
  for v in default tdprio10 tdprio4 tdprio48 tdprio32 tdprio129; do
    echo "=== $v" >> $O/ab.log
    if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
    python tools/perf_probe.py 1024 10 2>&1 | grep -E "time_domain" >> $O/ab.log
  done
done
cat $O/ab.log
