#!/bin/bash
# round 3, GPU call Y: per-phase wave priorities in the FFT kernels (two bits per compute phase; exchange phases stay at 3 in the
# N = 4096 kernels; k_fft16k_run: bits 9:8 = exchange level)
O=gpurun_out/r3y; mkdir -p $O
for i in 1 2; do
  for v in default fp_36_768 fp_64_804 fp_21_832 fp_6_789 fp_149_106 fp_106_874 fp_16_447; do
    if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
    echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep -E 'k_fft4096') | $(python tools/probe_cfg5.py 64 2>&1 | grep -E 'tp=4x k_fft16k') | $(python tools/probe_channels.py 2>&1 | grep -E 'channels 1 ' | cut -c1-60)" >> $O/ab.log
  done
done
cat $O/ab.log
