#!/bin/bash
# round 3, GPU call AM: whole-library compiler scheduling flags (-mllvm -enable-post-misched=0; -mllvm -amdgpu-schedule-relaxed-occupancy=1)
O=gpurun_out/r3am; mkdir -p $O
for i in 1 2; do for v in default postmisched0 relaxocc; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep -E 'k_fft4096|time_domain' | tr '\n' ' ') | $(python tools/probe_cfg5.py 64 2>&1 | grep -E 'tp=4x k_fft16k')" >> $O/ab.log
done; done
cat $O/ab.log
