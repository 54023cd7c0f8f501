#!/bin/bash
# round 3, GPU call AI: k_time_domain prefetch / staging with base + immediate addressing for whole rounds
O=gpurun_out/r3ai; mkdir -p $O
for i in 1 2 3; do for lib in head4 default; do
  if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
  echo "$lib $(python tools/perf_probe.py 1024 10 2>&1 | grep -E 'time_domain') | $(python tools/probe_cfg5.py 64 2>&1 | grep -E 'tp=4x k_time_domain') | $(python tools/probe_rates.py 2>&1 | grep -E '44100' | cut -c58-82)" >> $O/ab.log
done; done
unset SOUNDSCOPE_HIP_LIB
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/ab.log; tail -4 $O/gpu_all.log
