#!/bin/bash
# round 3, GPU call AH: k_fft16k_run epilogue with the retained bins dealt to the waves in equal contiguous shares
O=gpurun_out/r3ah; mkdir -p $O
for i in 1 2 3; do for lib in head3 default; do
  if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
  echo "$lib $(python tools/probe_cfg5.py 64 2>&1 | grep -E 'tp=4x k_fft16k') | $(python tools/probe_native16k.py 2>&1 | cut -c1-70)" >> $O/ab.log
done; done
unset SOUNDSCOPE_HIP_LIB
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/ab.log; tail -5 $O/gpu_all.log
