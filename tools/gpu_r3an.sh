#!/bin/bash
# round 3, GPU call AN: k_finalize threads per stream at the bench shape (tuning build, SS_FIN_THREADS)
O=gpurun_out/r3an; mkdir -p $O
export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so
for i in 1 2; do for t in 64 128 256 512; do
  echo "threads $t: $(SS_FIN_THREADS=$t python tools/perf_probe.py 1024 10 2>&1 | grep -E 'finalize')" >> $O/ab.log
done; done
cat $O/ab.log
