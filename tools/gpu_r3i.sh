#!/bin/bash
# round 3, GPU call I: bulk-issued epilogue loads / hoisted waits (both spectrum kernels) against the round-2 tree on one box; suite
O=gpurun_out/r3i; mkdir -p $O
for i in 1 2 3; do
  echo "=== r2tree" >> $O/ab.log; python tools/bin/r2tree/tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; python tools/bin/r2tree/tools/probe_cfg5.py >> $O/ab.log 2>&1
  echo "=== default" >> $O/ab.log; python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; python tools/probe_cfg5.py >> $O/ab.log 2>&1
done
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
grep -E "===|k_fft4096|k_time_domain  |tp=4x k_fft16k|tp=4x k_time" $O/ab.log | paste - - - - - ; grep -E "passed|failed|FAILED" $O/gpu_all.log
