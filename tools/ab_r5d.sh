#!/bin/bash
out=gpurun_out/r5d; mkdir -p $out
python -m pytest tests/test_gpu_columns.py tests/test_gpu_bench_shapes.py tests/test_gpu_independent.py tests/test_gpu_reference_suite.py -m gpu -q -x > $out/tests1.log 2>&1; tail -3 $out/tests1.log
for rep in 1 2; do echo "=== columns only, default (rep $rep)"; python tools/perf_probe.py 1024 20 --cols=160 | grep -E "k_fft|k_time"; done
echo "=== full rows"; python tools/perf_probe.py 1024 20 | grep -E "k_fft|k_time"
