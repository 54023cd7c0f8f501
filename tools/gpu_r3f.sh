#!/bin/bash
# round 3, GPU call F: whole suite on the pruned kernels, the io microbenchmark variants, the judged profile set
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
tools/bin/ubench_fftio > $O/ubench_fftio.txt 2>&1
python tools/probe_cfg5.py > $O/cfg5.log 2>&1
bash tools/profile_round.sh r3f/prof > $O/profile_round.log 2>&1
grep -E "passed|failed|FAILED" $O/gpu_all.log; cat $O/ubench_fftio.txt; grep -E "fft16k_run|time_domain" $O/cfg5.log; head -c 1500 $O/prof/bench.json; echo; head -30 $O/prof/kernel_stats.txt
