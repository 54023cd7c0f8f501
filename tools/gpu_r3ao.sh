#!/bin/bash
# round 3, GPU call AO: suite + bench probe after k_finalize went to four waves per stream
O=gpurun_out/r3ao; mkdir -p $O
python tools/perf_probe.py 1024 10 > $O/probe.log 2>&1
python tools/probe_long_stream.py >> $O/probe.log 2>&1
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/probe.log; tail -3 $O/gpu_all.log
