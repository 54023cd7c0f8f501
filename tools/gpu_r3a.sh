#!/bin/bash
# round 3, GPU call A: overlap stress test, K = 32 race reproduction, baseline timings, whole GPU suite
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests/test_gpu_overlap_stress.py -x -q > $O/stress.log 2>&1; echo "stress rc $?" >> $O/stress.log
for lib in k32 k32_noprio; do SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so timeout 300 python tools/probe_overlap_race.py 1024 30 > $O/race_$lib.log 2>&1; done
SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/k32.so timeout 300 python tools/probe_overlap_race.py 1024 20 --split > $O/race_k32_split.log 2>&1
timeout 300 python tools/probe_overlap_race.py 1024 30 > $O/race_prod.log 2>&1
python tools/perf_probe.py 1024 10 --check > $O/perf.log 2>&1
python tools/probe_cfg5.py > $O/cfg5.log 2>&1
python -m pytest tests -m gpu -x -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
tail -3 $O/stress.log $O/race_k32.log $O/gpu_all.log
