#!/bin/bash
# round 3, GPU call B: K = 32 interference microbenchmark, whole GPU suite with the new tests, default bench line
O=gpurun_out/r3b; mkdir -p $O
timeout 300 tools/bin/ubench_k32_interference 20000 3 > $O/k32_interference.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -n 3 $O/gpu_all.log; tail -n 5 $O/k32_interference.txt
