#!/bin/bash
# round 3, GPU call J: the judged profile set on the final kernels + suite + default bench line
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile_round.sh r3j/prof > $O/profile_round.log 2>&1
grep -E "passed|failed|FAILED" $O/gpu_all.log; tail -4 $O/bench_default.err; head -c 700 $O/bench_default.json; echo; head -14 $O/prof/kernel_stats.txt | cut -c1-210
