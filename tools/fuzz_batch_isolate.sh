#!/bin/bash
# tools/fuzz_batch.py in chunks of 100 seeds, one process each (a memory fault ends a chunk, not the series):  tools/fuzz_batch_isolate.sh [chunks] [first seed] [--big]
chunks=${1:-6}; first=${2:-1}; extra=${3:-}
mkdir -p gpurun_out/r5fz
for c in $(seq 0 $((chunks - 1))); do
  s=$((first + 100 * c))
  timeout 600 python tools/fuzz_batch.py 100 $s $extra > gpurun_out/r5fz/batch_$s.log 2>&1; echo "seeds $s.. rc $? : $(grep -v '^GPU\|^Failed\|coredump' gpurun_out/r5fz/batch_$s.log | tail -1 | cut -c1-200)"
  grep "^FAIL" gpurun_out/r5fz/batch_$s.log | cut -c1-700 | head -8
done
