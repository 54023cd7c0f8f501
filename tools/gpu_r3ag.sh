#!/bin/bash
# round 3, GPU call AG: bench step in the three launch modes under the new kernels (sequential / overlap 1 / overlap 2)
O=gpurun_out/r3ag; mkdir -p $O
for i in 1 2; do for m in 0 1 2; do
  python bench.py --steps 100 --warmup 10 --overlap $m --no-cpu --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap $m', d['ms_per_step'], d['value']/1e9, d['config']['mode'][:40])" >> $O/modes.log
done; done
cat $O/modes.log
