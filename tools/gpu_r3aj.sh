#!/bin/bash
# round 3, GPU call AJ: the whole -m gpu suite three times back to back on the final tree (flakiness check), smoke
O=gpurun_out/r3aj; mkdir -p $O
for i in 1 2 3; do
  python -m pytest tests -m gpu -q > $O/gpu_all_$i.log 2>&1; echo "suite $i rc $?" >> $O/summary.txt; grep -E "passed|failed" $O/gpu_all_$i.log >> $O/summary.txt
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/summary.txt 2>&1
cat $O/summary.txt
