#!/bin/bash
# round 3, GPU call AD: bench shape with three segments per stream (3072 waves -> the spill-free three-wave build) against four
O=gpurun_out/r3ad; mkdir -p $O
for i in 1 2 3; do
  bash tools/sweep_td_seg.sh 48000 2 1024 auto 25 34 >> $O/seg.log 2>&1
done
bash tools/sweep_td_seg.sh 96000 8 64 auto 3 4 5 >> $O/seg5.log 2>&1
bash tools/sweep_td_seg.sh 48000 2 768 auto 25 34 >> $O/seg768.log 2>&1
cat $O/seg.log $O/seg5.log $O/seg768.log
