#!/bin/bash
# round 3, GPU call Z: k_time_domain parameters re-swept under the graded wave priorities (chunk length, segment length, read batch)
O=gpurun_out/r3z; mkdir -p $O
export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so
python tools/sweep_td_chunk.py 48000 2 1024 auto 30 33 40 50 60 > $O/L_48k.log 2>&1
python tools/sweep_td_chunk.py 96000 8 64 auto 20 30 40 50 60 > $O/L_cfg5.log 2>&1
python tools/sweep_td_chunk.py 44100 2 1024 auto 30 49 63 > $O/L_44k.log 2>&1
bash tools/sweep_td_seg.sh 48000 2 1024 auto 13 20 25 34 50 > $O/seg_48k.log 2>&1
bash tools/sweep_td_seg.sh 96000 8 64 auto 2 3 4 5 7 > $O/seg_cfg5.log 2>&1
unset SOUNDSCOPE_HIP_LIB
for i in 1 2; do for v in default tdb15 tdb6; do
  if [ $v = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$v.so; fi
  echo "$v $(python tools/perf_probe.py 1024 10 2>&1 | grep time_domain)" >> $O/batch.log
done; done
cat $O/L_48k.log $O/L_cfg5.log $O/L_44k.log $O/seg_48k.log $O/seg_cfg5.log $O/batch.log
