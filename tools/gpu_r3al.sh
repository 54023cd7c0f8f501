#!/bin/bash
# round 3, GPU call AL: chunk lengths whose stride makes the pass reads two deep instead of four (48 kHz stereo: L = 25 -> 6 whole tiles of 32 x 25)
O=gpurun_out/r3al; mkdir -p $O
export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so
for i in 1 2; do python tools/sweep_td_chunk.py 48000 2 1024 auto 25 15 20 30 >> $O/L.log 2>&1; done
python tools/sweep_td_chunk.py 96000 2 512 auto 25 30 50 >> $O/L96.log 2>&1
cat $O/L.log $O/L96.log
