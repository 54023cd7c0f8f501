#!/bin/bash
# round 3, GPU call O: chunk-length sweep of k_time_domain at 44.1 / 88.2 kHz (tuning build) with the batched remainder
O=gpurun_out/r3o; mkdir -p $O
export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tune.so
python tools/sweep_td_chunk.py 44100 2 1024 auto 20 21 25 30 35 45 49 50 63 70 > $O/sweep_44k.log 2>&1
python tools/sweep_td_chunk.py 88200 2 1024 auto 21 30 35 45 49 63 > $O/sweep_88k.log 2>&1
python tools/sweep_td_chunk.py 48000 2 1024 auto 30 40 50 60 > $O/sweep_48k.log 2>&1
python tools/sweep_td_chunk.py 44100 1 2048 auto 21 30 35 45 49 63 > $O/sweep_44k_mono.log 2>&1
cat $O/sweep_44k.log $O/sweep_88k.log $O/sweep_48k.log $O/sweep_44k_mono.log
