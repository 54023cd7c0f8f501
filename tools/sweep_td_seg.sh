#!/bin/bash
# segment-length sweep of k_time_domain (sub-blocks per segment; needs a -DSS_TUNING build): tools/sweep_td_seg.sh <rate> <ch> <streams> seg...
rate=$1; ch=$2; streams=$3; shift 3
for s in "$@"; do
  if [ "$s" = "auto" ]; then SOUNDSCOPE_HIP_LIB=$(realpath tools/bin/tune.so) python tools/sweep_td_chunk.py $rate $ch $streams auto | sed "s/^/seg=auto /"
  else SS_TD_SEG_SUB=$s SOUNDSCOPE_HIP_LIB=$(realpath tools/bin/tune.so) python tools/sweep_td_chunk.py $rate $ch $streams auto | sed "s/^/seg=$s /"; fi
done
