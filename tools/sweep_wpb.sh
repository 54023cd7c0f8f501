#!/bin/bash
# windows per spectrum workgroup sweep on the config 3 probe (needs a -DSS_TUNING build: tools/bin/tune.so)
for w in "$@"; do
  echo "=== SS_FFT_WPB=$w"
  SS_FFT_WPB=$w SOUNDSCOPE_HIP_LIB=$(realpath tools/bin/tune.so) python tools/perf_probe.py 1024 8 | grep -E "fft4096"
done
