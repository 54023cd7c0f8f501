#!/bin/bash
# per-phase clock profile of k_fft4096_ms1 for several -DSS_FFT_PROF builds, back to back: tools/probe_fft_phases_ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "=== $lib"
  SOUNDSCOPE_HIP_LIB=$(realpath "$lib") python tools/probe_fft_phases.py 1024
done
