#!/bin/bash
# A/B: run tools/perf_probe.py against alternative builds of the library (same process setup)
for lib in "$@"; do
  echo "=== $lib"
  cp soundscope_amd/lib/libsoundscope_hip.so /tmp/orig.so
  cp "$lib" soundscope_amd/lib/libsoundscope_hip.so
  python tools/perf_probe.py 1024 5 --check
  cp /tmp/orig.so soundscope_amd/lib/libsoundscope_hip.so
done
