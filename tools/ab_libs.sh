#!/bin/bash
# A/B: run tools/perf_probe.py against alternative builds of the library (same box, back to back)
# usage: tools/ab_libs.sh [--args "1024 5 --check"] lib1.so lib2.so ...   ("default" = the in-tree build)
args="1024 5"
if [ "$1" = "--args" ]; then args="$2"; shift 2; fi
for lib in "$@"; do
  echo "=== $lib"
  if [ "$lib" = "default" ]; then python tools/perf_probe.py $args; else SOUNDSCOPE_HIP_LIB=$(realpath "$lib") python tools/perf_probe.py $args; fi
done
