#!/bin/bash
# A/B of library builds on BASELINE config 5 (tools/probe_cfg5.py): tools/ab_cfg5.sh lib1.so lib2.so ...  ("default" = in-tree)
for lib in "$@"; do
  echo "=== $lib"
  if [ "$lib" = "default" ]; then python tools/probe_cfg5.py 64; else SOUNDSCOPE_HIP_LIB=$(realpath "$lib") python tools/probe_cfg5.py 64; fi
done
