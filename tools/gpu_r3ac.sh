#!/bin/bash
# round 3, GPU call AC: spill-free three-waves-per-SIMD build of k_time_domain for grids that fit at that occupancy
# (head2 = the build before it, with the wave priorities)
O=gpurun_out/r3ac; mkdir -p $O
for i in 1 2; do for lib in head2 default; do
  echo "=== $lib" >> $O/ab.log
  if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
  python tools/perf_probe.py 1024 10 2>&1 | grep -E "time_domain" >> $O/ab.log
  python tools/probe_cfg5.py 64 2>&1 | grep -E "time_domain" >> $O/ab.log
  python tools/probe_long_stream.py >> $O/ab.log 2>&1
  python tools/probe_native16k.py >> $O/ab.log 2>&1
  python tools/tick_latency.py 2>&1 | grep -E "add_samples|session tick|whole tick" >> $O/ab.log
done; done
unset SOUNDSCOPE_HIP_LIB
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/ab.log; tail -3 $O/gpu_all.log
