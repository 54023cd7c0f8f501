#!/bin/bash
# round 3, GPU call X: graded wave priorities in k_time_domain (passes 3, scan 2, true peak 1, rest 0) against the build before
# them, across shapes; then the whole suite
O=gpurun_out/r3x; mkdir -p $O
for i in 1 2; do
  for lib in head default; do
    echo "=== $lib" >> $O/ab.log
    if [ $lib = default ]; then unset SOUNDSCOPE_HIP_LIB; else export SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so; fi
    python tools/perf_probe.py 1024 10 2>&1 | grep -E "time_domain|sum" >> $O/ab.log
    python tools/probe_cfg5.py 64 2>&1 | grep -E "time_domain|total" >> $O/ab.log
    python tools/probe_channels.py >> $O/ab.log 2>&1
    python tools/probe_rates.py >> $O/ab.log 2>&1
    python tools/probe_long_stream.py >> $O/ab.log 2>&1
  done
done
unset SOUNDSCOPE_HIP_LIB
python tools/tick_latency.py > $O/tick.log 2>&1
python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "suite rc $?" >> $O/gpu_all.log
cat $O/ab.log; cat $O/tick.log; tail -3 $O/gpu_all.log
