#!/bin/bash
# round 3, GPU call H: k_fft4096_ms1 of this round against the round-2 tree on ONE box, with the two additions switched off one at a time
O=gpurun_out/r3h; mkdir -p $O
for i in 1 2 3 4; do
  echo "=== r2tree" >> $O/ab.log; python tools/bin/r2tree/tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1
  for lib in default nozrow noldstab neither; do
    echo "=== $lib" >> $O/ab.log
    if [ $lib = default ]; then python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; else SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/$lib.so python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1; fi
  done
done
echo "=== r2tree cfg5" >> $O/ab5.log; python tools/bin/r2tree/tools/probe_cfg5.py >> $O/ab5.log 2>&1
echo "=== default cfg5" >> $O/ab5.log; python tools/probe_cfg5.py >> $O/ab5.log 2>&1
echo "=== r2tree cfg5" >> $O/ab5.log; python tools/bin/r2tree/tools/probe_cfg5.py >> $O/ab5.log 2>&1
echo "=== default cfg5" >> $O/ab5.log; python tools/probe_cfg5.py >> $O/ab5.log 2>&1
grep -E "===|k_fft4096|k_time_domain" $O/ab.log | paste - - - | sort; grep -E "===|tp=4x k_fft16k|tp=4x k_time" $O/ab5.log | paste - - -
