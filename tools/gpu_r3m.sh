#!/bin/bash
# round 3, GPU call M: k_time_domain with bank-conflict-free pass reads (timing only), fresh phase profile and PMC passes
O=gpurun_out/r3m; mkdir -p $O
for i in 1 2 3; do
  echo "=== default" >> $O/ab.log; python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1
  echo "=== tdnc" >> $O/ab.log; SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tdnc.so python tools/perf_probe.py 1024 10 >> $O/ab.log 2>&1
done
SOUNDSCOPE_HIP_LIB=$PWD/tools/bin/tdprof.so python tools/probe_td_phases.py > $O/td_phases.log 2>&1
bash tools/pmc_td.sh r3m/pmc > /dev/null 2>&1
grep -E "===|time_domain" $O/ab.log; cat $O/td_phases.log; cat $O/pmc/summary.txt | head -60
