#!/bin/bash
# L2 counters of k_fft16k_run at config 5 (where does its 1.33x input over-fetch come from?): separate --pmc passes over tools/probe_cfg5.py
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-pmc_c5_l2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/summary.txt
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum"; do
  timeout 300 rocprofv3 --pmc $grp -d $out/p -o p -- python $root/tools/probe_cfg5.py 64 > $out/log.txt 2>&1
  db=$(find $out/p -name '*.db' | head -1)
  echo "## rocprofv3 --pmc $grp -- python tools/probe_cfg5.py 64" >> $out/summary.txt
  [ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" | grep "k_fft16k_run" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" | cut -c1-130 >> $out/summary.txt
  rm -rf $out/p
done
cat $out/summary.txt
