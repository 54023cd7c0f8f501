#!/bin/bash
# one rocprofv3 --pmc pass over the config 3 probe: tools/pmc_one.sh <tag> COUNTER...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc "$@" -d $out/p -o p -- python $root/tools/perf_probe.py 1024 2 > $out/log.txt 2>&1
db=$(find $out/p -name '*.db' | head -1)
[ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" | grep -E "ssk::k_(fft4096|time_domain)" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+" | cut -c1-120
rm -rf $out/p
