#!/bin/bash
# PMC passes (one rocprofv3 run per counter group; no tracing) over tools/perf_probe.py.
# usage: [PROBE=tools/probe_native.py] tools/pmc_passes.sh <tag> "<probe args>" "<group1 counters>" "<group2 counters>" ...
set -u
tag=$1; shift
args=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $out/g$i -o p -- python $root/${PROBE:-tools/perf_probe.py} $args > $out/g$i.log 2>&1
  db=$(find $out/g$i -name '*.db' | head -1)
  if [ -n "$db" ]; then echo "## group $i: $grp" >> $out/summary.txt; python $root/tools/rocpd_summary.py "$db" | grep -E "ssk::" | grep -v "^ *[0-9]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +[0-9.]+ +None" >> $out/summary.txt; fi
  rm -rf $out/g$i
done
cat $out/summary.txt
