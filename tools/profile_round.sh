#!/bin/bash
# Refresh the judged measurement set on the GPU box: bench line, rocprofv3 kernel-trace stats of the
# same bench command, and HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs).
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python $root/bench.py --steps 10 --warmup 2 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $root/bench.py --steps 10 --warmup 2 > $out/kt_bench.json 2> $out/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/pmc_$c -o p -- python $root/tools/perf_probe.py 1024 2 > $out/pmc_$c.log 2>&1
done
for d in kt pmc_FETCH_SIZE pmc_WRITE_SIZE; do
  db=$(find $out/$d -name '*.db' | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" $out/$d.txt
done
cat $out/bench.json
