#!/bin/bash
# Kernel durations inside the reference's tick (ss_session_tick_file at the reference cadence): rocprofv3 kernel trace over
# bench.reference_tick_workload alone.   usage: tools/probe_tick_trace.sh [tag under gpurun_out/]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-tick_trace}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tick_only.py <<PY
import sys
sys.path.insert(0, "$root")
import bench, soundscope_amd as ssa
from soundscope_amd import _lib as L
r = bench.reference_tick_workload(ssa, L)
print("tick median %.1f us under the tracer" % r["gpu_tick_us"]["median"])
PY
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python /tmp/tick_only.py > $out/tick.log 2>&1
db=$(find $out/kt -name '*.db' | head -1)
[ -n "$db" ] && python $root/tools/rocpd_summary.py "$db" $out/tick_kernel_stats.txt
rm -rf $out/kt
tail -2 $out/tick.log; cat $out/tick_kernel_stats.txt | cut -c1-200
