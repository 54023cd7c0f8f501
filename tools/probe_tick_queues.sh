#!/bin/bash
# Which hardware queue does a session's loudness chain run on, and what does a tick look like on a slow one?
# (rocprofv3 kernel trace over tools/probe_tick_sessions.py, timelines by tools/tick_timeline.py)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-tick_queues}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o kt -- python $root/tools/probe_tick_sessions.py > $out/sessions.log 2>&1
db=$(find $out/kt -name '*.db' | head -1)
[ -n "$db" ] && python $root/tools/tick_timeline.py "$db" 1 > $out/timeline.txt
rm -rf $out/kt
grep -v rocprofv3 $out/sessions.log; cat $out/timeline.txt
