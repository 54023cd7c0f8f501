#!/bin/bash
# Where a session open spends its host time: rocprofv3 --hip-trace --stats over six opens of a 12 s file (HIP API totals).
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-open_hipapi}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/open6.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import soundscope_amd as ssa
from conftest import make_stereo
x = make_stereo(1, 48000 * 12, 48000)
for i in range(6):
    t0 = time.perf_counter(); s = ssa.FileSession(x, 2, 48000); t1 = time.perf_counter(); s.close()
    print(f"open {1e3 * (t1 - t0):.2f} ms", flush=True)
PY
rocprofv3 --hip-trace --stats -f csv -d $out/ht -o ht -- python /tmp/open6.py > $out/open.log 2>&1
f=$(find $out/ht -name '*hip_api_stats.csv' | head -1)
grep -v rocprofv3 $out/open.log | tail -7
[ -n "$f" ] && head -25 "$f" | cut -d, -f1-8
rm -rf $out/ht
