#!/bin/bash
# Host-side cost of the handle-less / one-shot entry points on a 12 s file, warm: wall times and rocprofv3 --hip-trace --stats totals.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-oneshot_hipapi}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/oneshot.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
x = make_stereo(1, 48000 * 12, 48000)
an = ssa.Analyzer(); an.create_loudness_meter(2, 48000)
for i in range(6):
    t0 = time.perf_counter(); w = ssa.Analyzer.get_waveform(x, 12.0); t1 = time.perf_counter()
    v = an.calculate_integrated_lufs(2, x); t2 = time.perf_counter()
    m, s = ssa.Analyzer.get_mid_and_side_samples(x) if hasattr(ssa.Analyzer, "get_mid_and_side_samples") else (None, None); t3 = time.perf_counter()
    print(f"get_waveform {1e3 * (t1 - t0):.2f} ms, calculate_integrated_lufs {1e3 * (t2 - t1):.2f} ms, mid/side {1e3 * (t3 - t2):.2f} ms", flush=True)
PY
rocprofv3 --hip-trace --stats -f csv -d $out/ht -o ht -- python /tmp/oneshot.py > $out/run.log 2>&1
f=$(find $out/ht -name '*hip_api_stats.csv' | head -1)
grep -v "rocprofv3\|^E2026\|^W2026" $out/run.log | tail -7
[ -n "$f" ] && head -16 "$f" | cut -d, -f1-7
rm -rf $out/ht
