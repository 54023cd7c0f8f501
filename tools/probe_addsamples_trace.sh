#!/bin/bash
# kernel durations of the Analyzer path's add_samples + get_shortterm_lufs (tick-sized slices through page-locked buffers)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r6g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ad.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
x = make_stereo(1, 48000 * 8, 48000)
an = ssa.Analyzer(); an.create_loudness_meter(2, 48000)
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(300):
        an.add_samples(x[k * 2048:k * 2048 + 16384]); an.get_shortterm_lufs()
    print("add_samples + get_shortterm wall", (time.perf_counter() - t0) / 300 * 1e6, "us")
PY
rocprofv3 --kernel-trace --stats -f csv -d $out/ht -o ht -- python /tmp/ad.py > $out/run.log 2>&1
grep "wall" $out/run.log
f=$(find $out/ht -name '*kernel_stats.csv' | head -1)
python - "$f" <<PY
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:6]: print(r[0][:60].ljust(60), *r[1:5])
PY
rm -rf $out/ht
python /tmp/ad.py
