#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r5open2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/open6.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import soundscope_amd as ssa
from conftest import make_stereo
x = make_stereo(1, 48000 * 12, 48000)
for i in range(4):
    t0 = time.perf_counter(); s = ssa.FileSession(x, 2, 48000); t1 = time.perf_counter(); s.close()
    print(f"open {1e3 * (t1 - t0):.2f} ms", flush=True)
import ctypes
time.sleep(0.05)
t0 = time.perf_counter(); s = ssa.FileSession(x, 2, 48000); t1 = time.perf_counter()
print(f"LAST open {1e3 * (t1 - t0):.2f} ms", flush=True)
PY
rocprofv3 --hip-trace --kernel-trace -f csv -d $out/ht -o ht -- python /tmp/open6.py > $out/open.log 2>&1
grep -v rocprofv3 $out/open.log | grep open
f=$(find $out/ht -name '*hip_api_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last open = calls after the largest idle gap near the end (the 50 ms sleep)
gaps = [(int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(len(rows)-1)]
g, i = max(gaps[len(gaps)//2:])
last = rows[i+1:]
t0 = int(last[0]["Start_Timestamp"])
print("calls in the last open:", len(last), "span us", (int(last[-1]["End_Timestamp"]) - t0)/1e3)
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3
    if d >= 8: print(f"  +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  {d:8.1f} us  {r['Function']}")
PY
rm -rf $out/ht
