root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r5h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/cap.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
rate = 48000
x = make_stereo(1, rate * 12, rate)
cap = ssa.CaptureSession(2, rate)
ring = np.concatenate([x, x, x])[:30 * rate]
for tick in range(100): cap.analyze_microphone_input(ring)
PY
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/kt -o kt -- python /tmp/cap.py > $out/cap.log 2>&1
db=$(find $out/kt -name '*.db' | head -1)
python $root/tools/rocpd_summary.py "$db" | head -12 | cut -c1-60,150-230
python - <<PY
import sqlite3
c = sqlite3.connect("$db")
try:
    cols = [r[1] for r in c.execute("pragma table_info('memory_copies')")]
    print(cols)
    for r in c.execute("select name, count(*), avg(end-start)/1e3, avg(size) from memory_copies group by name, size order by 3 desc limit 8"): print(r)
except Exception as e: print("no memcpy table", e)
PY
rm -rf $out/kt
