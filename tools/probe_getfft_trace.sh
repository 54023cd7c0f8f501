root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r6e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gf.py <<PY
import sys, time
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import numpy as np
import soundscope_amd as ssa
from conftest import make_stereo
x = make_stereo(1, 48000 * 4, 48000)
mid, side = ssa.get_mid_and_side_samples(x)
an = ssa.Analyzer(); an.create_loudness_meter(2, 48000)
for k in range(300): an.get_fft(mid[k * 100:k * 100 + 16384])
t0 = time.perf_counter()
for k in range(300): an.get_fft(mid[k * 100:k * 100 + 16384])
print("get_fft wall", (time.perf_counter() - t0) / 300 * 1e6, "us")
PY
rocprofv3 --hip-trace --kernel-trace --stats -f csv -d $out/ht -o ht -- python /tmp/gf.py > $out/run.log 2>&1
grep "get_fft wall" $out/run.log
f=$(find $out/ht -name '*hip_api_stats.csv' | head -1); head -8 "$f" | cut -d, -f1-7
f=$(find $out/ht -name '*kernel_stats.csv' | head -1); head -4 "$f" | cut -d, -f1-7
rm -rf $out/ht
python /tmp/gf.py
