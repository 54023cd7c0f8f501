#!/bin/bash
# kernel resource table of one .hip file (device-only compile, no GPU needed): tools/kres.sh ss_fft.hip [-DSS_...]
f=$1; shift
cd "$(dirname "$0")/../soundscope_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" --offload-device-only -c -o /dev/null "$f" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line or "warning" in line: sys.stderr.write(line)
        continue
    k, _, v = m.group(1).partition(":")
    k = k.strip(); v = v.strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    import subprocess
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-46s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-5s occ %-2s LDS %s" % (name[:46], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
