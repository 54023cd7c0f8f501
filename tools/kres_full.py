#!/usr/bin/env python3
"""Full-name kernel resource table from a saved -Rpass-analysis=kernel-resource-usage log: tools/kres_full.py <log> [name filter]"""
import re, subprocess, sys
cur = None; rows = []
for line in open(sys.argv[1]):
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    body = re.sub(r"^[^ ]*:\d+:\d+: +", "", m.group(1))
    k, _, v = body.partition(":"); k = k.strip(); v = v.strip()
    if k == "Function Name": cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r, n in zip(rows, names):
    n = n.split("(")[0]
    if flt in n:
        print(n, "VGPR", r.get("VGPRs"), "scratch", r.get("ScratchSize [bytes/lane]"), "occ", r.get("Occupancy [waves/SIMD]"), "LDS", r.get("LDS Size [bytes/block]"))
