#!/usr/bin/env python3
"""Two processes create an RCCL communicator; rank 1 names a GPU that does not exist.  Both must fail within seconds, each with the
reason (the verdict exchange right behind the join) — not rank 0 after SS_COMM_TIMEOUT_S inside the rendezvous."""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from soundscope_amd.distributed import Comm
rank = int(sys.argv[1])
t0 = time.time()
try:
    Comm(rank, 2, sys.argv[2], transport="rccl", device=(0 if rank == 0 else 97))
    print(f"rank {rank}: communicator created?!")
except Exception as e:
    print(f"rank {rank}: refused after {time.time() - t0:.1f} s: {e}")
''' % root
f = f"/tmp/ss_bad_rank_{os.getpid()}.rdzv"
ps = [subprocess.Popen([sys.executable, "-c", code, str(r), f], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
for p in ps:
    out, _ = p.communicate(timeout=150)
    print(out.strip()[-300:])
