"""N > 1 path of the PRODUCT's communicator on CPU (host-TCP transport of ss_comm_*, world sizes 2 and 3),
launched both directly (explicit rendezvous file) and through torchrun used purely as a process launcher
(RANK / WORLD_SIZE / MASTER_PORT -> ss_comm_init_from_env).  The product package must not import torch."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "_comm_worker.py")


def test_product_package_is_torch_free():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "soundscope_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+torch\b", src, re.M), f"{f} imports torch"
    assert not re.search(r"^\s*(import|from)\s+torch\b", open(os.path.join(ROOT, "bench.py")).read(), re.M)


@pytest.mark.parametrize("world", [2, 3])
def test_comm_host_tcp_explicit_rendezvous(world, tmp_path):
    f = str(tmp_path / "rdzv")
    env = dict(os.environ, OMP_NUM_THREADS="1", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), f], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in reversed(range(world))]   # rank 0 starts last
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-2000:] + e[-4000:]
    assert any(f"COMM_OK world={world} transport=host-tcp" in o for o, _ in outs)
    assert not os.path.exists(f)                 # rank 0 removes the rendezvous file once every rank has joined


def test_comm_host_tcp_from_launcher_env():
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.pop("SS_COMM_FILE", None)
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"COMM_OK world={world} transport=host-tcp" in p.stdout


def test_comm_single_rank_is_a_no_op():
    from soundscope_amd.distributed import Comm
    import numpy as np
    c = Comm(0, 1, None, transport="host-tcp")
    assert c.rank == 0 and c.size == 1
    c.barrier()
    assert list(c.allreduce_sum_u64(np.array([5, 7], np.uint64))) == [5, 7]
    c.close()
