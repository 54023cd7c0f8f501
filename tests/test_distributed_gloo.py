"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the sharded corpus gate."""
import os
import subprocess
import sys

import pytest

from soundscope_amd.distributed import shard_streams

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_streams_partitions():
    for n in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_streams(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert shard_streams(8192, 3, 8) == (3072, 1024)


@pytest.mark.parametrize("world", [2, 3])
def test_corpus_gate_allreduce_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29400 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_gloo_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"GLOO_OK world={world}" in p.stdout


def test_bench_two_rank_launch_fails_fast_without_a_gpu():
    """The driver's N > 1 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    on a box with no GPU: every rank must leave with a non-zero status and a clear message within seconds — no rank may sit
    in a rendezvous waiting for a peer that has already given up, and there is no CPU path to fall back to."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the device-less launch check; the GPU box runs bench.py itself")
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120, cwd=root)
    took = time.time() - t0
    out = p.stdout + p.stderr
    assert p.returncode != 0, out[-2000:]
    assert took < 60.0, f"the launch took {took:.0f} s to fail"
    # one clear line per rank that got as far as looking for a device: the launcher terminates the peers of the first rank that
    # fails (its monitor polls every 0.1 s), so on a loaded host the slower rank may be stopped while it still imports
    assert 1 <= out.count("bench.py needs a GPU") <= 2, out[-3000:]
    assert '"metric"' not in p.stdout                                    # and no bench line from a run that measured nothing
