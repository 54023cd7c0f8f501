"""Worker for tests/test_distributed_gloo.py: one rank of the sharded corpus gate on CPU (gloo).

Each rank computes the block / short-term histograms of ITS shard of streams (with the oracle,
standing in for the per-GPU kernels), the ranks all-reduce the 2x1000 histograms over gloo (torch is
test-side transport here; the product's own collective is covered by _comm_worker.py), and every rank
evaluates the gate redundantly with the product's host logic."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import make_stereo  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from soundscope_amd.distributed import corpus_gate, shard_streams  # noqa: E402

N_STREAMS, RATE, FRAMES = 7, 48000, 48000 * 4


def stream(i):
    return make_stereo(1000 + i, FRAMES, RATE, level=0.05 + 0.13 * i, gap=(i % 3 == 0))


def hist_of(ids):
    h = np.zeros(2000, np.int64)
    for i in ids:
        m = po.Meter(2, RATE)
        m.add_frames(stream(i))
        h[:1000] += m.block_hist().astype(np.int64)
        h[1000:] += m.st_hist().astype(np.int64)
    return h


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    first, count = shard_streams(N_STREAMS, rank, world)
    t = torch.from_numpy(hist_of(range(first, first + count)))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    got_i, got_lra = corpus_gate(t.numpy())
    ref = hist_of(range(N_STREAMS))
    assert np.array_equal(t.numpy(), ref), "all-reduced histogram != single-process sum"
    assert got_i == po.gated_loudness_hist(ref[:1000].astype(np.uint64))
    assert got_lra == po.loudness_range_hist(ref[1000:].astype(np.uint64))
    # every rank holds the same answer
    v = torch.tensor([got_i, got_lra], dtype=torch.float64)
    lst = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(lst, v)
    assert all(torch.equal(lst[0], x) for x in lst)
    # shards tile the corpus exactly
    sizes = [shard_streams(N_STREAMS, r, world) for r in range(world)]
    assert sum(c for _, c in sizes) == N_STREAMS and all(sizes[r][0] + sizes[r][1] == sizes[r + 1][0] for r in range(world - 1))
    dist.barrier()
    if rank == 0:
        print(f"GLOO_OK world={world} corpus_I={got_i:.4f} LRA={got_lra:.2f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
