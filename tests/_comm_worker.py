"""Worker for tests/test_distributed_comm.py: one rank of the sharded corpus gate through the PRODUCT's own
communicator (ss_comm_*, host-TCP transport so that it runs without a GPU).  No torch in this process.

Each rank computes the histograms of ITS shard of streams (with the oracle standing in for the per-GPU kernels),
the ranks all-reduce the 2x1000 u64 histograms with ss_comm_allreduce_u64_sum, every rank evaluates the gate
redundantly, and the max-over-ranks clock and the barrier bench.py relies on are exercised as well."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import make_stereo  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from soundscope_amd.distributed import Comm, corpus_gate, shard_streams  # noqa: E402

N_STREAMS, RATE, FRAMES = 7, 48000, 48000 * 4


def stream(i):
    return make_stereo(1000 + i, FRAMES, RATE, level=0.05 + 0.13 * i, gap=(i % 3 == 0))


def hist_of(ids):
    h = np.zeros(2000, np.uint64)
    for i in ids:
        m = po.Meter(2, RATE)
        m.add_frames(stream(i))
        h[:1000] += m.block_hist()
        h[1000:] += m.st_hist()
    return h


def main():
    assert "torch" not in sys.modules
    if len(sys.argv) > 3:                       # explicit rendezvous: rank world file
        comm = Comm(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], transport="host-tcp")
    else:                                       # RANK / WORLD_SIZE from a launcher
        comm = Comm.from_env("host-tcp")
    rank, world = comm.rank, comm.size
    assert world == int(os.environ.get("WORLD_SIZE", sys.argv[2] if len(sys.argv) > 2 else 1))
    first, count = shard_streams(N_STREAMS, rank, world)
    mine = hist_of(range(first, first + count))
    comm.barrier()
    got = comm.allreduce_sum_u64(mine.copy())
    ref = hist_of(range(N_STREAMS))
    assert np.array_equal(got, ref), "all-reduced histogram != single-process sum"
    gi, glra = corpus_gate(got)
    assert gi == po.gated_loudness_hist(ref[:1000]) and glra == po.loudness_range_hist(ref[1000:])
    # max-over-ranks clock
    t = comm.allreduce_max_f64(np.array([float(rank) + 0.5, -float(rank)]))
    assert t[0] == world - 0.5 and t[1] == 0.0
    # repeated collectives stay in step
    for k in range(20):
        v = comm.allreduce_sum_u64(np.full(3, rank + k, np.uint64))
        assert int(v[0]) == sum(r + k for r in range(world))
    comm.barrier()
    if rank == 0:
        print(f"COMM_OK world={world} transport={comm.transport} corpus_I={gi:.4f} LRA={glra:.2f}")
    comm.close()
    assert "torch" not in sys.modules


if __name__ == "__main__":
    main()
