"""Multi-GPU plumbing for the corpus-level integrated-LUFS gate.

Streams are independent, so the corpus is sharded across ranks with no data-path
exchange; the only collective is ONE all-reduce (sum) of the two 1000-bin u64
histograms (block energies, short-term energies) — 16 000 bytes, latency-bound on
xGMI — after which every rank evaluates the gate redundantly
(ebur128 loudness_global_multiple semantics).  torch.distributed is transport only:
backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import numpy as np


def shard_streams(n_total: int, rank: int, world: int):
    """Contiguous block partition of stream ids [0, n_total): -> (first, count)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def allreduce_histograms(hist_tensor):
    """In-place SUM all-reduce of an int64 tensor [2000] (block hist ++ short-term hist)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_tensor, op=dist.ReduceOp.SUM)
    return hist_tensor


def corpus_gate(hist2000):
    """(integrated LUFS, LRA) of the all-reduced histograms."""
    from .batch import corpus_integrated_lufs, corpus_loudness_range
    h = np.asarray(hist2000).astype(np.uint64)
    return corpus_integrated_lufs(h[:1000]), corpus_loudness_range(h[1000:])
