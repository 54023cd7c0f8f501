"""Multi-GPU plumbing for the corpus-level integrated-LUFS gate — no PyTorch.

Streams are independent, so the corpus is sharded across ranks (one process per GPU) with no
data-path exchange; the only collective is ONE all-reduce (sum) of the two 1000-bin u64
histograms (block energies, short-term energies) — 16 000 bytes, latency-bound on xGMI —
after which every rank evaluates the gate redundantly (ebur128 loudness_global_multiple
semantics).  The collective is the library's own (`ss_comm_*`, csrc/ss_comm.cpp): RCCL opened
directly by the C-ABI library, or the same calls staged over loopback TCP for CPU-only tests
and ranks that share one GPU.  A launcher such as torchrun only provides RANK / WORLD_SIZE.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .analyzer import _check


def shard_streams(n_total: int, rank: int, world: int):
    """Contiguous block partition of stream ids [0, n_total): -> (first, count)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class Comm:
    """One rank of the job's communicator (ss_comm)."""

    def __init__(self, rank=None, world=None, rendezvous_file=None, transport="rccl", device=None):
        """rank None: everything from the launcher's environment (ss_comm_init_from_env: RANK, WORLD_SIZE, the rank's GPU from
        SS_COMM_DEVICE or LOCAL_RANK; a LOCAL_RANK beyond the visible devices — launchers that mask one GPU per rank — wraps onto
        them).  device (RCCL; ignored by the host transport, which touches no GPU): the rank's GPU, made current INSIDE the call, behind the HSA IPC default —
        a rank creates its communicator before any other GPU call and does not call ss_set_device first (include/soundscope_hip.h)."""
        t = {"rccl": L.SS_COMM_RCCL, "host-tcp": L.SS_COMM_HOST_TCP}[transport]
        self._h = C.c_void_p()
        if rank is None:
            # (an explicit device with ranks from the environment: handed to the library as SS_COMM_DEVICE for this one call)
            import os
            had = os.environ.get("SS_COMM_DEVICE")
            if device is not None:
                os.environ["SS_COMM_DEVICE"] = str(int(device))
            try:
                rc = L.lib().ss_comm_init_from_env(t, C.byref(self._h))
            finally:
                if device is not None:
                    if had is None:
                        os.environ.pop("SS_COMM_DEVICE", None)
                    else:
                        os.environ["SS_COMM_DEVICE"] = had
        else:
            f = rendezvous_file.encode() if rendezvous_file else None
            if device is None:
                rc = L.lib().ss_comm_init(t, int(rank), int(world), f, C.byref(self._h))
            else:
                rc = L.lib().ss_comm_init_on_device(t, int(rank), int(world), int(device), f, C.byref(self._h))
        if rc != L.SS_OK:
            self._h = None
            _check(rc)

    @classmethod
    def from_env(cls, transport="rccl"):
        """RANK / WORLD_SIZE (+ MASTER_PORT or SS_COMM_FILE) as exported by the launcher."""
        return cls(transport=transport)

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ss_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rank(self):
        return L.lib().ss_comm_rank(self._h)

    @property
    def size(self):
        """Rank count as the transport reports it (ncclCommCount for RCCL)."""
        return L.lib().ss_comm_size(self._h)

    @property
    def transport(self):
        return L.lib().ss_comm_transport_name(self._h).decode()

    @property
    def library_version(self):
        """"2.27.7"-style version of the librccl behind an RCCL communicator (ncclGetVersion); None for the host transport."""
        v = int(L.lib().ss_comm_library_version(self._h))
        if v <= 0:
            return None
        return f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else f"{v // 1000}.{v // 100 % 10}.{v % 100}"

    def barrier(self):
        _check(L.lib().ss_comm_barrier(self._h))

    def allreduce_sum_u64(self, a):
        """In-place SUM all-reduce of a host uint64 array; returns it."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        _check(L.lib().ss_comm_allreduce_u64_sum(self._h, a.ctypes.data_as(C.POINTER(C.c_uint64)), a.size))
        return a

    def allreduce_max_f64(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        _check(L.lib().ss_comm_allreduce_f64_max(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))
        return a


def corpus_gate(hist2000):
    """(integrated LUFS, LRA) of the all-reduced histograms (block ++ short-term)."""
    from .batch import corpus_integrated_lufs, corpus_loudness_range
    h = np.asarray(hist2000).astype(np.uint64)
    return corpus_integrated_lufs(h[:1000]), corpus_loudness_range(h[1000:])
