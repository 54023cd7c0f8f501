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


def _planted_file(path, port, nonce=12345):
    with open(path, "w") as f:            # the library's own format: magic, port, id ("-" = host transport), nonce
        f.write(f"ssc2 {port} - {nonce}\n")


@pytest.mark.parametrize("kind", ["dead_port", "foreign_listener"])
def test_stale_rendezvous_file_is_not_consumed(kind, tmp_path):
    """A rendezvous file left by a crashed / earlier job must not be used: its port is dead, or whoever listens there
    does not answer with the file's nonce.  Ranks > 0 start first (so they see the stale file), rank 0 joins 2 s later
    and replaces it; everybody must end up in the NEW job."""
    import socket
    import time
    f = str(tmp_path / "rdzv")
    srv = None
    if kind == "dead_port":
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()     # nobody listens there any more
    else:
        srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(8); port = srv.getsockname()[1]    # accepts, never answers
    _planted_file(f, port)
    world = 3
    env = dict(os.environ, OMP_NUM_THREADS="1", WORLD_SIZE=str(world), SS_COMM_TIMEOUT_S="120")
    procs = {}
    for r in (2, 1):
        procs[r] = subprocess.Popen([sys.executable, WORKER, str(r), str(world), f], env=env, stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True)
    time.sleep(2.0)
    assert all(p.poll() is None for p in procs.values()), "a rank finished against a stale rendezvous file"
    procs[0] = subprocess.Popen([sys.executable, WORKER, "0", str(world), f], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)
    outs = {r: p.communicate(timeout=600) for r, p in procs.items()}
    if srv is not None:
        srv.close()
    for r, p in procs.items():
        assert p.returncode == 0, f"rank {r}: " + outs[r][0][-2000:] + outs[r][1][-4000:]
    assert f"COMM_OK world={world} transport=host-tcp" in outs[0][0]
    assert not os.path.exists(f)


def test_rendezvous_file_is_private_and_symlink_safe(tmp_path):
    """Rank 0 creates the file with O_EXCL | O_NOFOLLOW, mode 0600: a symlink planted at the temporary name is replaced,
    not followed, and the published file is not group/world readable (it holds the RCCL unique id)."""
    import stat
    import threading
    from soundscope_amd.distributed import Comm
    f = str(tmp_path / "rdzv")
    victim = tmp_path / "victim.txt"
    victim.write_text("untouched")
    os.symlink(str(victim), f + f".tmp.{os.getpid()}")            # what an attacker who guesses the temporary name would plant
    seen = {}

    def rank1():
        import time
        for _ in range(2000):
            if os.path.exists(f):
                seen["mode"] = stat.S_IMODE(os.stat(f).st_mode)
                break
            time.sleep(0.005)
        c = Comm(1, 2, f, transport="host-tcp")
        c.barrier(); c.close()

    t = threading.Thread(target=rank1)
    t.start()
    c = Comm(0, 2, f, transport="host-tcp")
    c.barrier()
    t.join(60)
    c.close()
    assert victim.read_text() == "untouched"
    assert seen.get("mode") == 0o600, seen
    assert not os.path.exists(f)


def test_back_to_back_inits_on_one_path(tmp_path):
    """Two communicators created one after the other on the SAME rendezvous path (a fast rank may re-enter the second
    init before rank 0 has removed the first file): the nonce handshake keeps the generations apart."""
    import threading
    from soundscope_amd.distributed import Comm
    import numpy as np
    f = str(tmp_path / "rdzv")
    errs = []

    def run(rank):
        try:
            for gen in range(6):
                c = Comm(rank, 2, f, transport="host-tcp")
                v = c.allreduce_sum_u64(np.array([gen * 10 + rank + 1], np.uint64))
                assert int(v[0]) == 2 * gen * 10 + 3, (gen, v)
                c.close()
        except Exception as e:       # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in (1, 0)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs


def test_comm_init_on_device_entry_point_and_its_argument_checks():
    """ss_comm_init_on_device (round 5): the rank's GPU is an argument, bound inside the call behind the HSA IPC default, so that a
    rank off device 0 need not (must not) call ss_set_device first.  On a machine without a GPU: the host transport ignores
    the device, a negative device is refused, and the RCCL transport fails with SS_ERR_DEVICE instead of touching anything."""
    import ctypes as C
    import numpy as np
    from soundscope_amd import _lib as L
    from soundscope_amd.distributed import Comm
    c = Comm(0, 1, None, transport="host-tcp", device=5)
    assert (c.rank, c.size, c.transport) == (0, 1, "host-tcp")
    assert list(c.allreduce_sum_u64(np.array([7, 9], np.uint64))) == [7, 9]
    c.close()
    h = C.c_void_p()
    assert L.lib().ss_comm_init_on_device(L.SS_COMM_HOST_TCP, 0, 1, -1, None, C.byref(h)) == L.SS_ERR_INVALID_ARG
    if L.lib().ss_device_count() <= 0:
        assert L.lib().ss_comm_init_on_device(L.SS_COMM_RCCL, 0, 1, 0, None, C.byref(h)) == L.SS_ERR_DEVICE


def test_a_rank_that_cannot_set_up_takes_every_rank_down_within_seconds(tmp_path):
    """RCCL transport, two ranks, no usable GPU on this box: what goes wrong on a rank BEFORE the ranks meet (no device, a device
    ordinal that does not exist, no librccl) is exchanged right behind the join, so every rank fails within seconds and names the
    reason — the healthy ones do not wait out SS_COMM_TIMEOUT_S for a peer that has already left (a rank whose LOCAL_RANK named
    no device cost its peer 2 x 180 s: tools/launch_path_two_ranks.sh).  Here both ranks lack a device."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("the device-less form; tools/probe_comm_bad_rank.py is the one-bad-rank form for a GPU box")
    f = str(tmp_path / "rdzv")
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from soundscope_amd.distributed import Comm\n"
            "t0 = time.time()\n"
            "try:\n"
            "    Comm(int(sys.argv[1]), 2, sys.argv[2], transport='rccl', device=int(sys.argv[1]))\n"
            "    print('CREATED')\n"
            "except Exception as e:\n"
            "    print(f'REFUSED {time.time() - t0:.1f} {e}')\n") % ROOT
    env = dict(os.environ, SS_COMM_TIMEOUT_S="60")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), f], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in (1, 0)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert time.time() - t0 < 30.0, outs
    for o in outs:
        assert "REFUSED" in o and "no HIP device" in o, o
