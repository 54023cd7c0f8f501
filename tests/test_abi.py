"""The C-ABI library loads and exports every symbol include/soundscope_hip.h declares (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import soundscope_amd as ssa
from soundscope_amd import _lib as L

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "soundscope_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = ctypes.CDLL(L.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in soundscope_hip.h but not exported"


def test_python_binding_covers_header():
    assert sorted(L.SYMBOLS) == declared_symbols()


def test_abi_version_and_status_strings():
    lib = L.lib()
    assert lib.ss_abi_version() == 2 == L.SS_ABI_VERSION
    assert lib.ss_status_string(0) == b"ok"
    assert b"NotAPowerOfTwo" in lib.ss_status_string(L.SS_ERR_NOT_POW2)


def test_no_cpu_fallback_without_device():
    """On a box without a GPU every compute entry point must fail loudly (SS_ERR_DEVICE)."""
    if L.lib().ss_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ssa.DeviceError):
        ssa.Analyzer()
    with pytest.raises(ssa.DeviceError):
        ssa.Analyzer.get_waveform(np.zeros(100, np.float32), 0.05)
    with pytest.raises(ssa.DeviceError):
        ssa.Batch(n_streams=1, frames_per_stream=48000)


def test_struct_sizes_match_header():
    # ss_batch_config: 8 x u32 + u64 + f64; ss_stream_result: 6 f64 + 2 u32; ss_batch_layout: 8 u32 + 2 u64
    assert ctypes.sizeof(L.BatchConfig) == 48
    assert ctypes.sizeof(L.StreamResult) == 56
    assert ctypes.sizeof(L.BatchLayout) == 48


def test_corpus_gate_host_helpers_match_oracle(oracle):
    """ss_corpus_* is host logic (A7/A8 on an all-reduced histogram); check it against the oracle."""
    rng = np.random.default_rng(3)
    for _ in range(20):
        h = np.zeros(1000, np.uint64)
        idx = rng.integers(300, 700, 40)
        h[idx] += rng.integers(1, 50, 40).astype(np.uint64)
        if rng.random() < 0.5:
            h[rng.integers(0, 200, 10)] += 5
        assert ssa.corpus_integrated_lufs(h) == oracle.gated_loudness_hist(h)
        assert ssa.corpus_loudness_range(h) == oracle.loudness_range_hist(h)
    z = np.zeros(1000, np.uint64)
    assert ssa.corpus_integrated_lufs(z) == -np.inf
    assert ssa.corpus_loudness_range(z) == 0.0


def build_c_client(tmp_path, name="cabi_client", env=None):
    """gcc -std=c99 on tests/cabi/<name>.c against the header and the in-tree library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    L.lib()                                              # builds the library on demand
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cabi", name + ".c"), "-o", exe,
                           "-L", libdir, "-lsoundscope_hip", "-lm", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr
    return dict(kv.split("=", 1) for kv in re.findall(r'(\w+=(?:"[^"]*"|\S+))', out.stdout))


def test_c99_batch_client_links_and_fails_loudly_without_device(tmp_path):
    """The batch / corpus-gate client (tests/cabi/cabi_batch.c) compiles as strict C99 against the header, links against
    the in-tree library, and without a GPU `ss_batch_create` returns SS_ERR_DEVICE."""
    kv = build_c_client(tmp_path, "cabi_batch")
    assert kv["abi"] == "2" and kv["sizeof_cfg"] == "48" and kv["sizeof_result"] == "56"
    if int(kv["devices"]) == 0:
        assert int(kv["create"]) == L.SS_ERR_DEVICE


def test_c99_client_links_and_fails_loudly_without_device(tmp_path):
    """The header is valid C99, every symbol the client uses resolves, and without a GPU the first compute
    entry point returns SS_ERR_DEVICE (no CPU fallback)."""
    kv = build_c_client(tmp_path)
    assert kv["abi"] == "2" and kv["sizeof_tick"] == "56"
    if int(kv["devices"]) == 0:
        assert int(kv["open"]) == L.SS_ERR_DEVICE
