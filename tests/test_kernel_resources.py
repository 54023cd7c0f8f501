"""Register allocation of the shipped kernels is gated: no instantiation may spill (private_segment_fixed_size > 0) unless it is on the
explicit allow-list below, and the instantiations the benchmark, the tick driver and an integrator's first shapes launch must not
spill at all.  CPU test: hipcc cross-compiles the kernel sources for gfx950 with -Rpass-analysis=kernel-resource-usage (device only,
no GPU needed, the four translation units side by side) and the remarks are parsed.

Template parameters of k_time_domain<FACTOR, RING, CT, WAVE, WPS, SPLIT, LATE>: true-peak oversampling, streaming ring, compile-time
channel count (0 = general), fused decimation (0 none, 1 general bins, 2 / 3 whole samples per bin), waves per SIMD the build is
register-allocated for, workgroup-shared tiles, state applied behind the scan (soundscope_amd/csrc/ss_td_impl.h).  The launcher takes
the WPS = 3 build (up to 168 VGPRs) whenever the grid fits three workgroups per CU and the WPS = 4 build (128 VGPRs) above that."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "soundscope_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ["ss_td_f4.hip", "ss_td_f2.hip", "ss_td_f0.hip", "ss_fft.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]       # the Makefile's CXXFLAGS

# Instantiations that may spill, with the most bytes per lane they may use (10 of 138 do at the end of round 6; 27 of 150 did in round
# 5).  All of them are register builds for FOUR waves per SIMD (128 VGPRs) of forms whose three-waves build (168 VGPRs) is spill-free;
# the launcher takes the four-waves build only for grids of more than 768 workgroups, where it still beats the spill-free build by
# 10-37 % (profiles/r05_td_kernel_resources.txt).
#   SPLIT (whole-stream workgroups, SS_TD_WHOLE_STREAMS: opt-in, 11-16 % slower than time segments at such grids anyway)
ALLOW = {
    r"k_time_domain<[420], false, [28], [0123], 4, true, false>": 48,      # (ten of them, 8-44 B; round 5: 64)
}

# what bench.py, the tick driver and the first shapes of an integrator launch: never a spill
MUST_BE_CLEAN = [
    "k_time_domain<4, false, 2, 2, 4, false, false>",     # BASELINE config 3 (1024 x 10 s stereo, fused decimation), the headline
    "k_time_domain<0, false, 2, 0, 3, false, false>",     # its hand-over launch
    "k_time_domain<4, false, 8, 1, 3, false, false>",     # BASELINE config 5, 4x
    "k_time_domain<2, false, 8, 1, 3, false, false>",     # ... at the crate's 2x
    "k_time_domain<0, false, 8, 0, 3, false, false>",     # its hand-over launch
    "k_time_domain<4, false, 2, 2, 2, true, true>",       # BASELINE config 2: one file, segments on eight waves
    "k_time_domain<4, false, 2, 1, 4, false, false>",     # stereo at 44.1 kHz (the reference's default rate): general decimation bins, big batch
    "k_time_domain<4, false, 2, 1, 3, false, false>",
    "k_time_domain<4, false, 6, 1, 4, false, false>",     # 5.1, big batch
    "k_time_domain<4, false, 6, 1, 3, false, false>",
    "k_time_domain<4, false, 8, 1, 4, false, false>",     # eight channels, big batch
    "k_time_domain<4, false, 1, 2, 4, false, false>",     # mono corpora
    "k_time_domain<4, true, 2, 0, 3, true, true>",        # the handle's add_samples / a tick's loudness call (stereo)
    "k_time_domain<4, true, 0, 0, 3, true, true>",        # ... any channel count
    "k_tick<4, 2>", "k_tick<4, 0>",                       # a tick of the reference in one launch
    "k_fft4096_ms1<4, 12, false>",                        # the roofline kernel
    "k_fft4096_ms1<4, 12, true>",                         # columns-only
    "k_fft16k_run<false, 3>",                             # BASELINE config 5's spectrum
    "k_fft16k_run<true, 2>",                              # the reference's native window, stereo 48 kHz
    "k_fft4096_pairw", "k_fft16k", "k_fft_generic",
]


def _resources(src):
    """[(demangled kernel name without arguments, {remark key: value})] of one translation unit"""
    r = subprocess.run([HIPCC, *FLAGS, "--offload-device-only", "-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"],
                       cwd=CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        body = re.sub(r"^[^ ]*:\d+:\d+: +", "", m.group(1))
        k, _, v = body.partition(":")
        k, v = k.strip(), v.strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run(["c++filt"] + [x["name"] for x in rows], capture_output=True, text=True).stdout.strip().split("\n")
    out = []
    for x, n in zip(rows, names):
        n = re.sub(r"^void ", "", n.split("(")[0]).replace("ssk::", "")
        out.append((n, x))
    return out


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.exists(HIPCC) and shutil.which("c++filt")):
        pytest.skip("no hipcc / c++filt in this environment")
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        per_file = list(ex.map(_resources, SOURCES))
    ks = {}
    for rows in per_file:
        for n, x in rows:
            ks[n] = x
    return ks


def test_no_spill_outside_the_allow_list(kernels):
    assert len(kernels) > 100                                           # (the three time-domain units alone instantiate ~140)
    bad = []
    for n, x in sorted(kernels.items()):
        scratch = int(x.get("ScratchSize [bytes/lane]", "0"))
        if scratch == 0:
            continue
        limit = max((v for pat, v in ALLOW.items() if re.fullmatch(pat, n)), default=None)
        if limit is None or scratch > limit:
            bad.append(f"{n}: {scratch} B/lane of scratch" + ("" if limit is None else f" (allowed: {limit})"))
    assert not bad, "kernel instantiations that spill:\n  " + "\n  ".join(bad)


def test_launched_shapes_are_spill_free_at_their_occupancy(kernels):
    missing = [n for n in MUST_BE_CLEAN if not any(k == n or k.startswith(n + "<") or k == n.split("<")[0] for k in kernels)]
    assert not missing, f"instantiations that no longer exist (update the list with the launcher): {missing}"
    for n in MUST_BE_CLEAN:
        for k, x in kernels.items():
            if k == n or (("<" not in n) and k.split("<")[0] == n):
                assert int(x.get("ScratchSize [bytes/lane]", "0")) == 0, (k, x.get("ScratchSize [bytes/lane]"))
    # the headline kernels at the occupancy DESIGN states: four waves per SIMD at 128 VGPRs, three at 168
    td = kernels["k_time_domain<4, false, 2, 2, 4, false, false>"]
    assert int(td["VGPRs"]) <= 128 and int(td["Occupancy [waves/SIMD]"]) == 4
    ms1 = kernels["k_fft4096_ms1<4, 12, false>"]
    assert int(ms1["VGPRs"]) <= 168 and int(ms1["Occupancy [waves/SIMD]"]) == 3 and int(ms1["LDS Size [bytes/block]"]) <= 160 * 1024 // 3
    run = kernels["k_fft16k_run<false, 3>"]
    assert int(run["VGPRs"]) <= 128 and int(run["LDS Size [bytes/block]"]) <= 80 * 1024
