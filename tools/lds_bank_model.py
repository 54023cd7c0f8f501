#!/usr/bin/env python3
"""LDS bank-conflict model of the batch FFT kernels' window loop (k_fft4096_ms1 / k_fft4096_pairw / k_fft16k_run).

The per-instruction rules are those of MI355X_MICROARCH.md, section LDS: 64 banks of 4 bytes; a wave64 access is served
in fixed lane groups, one LDS cycle per group when no two lanes of a group address different dwords of one bank:
  ds_read_b64   two 32-lane groups,            bank (a/4) mod 64
  ds_read_b128  four (non-contiguous) 16-lane groups, bank (a/4) mod 64
  ds_write_b64  four contiguous 16-lane groups, bank (a/4) mod 32
The layout constants (row stride, plane stride, publish swizzle) are read from ss_fft.hip, so the numbers printed here are
those of the code in the tree.  `python tools/lds_bank_model.py` prints LDS-array cycles per wave and window for every access
of k_fft4096_ms1's window loop next to the conflict-free minimum; tests/test_lds_layout_model.py asserts they are equal, and
profiles/r03_ab_fft_lds_layout.txt holds the counters that confirmed the model (SQ_LDS_BANK_CONFLICT = 0, SQ_LDS_IDX_ACTIVE
313 cycles per wave and window against 306 modelled)."""
import os
import re

_G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
_G128 = _G128 + [[32 + x for x in g] for g in _G128]
GROUPS = {
    "r64": [list(range(0, 32)), list(range(32, 64))],
    "r128": _G128,
    "w64": [list(range(16 * i, 16 * i + 16)) for i in range(4)],
}
WIDTH = {"r64": 2, "r128": 4, "w64": 2}           # dwords per lane
BANKS = {"r64": 64, "r128": 64, "w64": 32}
IDEAL = {k: len(v) for k, v in GROUPS.items()}     # cycles per wave-instruction without conflicts


def cycles(kind, dword_addr, active=None):
    """LDS-array cycles of one wave-instruction; dword_addr[lane] = first dword the lane touches"""
    total = 0
    for grp in GROUPS[kind]:
        per_bank = {}
        for lane in grp:
            if active is not None and not active[lane]:
                continue
            for d in range(WIDTH[kind]):
                a = dword_addr[lane] + d
                per_bank.setdefault(a % BANKS[kind], set()).add(a)
        total += max((len(v) for v in per_bank.values()), default=0)
    return total


def layout_from_source(path=None):
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soundscope_amd", "csrc", "ss_fft.hip")
    src = open(path).read()
    row = int(re.search(r"constexpr int kRowB = (\d+);", src).group(1))
    m = re.search(r"#define SPEC_POS\(k\) \(\(k\) \^ \(\(\(k\) >> (\d+)\) & (\d+)\)\)", src)
    shift, mask = int(m.group(1)), int(m.group(2))
    return {"row": row, "plane": 16 * row, "spec": (lambda k: k ^ ((k >> shift) & mask))}


def ms1_window(layout, first_bin=2, n_bins=1705):
    """[(name, cycles per wave and window, conflict-free minimum)] for the window loop of k_fft4096_ms1 (complex = 2 dwords)"""
    row, plane, spec = layout["row"], layout["plane"], layout["spec"]
    out = []

    def add(name, kind, fn, reps):
        tot = 0
        for w in range(4):
            for r in reps:
                tot += cycles(kind, [2 * fn(w * 64 + lane, r) for lane in range(64)])
        out.append((name, tot / 4.0, float(IDEAL[kind] * len(list(reps)))))

    add("exchange 1 write (ka; tb, hi)", "w64", lambda t, ka: ka * plane + (t & 15) * row + (t >> 4), range(16))
    add("exchange 1 read  (hi; tb, 0..15)", "r64", lambda t, j: (t >> 4) * plane + (t & 15) * row + j, range(16))
    add("exchange 2 write (kb; hi, tb)", "w64", lambda t, kb: kb * plane + (t >> 4) * row + (t & 15), range(16))
    add("exchange 2 read  (hi; tb, 0..15)", "r64", lambda t, j: (t >> 4) * plane + (t & 15) * row + j, range(16))
    add("publish (kc; swizzled t)", "w64", lambda t, kc: kc * 256 + spec(t), range(14))
    add("second-pass twiddles [kb][tb]", "r64", lambda t, kb: 8192 + kb * 16 + (t & 15), range(1, 16))
    ngroups = (n_bins + 3) // 4
    tot, ideal = 0, 0
    for w in range(4):
        for i in range(2):
            for e in range(4):
                ak, am, act = [], [], []
                for lane in range(64):
                    g = w * 64 + lane + 256 * i
                    act.append(g < ngroups)
                    k = first_bin + 4 * min(g, ngroups - 1) + e
                    ak.append(2 * spec(k))
                    am.append(2 * spec((4096 - k) & 4095))
                tot += cycles("r64", ak, act) + cycles("r64", am, act)
                ideal += 2 * sum(1 for grp in GROUPS["r64"] if any(act[lane] for lane in grp))     # a group without an active lane is free
    out.append(("epilogue bins and mirrors", tot / 4.0, ideal / 4.0))
    return out


if __name__ == "__main__":
    lay = layout_from_source()
    print(f"row stride {lay['row']} complex, plane stride {lay['plane']}")
    rows = ms1_window(lay)
    for name, c, i in rows:
        print(f"  {name:36s} {c:7.1f} LDS cycles per wave and window (conflict-free: {i:.1f})")
    print(f"  {'total':36s} {sum(r[1] for r in rows):7.1f} (conflict-free: {sum(r[2] for r in rows):.1f})")
