#!/usr/bin/env python3
"""Basic blocks of ONE kernel of an AMDGPU assembly file with their instruction mix — the static side of a kernel's instruction budget
(DESIGN 3.2): tools/isa_blocks.py <file.s> <mangled-name substring> [first block] [last block]
Columns: block index, label, VALU (f64 fma/mul/add | cvt | dpp/readlane/permute moves | other), MFMA, LDS, VMEM, SALU, waits/nops,
the s_setprio values inside (phase marks of the tile loop), and where the block branches."""
import re, sys
s = open(sys.argv[1]).read()
names = [m for m in re.findall(r'^(\S+):\s*; @', s, re.M) if sys.argv[2] in m]
name = names[0]
i = s.index('\n' + name + ':'); j = s.index('.Lfunc_end', i)
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in s[i:j].split('\n')[2:]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: cur = [m.group(1), []]; blocks.append(cur); continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    cur[1].append(t)
idx = {b[0]: k for k, b in enumerate(blocks)}
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(blocks) - 1
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_setprio') or op.startswith('s_sleep'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'):
        if '_f64' in op and op.startswith(('v_fma', 'v_mul', 'v_add', 'v_fmac')): return 'f64'
        if op.startswith('v_cvt'): return 'cvt'
        if 'dpp' in op or op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane')): return 'move'
        return 'valu'
    return 'other'
tot = {}
print(f"{'#':>4} {'label':<12} {'f64':>4} {'cvt':>4} {'move':>4} {'valu':>4} {'mfma':>4} {'lds':>4} {'vmem':>4} {'salu':>4} {'wait':>4}  notes")
for k, b in enumerate(blocks):
    if k < lo or k > hi: continue
    c = {}
    notes = []
    for ins in b[1]:
        op = ins.split()[0]
        dpp = 'dpp' in ins or 'row_' in ins or 'quad_perm' in ins
        kcls = 'move' if (dpp and op.startswith('v_mov')) else cls(op)
        c[kcls] = c.get(kcls, 0) + 1
        if op == 's_setprio': notes.append('prio' + ins.split()[1])
        m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', ins)
        if m:
            t = m.group(1) or m.group(2)
            notes.append(('^' if idx.get(t, 1 << 30) <= k else 'v') + str(idx.get(t, '?')))
    for q, v in c.items(): tot[q] = tot.get(q, 0) + v
    print(f"{k:>4} {b[0]:<12} " + ' '.join(f"{c.get(q, 0):>4}" for q in ('f64', 'cvt', 'move', 'valu', 'mfma', 'lds', 'vmem', 'salu', 'wait')) + '  ' + ' '.join(notes))
print('total', tot)
