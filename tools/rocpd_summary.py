#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace and/or PMC) as text.

Kernel rows are grouped by (kernel name, grid size), so the launches of different BASELINE configurations inside one
bench.py process stay apart (config 3's 1024-stream launches vs the config 2 / config 5 extras of the same kernel).
For every group two averages are given: over all launches, and over the launches that ran ALONE on the device (no other
kernel of >= 50 us overlapped them in time) — bench.py's overlapped steps run the spectrum kernel beside the time-domain
kernel, which stretches both; its per-kernel HIP-event numbers come from the sequential timing pass, i.e. they must
agree with the "alone" column.

usage: rocpd_summary.py results.db [out.txt]
"""
import sqlite3
import sys
from collections import defaultdict


def cols_of(c, table):
    return [r[1] for r in c.execute(f"pragma table_info('{table}')")]


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    kc = cols_of(c, "kernels")
    gx = "grid_x" if "grid_x" in kc else ("grid_size" if "grid_size" in kc else "0")
    extra = [x for x in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_x") if x in kc]
    sel = ", ".join(["name", "start", "end", gx] + extra)
    rows = list(c.execute(f"select {sel} from kernels order by start"))
    big = [(r[1], r[2]) for r in rows if r[2] - r[1] >= 50_000]          # intervals of kernels >= 50 us
    groups = defaultdict(list)
    for r in rows:
        name, s, e, g = r[0], r[1], r[2], r[3]
        alone = not any(bs < e and be > s and (bs, be) != (s, e) for bs, be in big)
        groups[(name, g)].append((e - s, alone, r[4:]))
    total = sum(d for v in groups.values() for d, _, _ in v) or 1
    print("KERNEL_DISPATCH stats by (kernel, grid); durations in us; 'alone' = launches no other kernel >= 50 us overlapped", file=out)
    print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_us':>11} {'pct':>6} {'alone':>6} {'alone_avg':>10}  "
          + " ".join(f"{x[:9]:>9}" for x in extra) + f" {'grid':>10}  name", file=out)
    for (name, g), v in sorted(groups.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
        ds = [d for d, _, _ in v]
        al = [d for d, a, _ in v if a]
        ex = v[0][2]
        print(f"{len(ds):6d} {sum(ds)/len(ds)/1e3:10.2f} {min(ds)/1e3:10.2f} {max(ds)/1e3:10.2f} {sum(ds)/1e3:11.2f} {100*sum(ds)/total:6.2f} "
              f"{len(al):6d} {(sum(al)/len(al)/1e3 if al else float('nan')):10.2f}  "
              + " ".join(f"{str(x):>9}" for x in ex) + f" {str(g):>10}  {name}", file=out)
    # PMC, if any: per (kernel, grid, counter)
    pm = []
    try:
        cc = cols_of(c, "counters_collection")
        if cc:
            g = "grid_size" if "grid_size" in cc else "0"
            vcol = "value" if "value" in cc else ("counter_value" if "counter_value" in cc else None)
            if vcol and "counter_name" in cc:
                pm = list(c.execute(f"""select kernel_name, {g}, counter_name, count(*), avg({vcol}), sum({vcol})
                                        from counters_collection group by kernel_name, {g}, counter_name
                                        order by kernel_name, {g}, counter_name"""))
    except sqlite3.OperationalError:
        pm = []
    if not pm:
        try:
            pm = list(c.execute("""select k.name, k.grid_x, p.counter_name, count(*), avg(p.counter_value), sum(p.counter_value)
                                   from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id
                                   group by k.name, k.grid_x, p.counter_name order by k.name, k.grid_x, p.counter_name"""))
        except sqlite3.OperationalError:
            pm = []
    if pm:
        print("\nPMC counters by (kernel, grid): dispatches, per-dispatch average, sum over dispatches", file=out)
        for r in pm:
            print(f"{r[3]:6d} {r[4]:20.1f} {r[5]:22.1f}  {r[2]:32s} grid={r[1]}  {r[0]}", file=out)


if __name__ == "__main__":
    main()
