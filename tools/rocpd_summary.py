#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace and/or PMC) as text.

usage: rocpd_summary.py results.db [out.txt]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    q = f"""select {name_col}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start),
               max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_size), max(workgroup_size)
            from kernels group by {name_col} order by sum(end-start) desc"""
    try:
        rows = list(c.execute(q))
    except sqlite3.OperationalError:
        q = f"select {name_col}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by {name_col} order by sum(end-start) desc"
        rows = [r + (None,) * 7 for r in c.execute(q)]
    total = sum(r[5] for r in rows) or 1
    print("KERNEL_DISPATCH stats (durations in us)", file=out)
    print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_us':>11} {'pct':>6}  vgpr agpr sgpr    lds scratch       grid  wg  name", file=out)
    for r in rows:
        name = r[0]
        print(f"{r[1]:6d} {r[2]/1e3:10.2f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} {r[5]/1e3:11.2f} {100*r[5]/total:6.2f}  "
              f"{str(r[6]):>4} {str(r[7]):>4} {str(r[8]):>4} {str(r[9]):>6} {str(r[10]):>7} {str(r[11]):>10} {str(r[12]):>3}  {name}", file=out)
    # PMC, if any
    try:
        pm = list(c.execute("""select k.name, p.counter_name, count(*), avg(p.value), sum(p.value)
                               from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id
                               group by k.name, p.counter_name order by k.name, p.counter_name"""))
    except sqlite3.OperationalError:
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            pm = list(c.execute("""select kernel_name, counter_name, count(*), avg(value), sum(value)
                                   from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"""))
        except sqlite3.OperationalError as e:
            pm = []
    if pm:
        print("\nPMC counters (per dispatch average, sum over dispatches)", file=out)
        for r in pm:
            print(f"{r[2]:6d} {r[3]:20.1f} {r[4]:22.1f}  {r[1]:32s} {r[0]}", file=out)


if __name__ == "__main__":
    main()
