#!/usr/bin/env python3
"""Timeline of the kernels of a few session ticks out of a rocprofv3 rocpd result: start (relative to the tick's first kernel),
duration, queue and stream of every dispatch.   usage: tick_timeline.py results.db [ticks to print per session, default 2]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kc = [r[1] for r in c.execute("pragma table_info('kernels')")]
print("columns of `kernels`:", ", ".join(kc))
want = [x for x in ("queue_id", "stream_id", "queue", "stream", "tid") if x in kc]
rows = list(c.execute(f"select name, start, end, {', '.join(want) if want else '0'} from kernels order by start"))
# a tick = the dispatches around one k_time_domain<..., true> (SPLIT) launch: everything within 150 us of its start
td = [i for i, r in enumerate(rows) if "k_time_domain" in r[0] and "true>" in r[0].split("(")[0]]
seen = {}
for i in td:
    key = tuple(rows[i][3:])                       # (queue, stream) of the loudness chain = one session
    n = seen.get(key, 0)
    seen[key] = n + 1
    if not (40 <= n < 40 + per):                   # ticks 40.. of every session
        continue
    t0 = min(r[1] for r in rows[max(0, i - 6):i + 7] if abs(r[1] - rows[i][1]) < 150_000)
    print(f"--- session {key}, tick {n}")
    for r in rows[max(0, i - 6):i + 7]:
        if abs(r[1] - rows[i][1]) < 150_000:
            print(f"  +{(r[1] - t0) / 1e3:7.2f} us  {(r[2] - r[1]) / 1e3:7.2f} us  {dict(zip(want, r[3:]))}  {r[0].split('(')[0][:60]}")
