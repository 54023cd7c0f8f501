cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/perf_probe.py 64 2 > /dev/null 2>&1
db=$(find /tmp/kt -name '*.db' | head -1)
echo DB $db
python3 - <<PY
import sqlite3
c=sqlite3.connect("$db")
for (n,t) in c.execute("select name,type from sqlite_master where type in ('table','view')"):
    cols=[r[1] for r in c.execute(f"pragma table_info('{n}')")]
    print(t,n,cols)
PY
find /tmp/kt -type f | head
