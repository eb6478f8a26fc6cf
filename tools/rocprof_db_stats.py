#!/usr/bin/env python3
"""Per-kernel call counts and average durations from a rocprofv3 results database (the default output format when
--output-format is not given): python tools/rocprof_db_stats.py <results.db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
like = "%" + (sys.argv[2] if len(sys.argv) > 2 else "") + "%"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
     "where s.kernel_name like ? group by 1 order by 4 desc limit 40")
for name, calls, avg, tot in db.execute(q, (like,)):
    print(f"{name[:90]:90s} {calls:6d} {avg / 1e6:9.3f} ms {tot / 1e6:9.2f} ms")
