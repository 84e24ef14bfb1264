#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel trace) per kernel: total ms, share, launches, average us."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
                        f"group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("kernel,total_ms,share_pct,launches,avg_us")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r[0][:120]},{r[2]:.3f},{100 * r[2] / tot:.1f},{r[1]},{r[3]:.1f}")
