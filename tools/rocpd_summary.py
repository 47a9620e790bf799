#!/usr/bin/env python3
"""rocpd_summary.py <results.db> [out.md] -- per-kernel stats (calls, total, avg, min, max, %) from a rocprofv3 rocpd
SQLite database (`rocprofv3 --kernel-trace --stats` writes this format on ROCm 7.2).  Used to produce profiles/*.md."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % | VGPR | LDS B |", "|---|---|---|---|---|---|---|---|---|"]
for name, calls, tot, avg, mn, mx, vg, lds in rows:
    short = name.split("(")[0].replace("zk::", "")
    lines.append(f"| {short[:60]} | {calls} | {tot/1e6:.3f} | {avg/1e6:.4f} | {mn/1e6:.4f} | {mx/1e6:.4f} | {100*tot/total:.1f} | {vg} | {lds} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out + "\n")
