#!/usr/bin/env python3
"""rocpd_timeline.py <results.db> [first_kernel] [last_kernel] -- the kernels of ONE call in launch order with start offsets, durations and the
idle gap before each (rocprofv3 --kernel-trace rocpd database): where a latency-bound call (a small MSM) spends its time between kernels.
Defaults: the last k_msm_digits ... k_msm_final29 span of the trace."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_msm_digits"
last = sys.argv[3] if len(sys.argv) > 3 else "k_msm_final29"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("zk::", "").replace("void ", "")
idx_last = max((i for i, r in enumerate(rows) if last in r[0]), default=None)
if idx_last is None:
    sys.exit("no %s in the trace" % last)
idx_first = max(i for i, r in enumerate(rows[: idx_last + 1]) if first in r[0])
t0 = rows[idx_first][1]
prev_end = t0
print("| kernel | start us | duration us | gap before us |\n|---|---|---|---|")
busy = 0
for name, s, e in rows[idx_first: idx_last + 1]:
    print(f"| {short(name)[:48]} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} |")
    busy += e - s
    prev_end = max(prev_end, e)
span = rows[idx_last][2] - t0
print(f"\nspan {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us")
