#!/usr/bin/env python3
"""pmc_query.py <results.db> <kernel-substring> -- counters of the matching kernels in a rocprofv3 --pmc database, summed per dispatch
(one line per dispatch: duration, then every collected counter).  Used for the notes under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2]
rows = db.execute("select dispatch_id, substr(kernel_name, 1, 48), counter_name, sum(value), max(end - start) from counters_collection "
                  "where kernel_name like ? group by dispatch_id, counter_name order by dispatch_id", (f"%{sub}%",)).fetchall()
by = {}
for disp, name, cname, val, dur in rows:
    by.setdefault((disp, name), {})[cname] = val
    by[(disp, name)]["_ns"] = dur
for (disp, name), c in by.items():
    ns = c.pop("_ns")
    print(f"dispatch {disp} {name} {ns/1e6:.3f} ms " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
