#!/usr/bin/env python3
"""roctx_summary.py DB [OUT.md] -- what MI355_TRACE=2 buys: reads the rocpd database of
`MI355_TRACE=2 rocprofv3 --marker-trace --kernel-trace -d DIR -o NAME -- <program>` and prints, per C-ABI entry point (= roctx range), the calls,
the host time inside the range, and the kernels that were dispatched from inside it with their GPU time."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
api = collections.OrderedDict()
for msg, dur in cur.execute("select extdata, duration from regions where category = 'MARKER_CORE_RANGE_API' order by start"):
    import json
    name = json.loads(msg).get("message", "?")
    a = api.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += dur / 1e6
kern = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for region, name, dur in cur.execute("select region, name, duration from kernels"):
    k = kern[region or "(outside any range)"][name.split("(")[0]]; k[0] += 1; k[1] += dur / 1e6
print("| entry point (roctx range) | calls | host ms in range | kernels dispatched inside (launches, GPU ms) |", file=out)
print("|---|---|---|---|", file=out)
for name, (calls, ms) in sorted(api.items(), key=lambda kv: -sum(v[1] for v in kern[kv[0]].values())):
    ks = sorted(kern[name].items(), key=lambda kv: -kv[1][1])
    desc = "; ".join(f"`{k}` x{c} {t:.2f}" for k, (c, t) in ks[:6]) + (f"; ... {len(ks) - 6} more" if len(ks) > 6 else "")
    print(f"| `{name}` | {calls} | {ms:.2f} | {desc or '-'} |", file=out)
print("\nTop kernels overall:\n", file=out)
print("| kernel | launches | total GPU ms | average us |", file=out); print("|---|---|---|---|", file=out)
for name, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 15"):
    print(f"| `{name.split('(')[0]}` | {calls} | {tot / 1e3:.2f} | {avg:.1f} |", file=out)
