#!/usr/bin/env python3
"""Where the GPU idles: gaps between consecutive dispatches, summed per (kernel before, kernel after) pair, inside the last `frac` of the
trace (the timed region of bench.py). usage: python profiles/idle_rocpd.py <db> [min_gap_us=8] [frac=0.5]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 8e3
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
t0 = rows[0][1] + (1.0 - frac) * (rows[-1][2] - rows[0][1])
rows = [r for r in rows if r[1] >= t0]
def sh(x):
    x = x.replace("(anonymous namespace)::", "").replace("void ", "").replace("mistark::", "")
    if x.startswith("rocprim") or x.startswith("hipcub"): return "rocprim"
    return x.split("(")[0].split("<")[0][:28]
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
pair = collections.Counter(); cnt = collections.Counter()
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g >= min_gap:
        pair[(sh(n0), sh(n1))] += g; cnt[(sh(n0), sh(n1))] += 1
print("window %.1f ms, busy %.1f ms (%.0f%%), idle in gaps >= %.0f us: %.1f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, min_gap / 1e3, sum(pair.values()) / 1e6))
n_eval = sum(1 for n, _, _ in rows if "k_eval_tet_closed" in n and "true, true" in n)
print("Newton iterations in the window (P+g+H tet evaluations): %d" % n_eval)
print("%-30s -> %-30s %7s %10s %9s" % ("after", "before", "count", "total_ms", "avg_us"))
for k, v in pair.most_common(22):
    print("%-30s -> %-30s %7d %10.3f %9.1f" % (k[0], k[1], cnt[k], v / 1e6, v / 1e3 / cnt[k]))
