#!/usr/bin/env python3
"""Prints the dispatch sequence around the largest idle gaps. usage: python profiles/seq_rocpd.py <db> [n_gaps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gaps = sorted(((rows[i + 1][1] - rows[i][2], i) for i in range(len(rows) - 1)), reverse=True)
def sh(x): return x.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mistark::", "")[:50] or "<unnamed>"
# skip the first few (warm-up allocations); show gaps ranked 10.. to see the recurring ones
for g, i in gaps[int(sys.argv[3]) if len(sys.argv) > 3 else 8:][:n]:
    print("gap %.1f us at dispatch %d (t=%.3f ms)" % (g / 1e3, i, (rows[i][2] - rows[0][1]) / 1e6))
    for j in range(max(0, i - 3), min(len(rows), i + 5)):
        print("   %s %-50s dur %.1f us%s" % ("*" if j == i else " ", sh(rows[j][0]), (rows[j][2] - rows[j][1]) / 1e3, "   <-- gap follows" if j == i else ""))
