#!/usr/bin/env python3
"""Prints the dispatch chain of contact searches from a rocprofv3 kernel trace: every kernel from k_contact_vertices (or the first kernel of a
search that reuses the boxes) to k_table_bounds / k_route, with start offsets, durations and the gap in front. usage: search_rocpd.py <db> [n_searches] [skip]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 10
def sh(x): return x.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mistark::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")[:60] or "<unnamed>"
starts = [i for i, r in enumerate(rows) if "k_contact_vertices" in r[0]]
for s in starts[skip:skip + n]:
    t0 = rows[s][1]
    print("---- search at dispatch %d" % s)
    for j in range(s, min(s + 60, len(rows))):
        nm = sh(rows[j][0])
        print("  %8.1f us  +gap %6.1f  dur %6.1f  %s" % ((rows[j][1] - t0) / 1e3, (rows[j][1] - rows[j - 1][2]) / 1e3 if j > 0 else 0.0, (rows[j][2] - rows[j][1]) / 1e3, nm))
        if "k_route" in nm or ("k_contact_vertices" in nm and j > s):
            break
