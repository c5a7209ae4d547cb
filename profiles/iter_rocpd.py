#!/usr/bin/env python3
"""Prints every dispatch of ONE Newton iteration (between two P+g+H tet evaluations in the middle of the trace) with the idle gap before
each kernel, PCG iterations collapsed. usage: python profiles/iter_rocpd.py <db> [which]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
def sh(x): return x.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mistark::", "")[:60] or "<unnamed>"
marks = [i for i, r in enumerate(rows) if "k_eval_tet" in r[0] and ("true>" in r[0].split("(")[0][-12:] or ", true" in r[0].split("(")[0])]
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if "k_eval_tet" in r[0]]
w = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[w], marks[w + 1]
print("Newton iteration %d of %d: dispatches %d..%d, span %.1f us" % (w, len(marks), a, b, (rows[b][1] - rows[a][1]) / 1e3))
busy = 0.0
i = a
pcg = ("k_spmv_fused", "k_pcg_step", "k_pcg_dir")
while i < b:
    n, s, e = rows[i]
    name = sh(n)
    gap = (s - rows[i - 1][2]) / 1e3
    if any(p in name for p in pcg):
        j = i
        tb = 0.0
        tg = 0.0
        while j < b and any(p in sh(rows[j][0]) for p in pcg):
            tb += (rows[j][2] - rows[j][1]) / 1e3
            if j > i: tg += (rows[j][1] - rows[j - 1][2]) / 1e3
            j += 1
        print("  gap %7.1f | %4d PCG launches: busy %.1f us, gaps inside %.1f us" % (gap, j - i, tb, tg))
        busy += tb
        i = j
        continue
    d = (e - s) / 1e3
    busy += d
    print("  gap %7.1f | %-60s %8.1f us" % (gap, name, d))
    i += 1
print("busy %.1f us of %.1f" % (busy, (rows[b][1] - rows[a][1]) / 1e3))
