#!/usr/bin/env python3
"""One Newton iteration with ABSOLUTE times: start and end of every dispatch in microseconds from the iteration's first kernel, and the HSA queue
it ran on (the engine's streams: main, auxiliary, early evaluation, pattern side stream), PCG launches collapsed per solve.
usage: [ABS_MARK=<kernel name part>] python profiles/abs_rocpd.py <db> [which]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else ", 0")))
def sh(x): return x.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mistark::", "")[:56] or "<unnamed>"
import os
mark = os.environ.get("ABS_MARK")  # (another scene: the kernel that opens a unit, e.g. ABS_MARK="k_sweep<true, true>" = the friction search of a time step)
marks = [i for i, r in enumerate(rows) if (mark in r[0] if mark else ("k_eval_tet" in r[0] and ", true" in r[0].split("(")[0]))]
w = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[w], marks[w + 1]
t0 = rows[a][1]
queues = {}
print("Newton iteration %d of %d: span %.1f us (columns: start, end, queue, kernel, duration)" % (w, len(marks), (rows[b][1] - t0) / 1e3))
pcg = ("k_spmv_fused", "k_pcg_step", "k_pcg_dir")
i = a
while i < b:
    n, s, e, q = rows[i]
    qi = queues.setdefault(q, len(queues))
    name = sh(n)
    if any(p in name for p in pcg):
        j = i
        while j < b and any(p in sh(rows[j][0]) for p in pcg): j += 1
        print("%8.1f %8.1f  q%d  %4d PCG launches" % ((s - t0) / 1e3, (rows[j - 1][2] - t0) / 1e3, qi, j - i))
        i = j
        continue
    print("%8.1f %8.1f  q%d  %-56s %7.1f" % ((s - t0) / 1e3, (e - t0) / 1e3, qi, name, (e - s) / 1e3))
    i += 1
