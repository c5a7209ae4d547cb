"""Histogram of the projection kernel launches of a rocprofv3 kernel trace: list length (from the grid) against duration."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = [c for c in cols if "grid" in c.lower()]
print(gcol)
rows = list(cur.execute("select name, %s, (end-start)/1e3 from kernels where name like '%%k_project_eig%%'" % gcol[0]))
import collections
for nb, epw in (("<3>", 6), ("<4>", 5), ("<5>", 4), ("<6>", 3)):
    r = [(g // 256 * 4 * epw, d) for n, g, d in rows if nb in n]
    if not r: continue
    bins = collections.defaultdict(lambda: [0, 0.0])
    for ne, d in r:
        b = 1
        while b < ne: b *= 4
        bins[b][0] += 1; bins[b][1] += d
    print(nb, "calls", len(r), "total ms %.1f" % (sum(d for _, d in r) / 1e3))
    for b in sorted(bins): print("   <=%8d elements: %5d calls, avg %8.1f us, total %7.1f ms" % (b, bins[b][0], bins[b][1] / bins[b][0], bins[b][1] / 1e3))
