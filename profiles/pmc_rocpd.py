#!/usr/bin/env python3
"""Average PMC counter value per dispatch for each kernel from a rocprofv3 --pmc rocpd database.
usage: python profiles/pmc_rocpd.py <results.db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "--schema" in sys.argv:
    for t in tables:
        print(t, [c[1] for c in cur.execute("pragma table_info('%s')" % t)])
    sys.exit(0)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
# rocpd: pmc_events(event_id -> kernel dispatch, pmc_id -> info_pmc(name), value); view counters_collection may exist
view = [t for t in tables if t.startswith("counters_collection")]
if view and "--real" in sys.argv:
    # dispatches that did real work only (the PCG driver's look-ahead launches after convergence are device-side no-ops)
    name = [a for a in sys.argv[2:] if not a.startswith("--")][0]
    vals = [v for (v,) in cur.execute("select value from %s where kernel_name like ? " % view[0], ("%" + name + "%",))]
    big = [v for v in vals if v > 0.1 * max(vals)]
    print("%s: %d dispatches, %d with real work, avg counter value over those: %.1f" % (name, len(vals), len(big), sum(big) / len(big)))
elif view:
    q = "select kernel_name, counter_name, count(*), avg(value), sum(value) from %s group by kernel_name, counter_name order by 5 desc" % view[0]
    print("%-70s %-14s %8s %16s" % ("kernel", "counter", "calls", "avg_per_dispatch"))
    for k, c, n, a, s in cur.execute(q):
        if sub in k:
            print("%-70s %-14s %8d %16.1f" % (k[:70], c, n, a))
else:
    print("no counters_collection view; tables:", tables)
