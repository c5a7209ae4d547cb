#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output of `--kernel-trace --stats`) database into a per-kernel table.
usage: python profiles/summarize_rocpd.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary; total kernel time %.3f ms over %d dispatches" % (tot, sum(r[1] for r in rows)))
print("%-72s %7s %10s %6s %10s %9s %10s %5s %5s %5s %8s %6s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us", "vgpr", "agpr", "sgpr", "scratch", "lds"))
for r in rows:
    print("%-72s %7d %10.3f %6.1f %10.2f %9.2f %10.2f %5d %5d %5d %8d %6d" % (r[0][:72], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]))
# SpMV launches that did real work (the PCG batches end with early-exit launches of ~1 us)
real = list(cur.execute("select count(*), avg(end-start)/1e3 from kernels where name like '%k_spmv%' and (end-start) > 5000"))
if real and real[0][0]:
    print("# k_spmv launches > 5 us (real work): n=%d avg=%.2f us" % (real[0][0], real[0][1]))
