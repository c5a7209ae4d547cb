#!/usr/bin/env python3
"""Per-kernel busy time and the idle gaps between consecutive dispatches inside the PCG loop, from a rocprofv3 rocpd database.
usage: python profiles/gaps_rocpd.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
import collections

busy = collections.Counter()
gap_after = collections.Counter()
cnt = collections.Counter()
for i, (name, s, e) in enumerate(rows):
    short = name.split("(")[0].replace("void ", "").replace("mistark::", "")[:40]
    busy[short] += e - s
    cnt[short] += 1
    if i + 1 < len(rows):
        g = rows[i + 1][1] - e
        if 0 < g < 200000:   # gaps up to 0.2 ms: back-to-back launches (longer ones are host phases)
            gap_after[short] += g
print("%-42s %8s %10s %10s %10s" % ("kernel", "calls", "busy_ms", "avg_us", "gap_after_avg_us"))
for k, v in busy.most_common(14):
    print("%-42s %8d %10.3f %10.2f %10.2f" % (k, cnt[k], v / 1e6, v / 1e3 / cnt[k], gap_after[k] / 1e3 / cnt[k]))
span = rows[-1][2] - rows[0][1]
print("span %.1f ms, busy %.1f ms, short gaps %.1f ms" % (span / 1e6, sum(busy.values()) / 1e6, sum(gap_after.values()) / 1e6))

# ---- inside the linear solves: from k_block_diag_inverse to the last PCG kernel before the next non-PCG kernel
pcg_names = ("k_block_diag_inverse", "k_pcg_init", "k_pcg_init2", "k_spmv_fused", "k_pcg_step", "k_pcg_dir", "__amd_rocclr_copyBuffer", "k_spmv_combine")
solves = []
cur = None
for name, s, e in rows:
    short = name.split("(")[0].replace("void ", "").replace("mistark::", "").split("<")[0]
    if short == "k_block_diag_inverse":
        if cur:
            solves.append(cur)
        cur = dict(start=s, end=e, busy=e - s, n=1, gaps=[])
    elif cur is not None and short in pcg_names:
        cur["gaps"].append(s - cur["end"])
        cur["end"] = e
        cur["busy"] += e - s
        cur["n"] += 1
    elif cur is not None:
        solves.append(cur)
        cur = None
if cur:
    solves.append(cur)
if solves:
    span = sum(x["end"] - x["start"] for x in solves)
    busy = sum(x["busy"] for x in solves)
    allg = sorted((g for x in solves for g in x["gaps"]), reverse=True)
    print("linear solves: %d, span %.1f ms, busy %.1f ms, idle %.1f ms; gaps > 5 us: %d (sum %.1f ms), top gaps us: %s" % (
        len(solves), span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(1 for g in allg if g > 5000), sum(g for g in allg if g > 5000) / 1e6,
        [round(g / 1e3, 1) for g in allg[:8]]))

# ---- all idle time attributed to the kernel that precedes it (where does the GPU wait for the host?)
idle = collections.Counter()
idle_n = collections.Counter()
for i, (name, s, e) in enumerate(rows[:-1]):
    short = name.split("(")[0].replace("void ", "").replace("mistark::", "")[:48]
    g = rows[i + 1][1] - e
    if g > 0:
        idle[short] += g
        idle_n[short] += 1
print("idle time by preceding kernel (total ms, count, avg us):")
for k, v in idle.most_common(16):
    print("  %-50s %9.2f %7d %9.1f" % (k, v / 1e6, idle_n[k], v / 1e3 / idle_n[k]))

# ---- idle gaps > 20 us by (previous kernel -> next kernel): which host round trips cost the most
pair = collections.Counter()
pair_n = collections.Counter()
def sh(n):
    return n.split("(")[0].replace("void ", "").replace("mistark::", "").replace("(anonymous namespace)::", "")[:34]
for i, (name, s, e) in enumerate(rows[:-1]):
    g = rows[i + 1][1] - e
    if g > 20000:
        k = sh(name) + " -> " + sh(rows[i + 1][0])
        pair[k] += g
        pair_n[k] += 1
print("idle gaps > 20 us by (previous -> next) kernel (total ms, count, avg us):")
for k, v in pair.most_common(22):
    print("  %-74s %8.2f %6d %8.1f" % (k, v / 1e6, pair_n[k], v / 1e3 / pair_n[k]))
