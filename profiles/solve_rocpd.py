#!/usr/bin/env python3
"""Per linear solve (k_pcg_prologue, or k_block_diag_inverse in traces of earlier builds, ... last PCG kernel): wall span on the GPU, busy time, launched / effective CG iterations and
the idle time before the next kernel, from a rocprofv3 --kernel-trace rocpd database. usage: python profiles/solve_rocpd.py <db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
def sh(x): return x.split("(")[0].replace("void ", "").replace("mistark::", "").split("<")[0]
PCG = {"k_spmv_fused", "k_spmv_dir", "k_pcg_step", "k_pcg_dir", "k_pcg_init", "k_pcg_init2", "k_pcg_prologue", "k_block_diag_inverse", "k_spmv_combine", "k_copy_ctrl", "k_pcg_check"}
FIRST = "k_pcg_prologue" if any(sh(r[0]) == "k_pcg_prologue" for r in rows) else "k_block_diag_inverse"
solves = []
cur = None
for i, (name, s, e) in enumerate(rows):
    k = sh(name)
    if k == FIRST:
        if cur: solves.append(cur)
        cur = dict(start=s, end=e, busy=e - s, spmv=0, spmv_real=0, spmv_t=0, step_t=0, dir_t=0, gaps=0, last=i)
    elif cur is not None and (k in PCG or name.startswith("__amd_rocclr_copyBuffer")) and s - cur["end"] < 300000:
        if k in PCG:
            cur["gaps"] += max(0, s - cur["end"])
            cur["busy"] += e - s
            cur["end"] = e
            cur["last"] = i
            if k in ("k_spmv_fused", "k_spmv_dir"):
                cur["spmv"] += 1
                if e - s > 5000:
                    cur["spmv_real"] += 1
                    cur["spmv_t"] += e - s
            elif k == "k_pcg_step" and e - s > 3000: cur["step_t"] += e - s
            elif k == "k_pcg_dir" and e - s > 3000: cur["dir_t"] += e - s
    elif cur is not None:
        cur["next_gap"] = s - cur["end"]
        solves.append(cur)
        cur = None
if cur: solves.append(cur)
big = [s for s in solves if s["spmv_real"] >= 5]
if not big:
    print("no solves found"); sys.exit(0)
n = len(big)
avg = lambda key: sum(s.get(key, 0) for s in big) / n
print("%d solves with >= 5 effective iterations" % n)
print("avg span %.1f us, busy %.1f us, gaps inside %.1f us, idle before the next kernel %.1f us" % (avg("end") / 1e3 - avg("start") / 1e3, avg("busy") / 1e3, avg("gaps") / 1e3, avg("next_gap") / 1e3))
print("avg iterations launched %.1f, effective %.1f; per effective iteration: spmv %.2f us, step %.2f us, dir %.2f us" % (
    avg("spmv"), avg("spmv_real"), avg("spmv_t") / avg("spmv_real") / 1e3, avg("step_t") / avg("spmv_real") / 1e3, avg("dir_t") / avg("spmv_real") / 1e3))
