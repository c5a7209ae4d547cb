#!/usr/bin/env python3
"""Per-rank stage times of a sharded in-process run (tools/shard_model.py W under rocprofv3 --kernel-trace): the W ranks share one stream,
so every kernel's duration is its solo duration; the slowest rank's share of a stage is estimated from the per-kernel maxima.
usage: python profiles/shard_rocpd.py <db> W n_newton_total n_cg_total"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
W, n_newton, n_cg = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = list(db.execute("select name, start, end from kernels order by start"))
def sh(x): return x.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mistark::", "")[:48]
stages = {
    "cg: spmv": ["k_spmv_fused"], "cg: step": ["k_pcg_step"], "cg: dir (+ ghosts)": ["k_pcg_dir"], "cg: fold / fold + pack": ["k_fold_partials", "k_fold_pack"],
    "halo (gradient): pack+unpack": ["k_pack_rows", "k_unpack_ghosts"], "exchange (in-process copy kernel)": ["k_allgather_local"],
    "eval tets": ["k_eval_tet_closed", "k_grad_gather"], "eval other": ["k_eval_pgh", "k_eval_p<", "k_fold_hot", "k_eval_custom"],
    "assembly": ["k_assemble"], "projection": ["k_project", "k_active_blocks"], "contact detection": ["k_contact", "k_bp_", "k_sweep", "k_table_bounds", "k_route"],
    "pattern (contact part)": ["k_keys", "k_heads", "k_slots", "k_rows", "k_chunk", "k_crow", "k_long_slots", "k_make_desc", "k_copy_u32"],
}
agg = {k: [0, 0.0, 0.0] for k in stages}
other = [0, 0.0]
for n, s, e in rows:
    d = (e - s) / 1e3
    nm = sh(n)
    for k, pats in stages.items():
        if any(p in nm for p in pats):
            a = agg[k]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
            break
    else:
        other[0] += 1; other[1] += d
print("W = %d ranks, %d Newton iterations and %d CG iterations in total (per rank: %d / %d)" % (W, n_newton, n_cg, n_newton // W, n_cg // W))
print("%-36s %8s %12s %10s %10s" % ("stage", "launches", "total ms", "avg us", "max us"))
for k, (c, t, m) in agg.items():
    if c: print("%-36s %8d %12.3f %10.2f %10.2f" % (k, c, t / 1e3, t / c, m))
print("%-36s %8d %12.3f" % ("everything else (sorts, copies, ...)", other[0], other[1] / 1e3))
cg = sum(agg[k][1] for k in agg if k.startswith("cg")) / max(n_cg, 1)
print("kernels per CG iteration and rank (average over ranks): %.2f us" % cg)
