"""Developer tool: configs[3] on W ranks inside ONE process on one MI355X (in-process all-gather, one shared stream: the ranks' kernels run one
after the other, so a kernel trace of this run gives every rank's kernels their solo durations). Prints what each rank holds and the
aggregate rate; run it under `rocprofv3 --kernel-trace --stats` and feed the database to profiles/shard_rocpd.py for the per-rank stage
times the multi-GPU cost model in DESIGN.md uses.   usage: python tools/shard_model.py W [newton_steps] [nx,ny,nz]"""
import sys
import threading
import time

sys.path.insert(0, ".")
from bench import build_scene, run_newton_steps
from stark_amd import capi
from stark_amd import sim as S

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nx, ny, nz = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "44,44,43").split(",")]
L = capi.lib()
group = L.mistark_local_group_create(W) if W > 1 else None
out = [None] * W
err = [None] * W


def work(r):
    try:
        import os
        sim = build_scene(S, nx, ny, nz, 0, scene=os.environ.get("SCENE", "contact"))
        if W > 1:
            sim.set_dist_local(group, r, W)
        run_newton_steps(sim, S, capi, 4)
        t0 = time.perf_counter()
        newton, n_ls, n_cg, t_ls = run_newton_steps(sim, S, capi, steps)
        dt = time.perf_counter() - t0
        import ctypes as C
        info = (C.c_int64 * 6)()
        L.mistark_dist_info(sim.engine_handle(), info, 6)
        out[r] = dict(newton=newton, n_ls=n_ls, n_cg=n_cg, dt=dt, info=list(info))
        sim.close()
    except BaseException as e:  # noqa: BLE001
        err[r] = e


th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
for t in th:
    t.start()
for t in th:
    t.join()
for e in err:
    if e is not None:
        raise e
for r, o in enumerate(out):
    print("rank %d: rows %d ghosts %d send rows %d elements evaluated %d matrix blocks %d + %d | %d Newton its, %d solves, %d CG its, %.3f s" % (
        r, *o["info"], o["newton"], o["n_ls"], o["n_cg"], o["dt"]))
print("TOTALS %d %d" % (sum(o["newton"] for o in out) + 4 * W, sum(o["n_cg"] for o in out)))
print("W=%d: %.1f Newton-steps/s with all ranks serialised on one GPU (not a scaling figure)" % (W, out[0]["newton"] / max(o["dt"] for o in out)))
