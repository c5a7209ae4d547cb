#!/usr/bin/env python3
"""One rank of a multi-PROCESS sharded run (launched by tests/test_gpu_multiprocess.py through torch.distributed.run, all ranks on device 0 of
the test box): the IPC-window transport with the fused PCG iteration (one exposed exchange per iteration, kernels pushing into the peers'
windows and polling their own) against the single-rank engine — the cases of tests/test_gpu_sharded.py, which runs the ranks as threads of
one process over a copy kernel. Every rank computes the single-rank reference itself (small problems), asserts on its own results, and the
ranks compare the bits of what is replicated. Prints 'MP-OK <case>' on rank 0 when every rank passed.

    python -m torch.distributed.run ... tests/mp_sharded_worker.py stages <fixture> | newton | scene | fullsize
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    import torch
    import torch.distributed as dist

    from oracle import evaluator as ev
    from stark_amd import capi

    case = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo")

    def allgather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    comm = capi.IpcComm(0, rank, world, 64 << 20, allgather)
    opts = dict(kv.split("=") for kv in filter(None, os.environ.get("MP_ENGINE_OPTS", "").split(",")))
    failure = None
    try:
        if case == "stages":
            from gpu_util import engine_from_problem

            name = sys.argv[2]
            prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
            x = np.sin(0.37 * np.arange(man["ndofs"]))

            def stages(eng):
                E, g = eng.eval(capi.EVAL_P_G_H)
                Ep, _ = eng.eval(capi.EVAL_P)
                eng.eval(capi.EVAL_P_G_H)
                eng.assemble()
                y = eng.spmv(x)
                du0, info0 = eng.pcg(man["pcg"]["abs_tol"])
                eng.project(1e-10)
                yp = eng.spmv(x)
                eng.assemble()
                y2 = eng.spmv(x)
                du, info = eng.pcg(1e-8, 1e-6, 5000)
                # a third solve at a tolerance the first iterations already meet, and one that stops at the iteration cap
                du1, info1 = eng.pcg(1e-1, 1e-1, 5000)
                du2, info2 = eng.pcg(1e-300, 1e-300, 7)
                return dict(E=E, Ep=Ep, g=g, y=y, yp=yp, y2=y2, du=du, its=info.n_iterations, conv=info.converged, du0=du0, its0=info0.n_iterations, conv0=info0.converged,
                            du1=du1, its1=info1.n_iterations, conv1=info1.converged, du2=du2, its2=info2.n_iterations, conv2=info2.converged)

            single = engine_from_problem(prob, man)
            ref = stages(single)
            single.close()
            eng = engine_from_problem(prob, man)
            eng.dist_init_ipc(comm)
            for k, v in opts.items():
                eng.set_option(k, int(v))
            r = stages(eng)
            di = eng.dist_info()
            rows = di[0]
            eng.close()
            # which iteration ran: the fused one unless the engine was told otherwise
            assert (di[6], di[7]) == ((0, 4) if opts.get("no_fused_pcg") == "1" else (4, 0)), di
            gs = max(np.abs(ref["g"]).max(), 1e-300)
            assert abs(r["E"] - ref["E"]) <= 1e-12 * max(1.0, abs(ref["E"])) and abs(r["Ep"] - ref["Ep"]) <= 1e-12 * max(1.0, abs(ref["Ep"]))
            assert np.abs(r["g"] - ref["g"]).max() <= 1e-12 * gs
            for k in ("y", "y2", "yp"):
                assert np.abs(r[k] - ref[k]).max() <= 2e-6 * np.abs(ref[k]).max(), k
            # iteration counts within +-1 of the single-rank solve (the parity rule for the PCG), same verdicts, same solutions
            assert r["conv0"] == ref["conv0"] and abs(r["its0"] - ref["its0"]) <= 1, (r["its0"], ref["its0"])
            assert r["conv"] == ref["conv"] and abs(r["its"] - ref["its"]) <= 2, (r["its"], ref["its"])
            assert r["conv1"] == ref["conv1"] and abs(r["its1"] - ref["its1"]) <= 1, (r["its1"], ref["its1"])
            assert r["conv2"] == ref["conv2"] == 0 and r["its2"] == ref["its2"] == 7
            assert np.abs(r["du"] - ref["du"]).max() <= 1e-4 * max(np.abs(ref["du"]).max(), 1e-300)
            if "rb" not in name:
                assert np.abs(r["du0"] - ref["du0"]).max() <= 1e-3 * max(np.abs(ref["du0"]).max(), 1e-300)
                assert np.abs(r["du2"] - ref["du2"]).max() <= 1e-3 * max(np.abs(ref["du2"]).max(), 1e-300)
            summary = dict(rows=rows, key=[r["E"], r["Ep"], r["its"], r["its0"], r["its1"]] + [float(np.abs(r[k]).sum()) for k in ("g", "y", "yp", "du", "du0", "du1", "du2")])
        elif case == "newton":
            from gpu_util import engine_from_problem

            prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))

            def solve(eng, lazy):
                eng.set_option("lazy_hessians", lazy)
                res, st = eng.newton_solve()
                return res, st.newton_iterations, st.cg_iterations, eng.get_dofs()

            single = engine_from_problem(prob, man)
            ref = solve(single, 1)
            single.close()
            assert ref[0] == "Successful"
            key = []
            for lazy in (1, 0):
                eng = engine_from_problem(prob, man)
                eng.dist_init_ipc(comm)
                r = solve(eng, lazy)
                eng.close()
                assert r[0] == "Successful" and r[1] == ref[1] and abs(r[2] - ref[2]) <= ref[1] + 2, (r[:3], ref[:3])
                assert np.abs(r[3] - ref[3]).max() <= 1e-6 * max(np.abs(ref[3]).max(), 1e-300)
                key += [r[1], r[2], float(np.abs(r[3]).sum())]
            summary = dict(rows=1, key=key)
        elif case in ("scene", "fullsize"):
            from bench import build_scene
            from stark_amd import sim as S

            grid, steps = ((10, 10, 10), 4) if case == "scene" else ((44, 44, 43), 2)

            def run(sim):
                its, cg = [], 0
                for _ in range(steps):
                    assert sim.run_one_step()
                    st = sim.info().last_stats
                    its.append(st.newton_iterations)
                    cg += st.cg_iterations
                return its, cg, sim.points("x0")

            single = build_scene(S, *grid, 0)
            ref_its, ref_cg, ref_x = run(single)
            single.close()
            assert sum(ref_its) > 0
            sim = build_scene(S, *grid, 0)
            sim.set_dist_ipc(comm, rank, world)
            its, cg, x = run(sim)
            import ctypes as C
            info = (C.c_int64 * 8)()
            capi.lib().mistark_dist_info(sim.engine_handle(), info, 8)
            sim.close()
            assert info[6] > 0 and info[7] == 0, list(info)  # every linear solve took the fused iteration
            assert all(abs(a - b) <= 1 for a, b in zip(its, ref_its)), (its, ref_its)
            assert abs(cg - ref_cg) <= 0.05 * ref_cg + 5 * sum(ref_its), (cg, ref_cg)
            assert np.abs(x - ref_x).max() <= 1e-5
            summary = dict(rows=int(info[0]), key=[its, cg, float(np.abs(x).sum())], ref=[ref_its, ref_cg])
        else:
            raise SystemExit("unknown case " + case)
    except BaseException as e:  # noqa: BLE001
        failure = "rank %d: %r" % (rank, e)
        summary = None
    # (a rank that failed leaves the others waiting in an exchange until its time-out: collect whatever every rank has to say)
    everything = allgather((failure, summary))
    comm.close()
    dist.destroy_process_group()
    errs = [f for f, _ in everything if f]
    if errs:
        if rank == 0:
            print("MP-FAILED " + " | ".join(errs))
        sys.exit(1)
    sums = [s for _, s in everything]
    assert all(s["key"] == sums[0]["key"] for s in sums), "ranks disagree on replicated values: %r" % sums  # identical bits everywhere
    if rank == 0:
        print("MP-OK %s %s" % (" ".join(sys.argv[1:]), json.dumps(dict(rows=[s["rows"] for s in sums], key=sums[0]["key"][:5], ref=sums[0].get("ref")))))


if __name__ == "__main__":
    main()
