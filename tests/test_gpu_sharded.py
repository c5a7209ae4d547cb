"""GPU: the sharded path (block rows partitioned over the ranks, every rank evaluates the elements touching its rows and assembles and
solves its rows of the system) run with several engine contexts in ONE process on one MI355X (in-process all-gather, one host
thread per rank): 2, 3 and 8 ranks must reproduce the single-rank results, and every rank must hold identical bits of everything that is
replicated. The RCCL transport differs only in who carries the all-gather (tests/test_dist_cpu.py covers the partition, halo and
fused-dot arithmetic with gloo on CPU)."""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import evaluator as ev  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run_ranks(world, fn):
    """fn(rank) in one thread per rank; returns the list of results (re-raises the first exception)."""
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("name", ["tetbeam_full_4x1x1", "tetbeam_eo_4x1x1_big", "cloth_shells_6", "contactmix_t1", "rbchain"])
def test_sharded_stages_equal_single_rank(name, world):
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    x = np.sin(0.37 * np.arange(man["ndofs"]))

    def stages(eng):
        E, g = eng.eval(capi.EVAL_P_G_H)
        Ep, _ = eng.eval(capi.EVAL_P)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        y = eng.spmv(x)
        du0, info0 = eng.pcg(man["pcg"]["abs_tol"])
        eng.project(1e-10)                       # every element Hessian to PSD; the assembled matrix is patched in place
        yp = eng.spmv(x)                         # (an interface element is projected by both sides: the same numbers)
        eng.assemble()                           # ... and must equal the matrix assembled from the projected Hessians
        y2 = eng.spmv(x)
        du, info = eng.pcg(1e-8, 1e-6, 5000)
        return dict(E=E, Ep=Ep, g=g, y=y, yp=yp, y2=y2, du=du, its=info.n_iterations, conv=info.converged, du0=du0, its0=info0.n_iterations, conv0=info0.converged)

    single = engine_from_problem(prob, man)
    ref = stages(single)
    single.close()
    L = capi.lib()
    group = L.mistark_local_group_create(world)

    def rank_fn(r):
        eng = engine_from_problem(prob, man)
        eng.dist_init_local(group, r)
        res = stages(eng)
        res["info"] = eng.dist_info()
        res["owner"] = eng.dist_row_owner()
        eng.close()
        return res

    res = run_ranks(world, rank_fn)
    L.mistark_local_group_destroy(group)
    gs = max(np.abs(ref["g"]).max(), 1e-300)
    # the ranks' rows tile the block rows; every rank sees the same owner map
    assert sum(r["info"][0] for r in res) == man["ndofs"] // 3
    assert all((r["owner"] == res[0]["owner"]).all() for r in res)
    for r in res:
        assert abs(r["E"] - ref["E"]) <= 1e-12 * max(1.0, abs(ref["E"])) and abs(r["Ep"] - ref["Ep"]) <= 1e-12 * max(1.0, abs(ref["Ep"]))
        assert np.abs(r["g"] - ref["g"]).max() <= 1e-12 * gs
        # a rank sums the contributions of a block in the order of ITS element list: last-bit differences in the float matrix
        assert np.abs(r["y"] - ref["y"]).max() <= 2e-6 * np.abs(ref["y"]).max()
        assert np.abs(r["y2"] - ref["y2"]).max() <= 2e-6 * np.abs(ref["y2"]).max()
        assert np.abs(r["yp"] - ref["yp"]).max() <= 2e-6 * np.abs(ref["yp"]).max()
        assert np.abs(r["yp"] - r["y2"]).max() <= 2e-6 * np.abs(r["y2"]).max()
        # the row-sharded PCG: iteration counts within +-1 of the single-rank solve (and of the reference's, which the single-rank
        # solve is pinned to in test_gpu_parity.py), same convergence verdict, same solution
        assert r["conv0"] == ref["conv0"] and abs(r["its0"] - ref["its0"]) <= 1
        assert r["conv"] == ref["conv"] and abs(r["its"] - ref["its"]) <= 2   # (a tighter solve than the reference ever asks for, on the projected matrix)
        assert np.abs(r["du"] - ref["du"]).max() <= 1e-4 * max(np.abs(ref["du"]).max(), 1e-300)
        if "rb" not in name:  # (stiff rigid-body constraint systems amplify the float rounding of the matrix beyond this)
            assert np.abs(r["du0"] - ref["du0"]).max() <= 1e-3 * max(np.abs(ref["du0"]).max(), 1e-300)
    # every rank holds the SAME bits of what is replicated
    for r in res[1:]:
        assert r["E"] == res[0]["E"] and r["Ep"] == res[0]["Ep"] and (r["g"] == res[0]["g"]).all() and (r["y"] == res[0]["y"]).all()
        assert (r["yp"] == res[0]["yp"]).all() and (r["du"] == res[0]["du"]).all() and (r["du0"] == res[0]["du0"]).all() and r["its"] == res[0]["its"]


@pytest.mark.parametrize("rtc", [0, 1])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_user_defined_potentials_count_interface_energies_once(world, rtc):
    """Every potential of the beam registered as a USER-DEFINED potential (its symx::Sequence: the device interpreter, rtc = 0, or the kernels
    emitted from it, rtc = 1) on sharded ranks: interface elements are evaluated by every rank that owns one of their rows, and their energy
    must count on one of them only (energy_here) — round 5 found the custom kernels adding it on both sides (energy-only evaluations were right,
    evaluations with gradient / Hessian counted interface elements twice: a line search comparing the two kinds would have been off)."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_full_4x1x1.npz"))

    def stages(eng):
        eng.set_option("custom_rtc", rtc)
        E, g = eng.eval(capi.EVAL_P_G_H)
        Ep, _ = eng.eval(capi.EVAL_P)
        Eg, gg = eng.eval(capi.EVAL_P_G)
        return dict(E=E, Ep=Ep, Eg=Eg, g=g, gg=gg)

    single = engine_from_problem(prob, man, custom_ops=z)
    ref = stages(single)
    single.close()
    assert abs(ref["E"] - man["E"]) <= 1e-12 * max(1.0, abs(man["E"]))
    L = capi.lib()
    group = L.mistark_local_group_create(world)

    def rank_fn(r):
        eng = engine_from_problem(prob, man, custom_ops=z)
        eng.dist_init_local(group, r)
        nbr = eng.ndofs // 3
        eng.dist_set_row_owner(np.arange(nbr, dtype=np.int32) * world // nbr)   # (the partitioner keeps a problem this small on one rank)
        res = stages(eng)
        info = (C.c_int64 * 15)()
        assert L.mistark_dist_info(eng.h, info, 15) == 0
        res["info"] = list(info)
        eng.close()
        return res

    res = run_ranks(world, rank_fn)
    L.mistark_local_group_destroy(group)
    assert all(r["info"][14] > 0 for r in res)
    assert sum(r["info"][14] for r in res) > res[0]["info"][13]   # (interface elements exist: evaluated by more than one rank)
    for r in res:
        for k in ("E", "Ep", "Eg"):
            assert abs(r[k] - ref[k]) <= 1e-12 * max(1.0, abs(ref[k])), (k, r[k], ref[k])
        assert np.abs(r["g"] - ref["g"]).max() <= 1e-12 * np.abs(ref["g"]).max()
        assert np.abs(r["gg"] - ref["gg"]).max() <= 1e-12 * np.abs(ref["gg"]).max()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_newton_lazy_and_full(world):
    """A Newton solve of the tet beam on sharded ranks, on the lazy path (float upper-triangle pool, the default inside the Newton loop) and
    on the full double pool: same iteration counts and iterate as one rank."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))

    def solve(eng, lazy):
        eng.set_option("lazy_hessians", lazy)
        res, st = eng.newton_solve()
        return res, st.newton_iterations, st.cg_iterations, eng.get_dofs()

    single = engine_from_problem(prob, man)
    ref = solve(single, 1)
    single.close()
    assert ref[0] == "Successful"
    L = capi.lib()
    for lazy in (1, 0):
        group = L.mistark_local_group_create(world)

        def rank_fn(r):
            eng = engine_from_problem(prob, man)
            eng.dist_init_local(group, r)
            out = solve(eng, lazy)
            eng.close()
            return out

        res = run_ranks(world, rank_fn)
        L.mistark_local_group_destroy(group)
        for r in res:
            assert r[0] == "Successful" and r[1] == ref[1] and abs(r[2] - ref[2]) <= ref[1] + 2
            assert np.abs(r[3] - ref[3]).max() <= 1e-6 * max(np.abs(ref[3]).max(), 1e-300)
        assert all((r[3] == res[0][3]).all() for r in res)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_contact_scene_trajectory(world):
    """The block-on-box contact scene (device detection, friction, rigid body) stepped by sharded ranks: same Newton iteration counts
    and end state as the reference trajectory, identical on all ranks."""
    from stark_amd import capi
    from stark_amd import sim as S

    z = np.load(os.path.join(GOLDEN, "traj_blockbox_3.npz"))
    traj = json.loads(bytes(z["traj_json"]).decode())
    sc = traj["scene"]
    L = capi.lib()
    group = L.mistark_local_group_create(world)

    def rank_fn(r):
        st = S.default_settings()
        st.init_frictional_contact = 1
        sim = S.Simulation(st)
        gp = S.contact_global_params()
        gp.default_contact_thickness = sc["thickness"]
        gp.min_contact_stiffness = sc["kmin"]
        sim.set_contact_global_params(gp)
        rb = sim.add_rigid_box("box", 1.0, (sc["bx"], sc["bx"], sc["bz"]))
        sim.rb_add_constraint("fix", rb)
        Lb = sc["L"]
        ps = sim.add_volume_grid("block", (0.0, 0.0, 0.5 * sc["bz"] + sc["gap"] + 0.5 * Lb), (Lb, Lb, Lb), (sc["nx"], sc["ny"], sc["nz"]), S.soft_rubber())
        sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), sc["mu"])
        sim.set_dist_local(group, r, world)
        its = []
        for _ in traj["steps"]:
            assert sim.run_one_step()
            its.append(sim.info().last_stats.newton_iterations)
        x = sim.points("x0")
        sim.close()
        return its, x

    res = run_ranks(world, rank_fn)
    L.mistark_local_group_destroy(group)
    for its, x in res:
        assert its == traj["newton_iterations"]
        assert np.abs(x - z["x_end"]).max() <= 1e-4 * np.abs(z["x_end"]).max()
    assert all((r[1] == res[0][1]).all() for r in res)


def test_rccl_transport_single_rank_roundtrip():
    """The RCCL entry points (dlopen'ed librccl: unique id, communicator, ncclAllGather on the engine's stream) on the one GPU of the test
    box: a one-rank communicator must return its input."""
    import ctypes as C

    import stark_amd

    eng = stark_amd.Engine(0)
    x = np.linspace(-3.0, 5.0, 1000)
    y = x.copy()
    assert eng.L.mistark_dist_rccl_selftest(eng.h, y.ctypes.data, len(y)) == 0, eng.L.mistark_last_error(eng.h)
    assert (x == y).all()
    buf = C.create_string_buffer(128)
    assert eng.L.mistark_dist_unique_id(buf) == 0 and any(b != 0 for b in buf.raw)
    eng.close()


def test_rccl_allreduce_leg_single_rank():
    """The RCCL leg of an N > 1 bench line (mistark_rccl_allreduce_bench: ncclCommInitRank, ncclAllReduce(f64, sum) of ndofs and of 3 doubles,
    results checked against the closed-form sum, ncclCommCount) with a one-rank communicator on the test box's GPU — the same entry points,
    argument order and enum values the N-rank call uses."""
    import ctypes as C

    from stark_amd import capi

    L = capi.lib()
    uid = C.create_string_buffer(128)
    assert L.mistark_dist_unique_id(uid) == 0
    out = (C.c_double * 4)()
    err = C.create_string_buffer(512)
    assert L.mistark_rccl_allreduce_bench(0, 0, 1, uid.raw, 517050, 50, out, err, 512) == 0, err.value
    assert out[0] == 1.0 and 0.0 < out[1] < 1e5 and 0.0 < out[2] < 1e5 and out[3] >= 0.0
    # a rank outside the world is an argument error with a message, not a crash
    assert L.mistark_rccl_allreduce_bench(0, 2, 1, uid.raw, 8, 1, out, err, 512) != 0 and b"bad arguments" in err.value


@pytest.mark.parametrize("world", [3, 8])
def test_sharded_contact_scene_real_partition(world):
    """A block large enough for a real partition (2 331 block rows, recursive coordinate bisection from the scene's rest positions) on a
    rigid box with frictional contact: contact potentials (device-side tables, evaluated by every rank, the collision vertices shared as
    ghosts), rigid-body rows on the last rank. Against the same scene on one rank: Newton iteration counts per time step (+-1) and the end
    state; identical bits on all ranks; every rank owns rows."""
    from bench import build_scene
    from stark_amd import capi
    from stark_amd import sim as S

    def run(sim):
        its = []
        for _ in range(4):
            assert sim.run_one_step()
            its.append(sim.info().last_stats.newton_iterations)
        return its, sim.points("x0")

    single = build_scene(S, 10, 10, 10, 0)
    ref_its, ref_x = run(single)
    single.close()
    assert sum(ref_its) > 0
    L = capi.lib()
    group = L.mistark_local_group_create(world)

    def rank_fn(r):
        import ctypes as C
        sim = build_scene(S, 10, 10, 10, 0)
        sim.set_dist_local(group, r, world)
        out = run(sim)
        info = (C.c_int64 * 6)()
        L.mistark_dist_info(sim.engine_handle(), info, 6)
        sim.close()
        return out + (list(info),)

    res = run_ranks(world, rank_fn)
    L.mistark_local_group_destroy(group)
    assert sum(r[2][0] for r in res) == 11 ** 3 + 10 ** 3 + 2 and all(r[2][0] > 0 for r in res)
    for its, x, _ in res:
        assert all(abs(a - b) <= 1 for a, b in zip(its, ref_its)), (its, ref_its)
        assert np.abs(x - ref_x).max() <= 1e-5
    assert all((r[1] == res[0][1]).all() and r[0] == res[0][0] for r in res)


def test_sharded_full_size_headline_scene():
    """configs[3] at full size (998 976 tets, 517 050 DoF, frictional contact) on 4 ranks: two time steps against the same scene on one
    rank — Newton iteration counts (+-1), CG iterations per solve, end state; identical bits on all ranks; the partition is balanced."""
    import ctypes as C

    from bench import build_scene
    from stark_amd import capi
    from stark_amd import sim as S

    world, steps = 4, 2

    def run(sim):
        its, cg = [], 0
        for _ in range(steps):
            assert sim.run_one_step()
            st = sim.info().last_stats
            its.append(st.newton_iterations)
            cg += st.cg_iterations
        return its, cg, sim.points("x0")

    single = build_scene(S, 44, 44, 43, 0)
    ref_its, ref_cg, ref_x = run(single)
    single.close()
    assert sum(ref_its) > 0
    L = capi.lib()
    group = L.mistark_local_group_create(world)

    def rank_fn(r):
        sim = build_scene(S, 44, 44, 43, 0)
        sim.set_dist_local(group, r, world)
        out = run(sim)
        info = (C.c_int64 * 6)()
        L.mistark_dist_info(sim.engine_handle(), info, 6)
        sim.close()
        return out + (list(info),)

    res = run_ranks(world, rank_fn)
    L.mistark_local_group_destroy(group)
    rows = [r[3][0] for r in res]
    assert sum(rows) == 517050 // 3 and max(rows) - min(rows) <= 0.02 * max(rows)
    for its, cg, x, _ in res:
        assert all(abs(a - b) <= 1 for a, b in zip(its, ref_its)), (its, ref_its)
        assert abs(cg - ref_cg) <= 0.05 * ref_cg + 5 * sum(ref_its)
        assert np.abs(x - ref_x).max() <= 1e-5
    assert all((r[2] == res[0][2]).all() and r[0] == res[0][0] and r[1] == res[0][1] for r in res)


def test_sharded_pcg_samples_its_spmv_on_the_device_clock_too():
    """bench.py --gpus N reads the roofline figure of rank 0's SpMV launches the same way as on one GPU: the sharded PCG brackets every 32nd
    launch with HIP events AND lets its workgroups stamp the device clock. Two in-process ranks, solves forced to the iteration cap: the
    sampling changes no bit of the solution, both figures are there, the clock's below the event bracket."""
    import ctypes as C

    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))
    L = capi.lib()
    group = L.mistark_local_group_create(2)

    def rank_fn(r):
        eng = engine_from_problem(prob, man)
        eng.dist_init_local(group, r)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        du0, info0 = eng.pcg(1e-300, 1e-300, 100)
        eng.spmv_timing(reset=1)
        du, info = eng.pcg(1e-300, 1e-300, 100)
        ms, n = C.c_double(), C.c_int64()
        assert eng.L.mistark_spmv_device_clock(eng.h, C.byref(ms), C.byref(n)) == 0
        ev_ms, ev_n, _ = eng.spmv_timing(reset=-1)
        eng.close()
        return (du0 == du).all(), info0.n_iterations, info.n_iterations, ms.value, n.value, ev_ms, ev_n

    res = run_ranks(2, rank_fn)
    L.mistark_local_group_destroy(group)
    for same, it0, it1, clk_ms, clk_n, ev_ms, ev_n in res:
        assert same and it0 == it1 >= 64
        assert clk_n == ev_n == it1 // 32
        assert 0.0 < clk_ms < ev_ms < 1.0
