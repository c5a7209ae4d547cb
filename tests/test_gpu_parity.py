"""GPU parity (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI, against
 (a) the golden vectors of the unmodified reference (tests/golden/*.npz) and (b) the CPU oracle on the same inputs.

Tolerances: element P/g/H 1e-11 relative; E/grad sums 1e-12 * sum|terms| (atomics: order-dependent);
projected Hessians 1e-9 relative Frobenius; BSR blocks: float eps * contributions; PCG: iteration count +-1 and solution
within 10*rel_tol; Newton on contact-free scenes: identical iteration counts, evaluation points within 1e-6 relative.
"""
import glob
import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import evaluator as ev

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fixture_list import stage_dumps  # noqa: E402

DUMPS = stage_dumps()
ELEMENT_TOL = {"EnergyDiscreteShells": 1e-8}  # ill-conditioned acos near 1, see tests/test_oracle_golden.py


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("variant", ["default", "generic_atomic", "split_matrix", "split_matrix_atomic", "custom_ops", "custom_rtc"])
@pytest.mark.parametrize("path", DUMPS, ids=[os.path.basename(p)[:-4] for p in DUMPS])
def test_stages_match_reference(path, variant):
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(path)
    # custom_ops: no compiled kernel is used; every potential runs the reference's symx::Sequence through the device interpreter
    # custom_rtc: the same sequences EMITTED as HIP source and compiled by hipRTC (what a user-defined potential gets by default; the
    # interpreter is its fallback): every stage output below must hold for both
    eng = engine_from_problem(prob, man, custom_ops=z if variant.startswith("custom_") else None)
    if variant.startswith("custom_"):
        eng.set_option("custom_rtc", 1 if variant == "custom_rtc" else 0)
    assert eng.ndofs == man["ndofs"]
    # closed-form kernels for every contact / friction table (by default the table's size decides: kernels.hip closed_contact_pays)
    eng.set_option("contact_closed_min_lanes", 0)
    if variant == "generic_atomic":  # generic hyper-dual kernels for every potential + atomic scatter assembly
        eng.set_option("force_generic", 1)
        eng.set_option("atomic_assembly", 1)
    if variant.startswith("split_matrix"):
        # A = A_static + A_dynamic: contact/friction potentials (or, without contacts, every second potential) go to the
        # dynamic part that the engine re-patterns alone when contact sets change
        names = {pi: prob.potentials[pi].name for pi in eng.pot_ids}
        dyn = [pi for pi, n in names.items() if "contact" in n or "friction" in n]
        if not dyn:
            dyn = sorted(names)[1::2]
        for pi in dyn:
            eng.set_dynamic(eng.pot_ids[pi], True)
        if variant.endswith("atomic"):
            eng.set_option("atomic_assembly", 1)

    # ---- evaluation -----------------------------------------------------------------------------------------------
    E, grad = eng.eval(capi.EVAL_P_G_H)
    scale = sum(abs(p.get("E", 0.0)) for p in man["potentials"])
    assert abs(E - man["E"]) <= 1e-12 * max(1.0, scale)
    gtol = max([1e-11] + [ELEMENT_TOL.get(p["name"], 0) for p in man["potentials"] if p["n_elem"] > 0])
    assert _rel(grad, z["grad"]) < gtol
    E_only, _ = eng.eval(capi.EVAL_P)
    assert abs(E_only - man["E"]) <= 1e-12 * max(1.0, scale)
    E_pg, grad_pg = eng.eval(capi.EVAL_P_G)
    assert abs(E_pg - man["E"]) <= 1e-12 * max(1.0, scale)
    assert _rel(grad_pg, z["grad"]) < gtol
    eng.eval(capi.EVAL_P_G_H)
    if variant.startswith("custom_"):   # the path under test did run: emitted kernels launched (or, for the interpreter variant, none built)
        assert (eng.counter("rtc_launches") > 0) == (variant == "custom_rtc"), (eng.counter("rtc_builds"), eng.counter("rtc_launches"))

    for pi, pid in eng.pot_ids.items():
        ref = man["potentials"][pi]
        n_elem = ref["n_elem"]
        H, rows = eng.element_hessians(pid, n_elem)
        assert (rows == z["p%d_hrows" % pi]).all()
        tol = ELEMENT_TOL.get(ref["name"], 1e-11)
        assert _rel(H, z["p%d_hvals" % pi]) < tol, ref["name"]
        Ee = eng.element_energies(pid, n_elem)
        assert abs(Ee.sum() - ref["E"]) <= 1e-11 * max(1.0, np.abs(Ee).sum())

    # ---- assembly (unprojected) ------------------------------------------------------------------------------------
    eng.assemble()
    row_ptr, cols, vals = eng.get_bsr()
    nbr = len(row_ptr) - 1
    S = sp.bsr_matrix((vals.astype(np.float64), cols, row_ptr), shape=(3 * nbr, 3 * nbr)).tocsr()
    Sref = sp.coo_matrix((z["A_vals"], (z["A_rows"], z["A_cols"])), shape=S.shape).tocsr()
    assert abs(S - Sref).max() <= 64 * np.finfo(np.float32).eps * abs(Sref).max()
    # same block pattern (the reference drops nothing here: every block row has its diagonal block)
    assert 9 * len(cols) == man["nnz_scalar"]

    x = np.sin(0.37 * np.arange(prob.ndofs))
    assert _rel(eng.spmv(x), z["spmv_y"]) < 1e-5
    assert _rel(eng.apply_preconditioner(x), z["prec_z"]) < 1e-4

    # ---- PCG with the Newton forcing tolerance ----------------------------------------------------------------------
    xs, info = eng.pcg(man["pcg"]["abs_tol"])
    assert bool(info.converged) == bool(man["pcg"]["converged"])
    assert abs(info.n_iterations - man["pcg"]["iterations"]) <= 1
    if info.n_iterations == man["pcg"]["iterations"] and info.converged:
        # the solution must solve the REFERENCE's system as well as the reference's own iterate does. (A direct comparison
        # of the iterates is only meaningful for well-conditioned systems: the matrices agree to float rounding, which
        # the condition number of stiff rigid-body constraint systems amplifies beyond 10*rel_tol.)
        b = -z["grad"]
        res_ours = np.linalg.norm(b - Sref @ xs) / np.linalg.norm(b)
        res_ref = np.linalg.norm(b - Sref @ z["pcg_x"]) / np.linalg.norm(b)
        assert res_ours <= 3.0 * res_ref + 1e-6
        if "rb" not in os.path.basename(path):
            assert _rel(xs, z["pcg_x"]) < 1e-3
    # and against the oracle's PCG on the oracle's matrix
    _, grad_o, outs = ev.evaluate_all(prob)
    A = ev.assemble(outs, prob.ndofs)
    xo, info_o = ev.solve_pcg(A, -grad_o, man["pcg"]["abs_tol"])
    assert abs(info.n_iterations - info_o.n_iterations) <= 1

    # ---- projection of every element Hessian ----------------------------------------------------------------------------
    n_proj, n_changed = eng.project(1e-10, False, None)
    assert n_proj == man["n_hessians"]
    changed_ref = 0
    for pi, pid in eng.pot_ids.items():
        ref = man["potentials"][pi]
        H, _ = eng.element_hessians(pid, ref["n_elem"])
        Hp = z["p%d_hvals_proj" % pi]
        tol = max(1e-9, 10 * ELEMENT_TOL.get(ref["name"], 0))
        num = np.sqrt(((H - Hp) ** 2).sum(axis=(1, 2)))
        den = np.sqrt((Hp ** 2).sum(axis=(1, 2)))
        assert (num <= tol * np.maximum(den, 1e-300)).all(), ref["name"]
        changed_ref += int((np.abs(Hp - z["p%d_hvals" % pi]).max(axis=(1, 2)) > 0).sum())
    assert n_changed == changed_ref
    # projecting again is a no-op (idempotence / is_projected bookkeeping)
    assert eng.project(1e-10, False, None) == (0, 0)
    eng.close()


@pytest.mark.parametrize("proj_variant", [1, 2, 4, 7, 10])
@pytest.mark.parametrize("name", ["tetbeam_eo_4x1x1_big", "contactmix_t1", "rbchain", "cloth_shells_6"])
def test_projection_variants_agree(name, proj_variant):
    """The PSD projection has a register-resident kernel (default), the earlier LDS kernel (bit 1), per-potential launches instead of
    the batched one (bit 2; tets and membrane triangles then take the reduced-matrix kernel for translation-invariant elements, unless
    bit 8 is set) and IEEE division/square root for the rotation angles (bit 4): every combination gives the reference's
    projected Hessians and the same "changed" count, and agrees with the default path to 1e-10 (the Jacobi sweeps stop at
    off(A) <= 1e-12 ||A||; the variants' rotations differ in rounding, so they stop at different points below that)."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    path = os.path.join(GOLDEN, name + ".npz")
    prob, man, z = ev.load_fixture(path)
    out = []
    for variant in (0, proj_variant):
        eng = engine_from_problem(prob, man)
        eng.set_option("proj_variant", variant)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()                              # the deltas also go into the assembled matrix
        n_proj, n_changed = eng.project(1e-10, False, None)
        H = {pi: eng.element_hessians(pid, man["potentials"][pi]["n_elem"])[0] for pi, pid in eng.pot_ids.items()}
        out.append((n_proj, n_changed, H, eng.get_bsr()[2]))
        eng.close()
    (np0, nc0, H0, v0), (np1, nc1, H1, v1) = out
    assert np0 == np1 == man["n_hessians"] and nc0 == nc1
    for pi in H0:
        Hp = z["p%d_hvals_proj" % pi]
        den = np.maximum(np.sqrt((Hp ** 2).sum(axis=(1, 2))), 1e-300)
        assert (np.sqrt(((H0[pi] - H1[pi]) ** 2).sum(axis=(1, 2))) <= 1e-10 * den).all()
        tol = max(1e-9, 10 * ELEMENT_TOL.get(man["potentials"][pi]["name"], 0))
        assert (np.sqrt(((H1[pi] - Hp) ** 2).sum(axis=(1, 2))) <= tol * den).all()
    assert np.abs(v0 - v1).max() <= 4e-6 * np.abs(v0).max()   # float atomics of the deltas


# (rigid-body trajectories need the rigid-body state update and constraint hardening of the host layer: tests/test_gpu_scene.py)
TRAJ = [p for p in DUMPS if os.path.basename(p).startswith("traj_") and "rb" not in os.path.basename(p) and "box" not in os.path.basename(p) and "attach" not in os.path.basename(p) and "llt" not in os.path.basename(p)]  # (rigid-body and contact trajectories: scene tests)


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[:-4] for p in TRAJ])
def test_newton_trajectory_matches_reference(path):
    """Contact-free scenes: same Newton iteration counts as the reference, evaluation points within 1e-6 relative."""
    import ctypes as C

    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(path)
    traj = json.loads(bytes(z["traj_json"]).decode())
    eng = engine_from_problem(prob, man)
    iv1, ix0, iv0 = ev.point_state_arrays(prob)
    a_v1, a_x0, a_v0 = eng.array_ids[(iv1, 3)], eng.array_ids[(ix0, 3)], eng.array_ids[(iv0, 3)]
    pts = []

    def before_eval(_):
        u = eng.get_dofs()
        if not pts or np.abs(pts[-1] - u).max() > 0:
            pts.append(u)

    cb = capi.NewtonCallbacks()
    cb.before_energy_evaluation = capi.VOIDCB(before_eval)
    newton_its, cg_total, cg_series, ls_bt_series = [], 0, [], []
    for s in range(len(traj["steps"])):
        eng.fill(a_v1, 0.0)                      # before_time_step: v1 <- 0 (PointDynamics.cpp:58-62)
        pts_before = len(pts)
        res, st = eng.newton_solve(None, cb)
        assert res == "Successful"
        newton_its.append(st.newton_iterations)
        cg_total += st.cg_iterations
        log = eng.newton_iteration_log()
        assert sum(r.linear_solves for r in log) == st.n_linear_solves and sum(r.cg_iterations_all for r in log) == st.cg_iterations
        assert sum(r.ls_bt for r in log) == st.ls_bt_iterations and sum(r.n_hessians for r in log if r.logged) == st.n_hessians
        cg_series += [r.cg_iterations_last for r in log if r.logged]
        ls_bt_series += [r.ls_bt for r in log if r.line_search]
        eng.axpby(a_x0, 1.0, a_x0, prob.dt, a_v1)  # on_time_step_accepted: x0 += dt v1; v0 = v1 (PointDynamics.cpp:64-78)
        eng.axpby(a_v0, 1.0, a_v1)
        # a duplicate at a step boundary (v1 = 0 again) must be kept, as in the reference trace
        assert len(pts) > pts_before
    assert newton_its == traj["newton_iterations"]
    assert abs(cg_total - sum(traj["cg_iterations"])) <= max(2, 0.05 * sum(traj["cg_iterations"]))
    # the reference's Logger series, entry by entry: CG iterations of every Newton iteration's last solve, Armijo backtracks of every line search
    assert len(cg_series) == len(traj["cg_iterations"]) and all(abs(a - b) <= 1 for a, b in zip(cg_series, traj["cg_iterations"])), (cg_series, traj["cg_iterations"])
    assert ls_bt_series == traj["ls_bt"]
    ref = z["iterates"]
    assert len(pts) == ref.shape[0]
    for a, b in zip(pts, ref):
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max())
    eng.download(a_x0)
    assert np.abs(eng.host_arrays[ix0] - z["x_end"]).max() <= 1e-6 * np.abs(z["x_end"]).max()
    eng.close()


@pytest.mark.parametrize("name", ["contactmix_t1", "tetbeam_full_4x1x1"])
def test_dynamic_connectivity_update_equals_fresh_build(name):
    """Changing the connectivity of dynamic potentials re-patterns only the dynamic matrix part; the result must be
    bit-identical to an engine built from scratch with the new connectivity (deterministic gather assembly)."""
    import copy

    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    eng = engine_from_problem(prob, man)
    names = {pi: prob.potentials[pi].name for pi in eng.pot_ids}
    dyn = [pi for pi, n in names.items() if "contact" in n or "friction" in n] or sorted(names)[1::2]
    for pi in dyn:
        eng.set_dynamic(eng.pot_ids[pi], True)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    x = np.random.default_rng(5).standard_normal(eng.ndofs)
    y_full = eng.spmv(x)
    prob2 = copy.deepcopy(prob)
    for k, pi in enumerate(dyn):
        c = prob.potentials[pi].conn
        sub = c[::2] if k % 2 == 0 else c[:0]      # half of the rows / no rows at all
        if name.startswith("tetbeam") and k % 2 == 1:
            sub = c[1::3]
        prob2.potentials[pi].conn = np.ascontiguousarray(sub)
        eng.update_connectivity(eng.pot_ids[pi], sub)
    E1, g1 = eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    rp1, c1, v1 = eng.get_bsr()
    y1 = eng.spmv(x)
    z1 = eng.apply_preconditioner(x)
    assert np.abs(y1 - y_full).max() > 0
    fresh = stark_amd_engine_keep_empty(prob2)
    E2, g2 = fresh.eval(capi.EVAL_P_G_H)
    fresh.assemble()
    rp2, c2, v2 = fresh.get_bsr()
    assert E1 == E2 or abs(E1 - E2) <= 1e-13 * abs(E2)
    assert np.abs(g1 - g2).max() <= 1e-12 * max(1.0, np.abs(g2).max())
    # same matrix: compare as scipy BSR (the split engine may hold structurally-present zero blocks the fresh one lacks)
    import scipy.sparse as sp

    n = eng.ndofs // 3
    A1 = sp.bsr_matrix((v1.astype(np.float64), c1, rp1), shape=(3 * n, 3 * n)).tocsr()
    A2 = sp.bsr_matrix((v2.astype(np.float64), c2, rp2), shape=(3 * n, 3 * n)).tocsr()
    d = abs(A1 - A2)
    assert d.max() <= 2e-7 * abs(A2).max()   # float sums split over two parts round differently in the last bit
    y2 = fresh.spmv(x)
    assert np.abs(y1 - y2).max() <= 1e-6 * np.abs(y2).max()
    z2 = fresh.apply_preconditioner(x)
    assert np.abs(z1 - z2).max() <= 1e-5 * np.abs(z2).max()
    du1, info1 = eng.pcg(1e-10, 1e-8, 2000)
    du2, info2 = fresh.pcg(1e-10, 1e-8, 2000)
    assert info1.converged == info2.converged and abs(info1.n_iterations - info2.n_iterations) <= 2
    assert np.abs(du1 - du2).max() <= 1e-4 * max(np.abs(du2).max(), 1e-300)
    eng.close()
    fresh.close()


def stark_amd_engine_keep_empty(prob):
    from gpu_util import engine_from_problem

    return engine_from_problem(prob)


@pytest.mark.parametrize("name", ["tetbeam_eo_4x1x1", "tetbeam_eo_4x1x1_big", "tetbeam_full_4x1x1", "tetbeam_softrubber_6x2x2", "contactmix_t1"])
def test_lazy_hessian_path_matches_full(name):
    """Inside the Newton loop the closed-form tets write FLOAT upper-triangle blocks only (360 instead of 1152 bytes per tet) and the double
    blocks of the elements a projection round selects are recomputed on demand (option lazy_eval for staged calls). Same gradient, the same
    matrix up to the float rounding of the contributions (which is what the reference's own assembly does: BlockedSparseMatrix.h:781-814
    casts every element block to float before adding it), the same projection decisions, and the projected matrices agree."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    path = os.path.join(GOLDEN, name + ".npz")
    prob, man, z = ev.load_fixture(path)
    eps32 = np.finfo(np.float32).eps
    out = []
    for lazy in (0, 1):
        eng = engine_from_problem(prob, man)
        eng.set_option("lazy_eval", lazy)
        E, grad = eng.eval(capi.EVAL_P_G_H)
        assert _rel(grad, z["grad"]) < 1e-11
        eng.assemble()
        rp, cols, v0 = eng.get_bsr()
        S = sp.bsr_matrix((v0.astype(np.float64), cols, rp), shape=(prob.ndofs, prob.ndofs)).tocsr()
        Sref = sp.coo_matrix((z["A_vals"], (z["A_rows"], z["A_cols"])), shape=S.shape).tocsr()
        assert abs(S - Sref).max() <= 64 * eps32 * abs(Sref).max()
        xs, info = eng.pcg(man["pcg"]["abs_tol"])
        assert abs(info.n_iterations - man["pcg"]["iterations"]) <= 1
        # a progressive round (rows with a large gradient), then everything: deltas go into the assembled matrix
        act = (np.abs(z["grad"]).reshape(-1, 3).max(axis=1) >= 0.3 * np.abs(z["grad"]).max()).astype(np.uint8)
        r1 = eng.project(1e-10, False, act)
        v1 = eng.get_bsr()[2].copy()
        r2 = eng.project(1e-10, False, None)
        v2 = eng.get_bsr()[2].copy()
        out.append((grad, v0, r1, v1, r2, v2))
        if lazy:
            for pi, pid in eng.pot_ids.items():
                if "TetStrain" in man["potentials"][pi]["name"]:
                    with pytest.raises(RuntimeError):
                        eng.element_hessians(pid, man["potentials"][pi]["n_elem"])
        eng.close()
    a, b = out
    assert _rel(b[0], a[0]) < 1e-12
    scale = np.abs(a[1]).max()
    assert np.abs(b[1] - a[1]).max() <= 16 * eps32 * scale
    assert b[2] == a[2] and b[4] == a[4]          # (n_projected, n_changed) of both rounds
    assert a[4][0] > 0
    assert np.abs(b[3] - a[3]).max() <= 64 * eps32 * scale
    assert np.abs(b[5] - a[5]).max() <= 64 * eps32 * scale


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("name", ["tetbeam_softrubber_6x2x2", "tetbeam_eo_8x2x2", "contactmix_t1", "rbchain", "cloth_shells_6", "attachzoo"])
def test_split_and_mirrored_gather_equal_the_lane_per_block_gather(name, lazy):
    """Whole-part assemblies deal a wavefront's long lists out to groups of eight lanes (k_assemble_gather_split) and, with the lazy float pool,
    write a block and its transpose from ONE sum (k_sym_classify): the same contributions in another order of additions. Against the kernel that
    sums every block in list order (options no_split_gather, no_sym_gather): identical pattern; every entry within ONE float rounding step of it
    (double accumulators: the sums differ in the last bits of a double, which the rounding to float sees only when a sum sits on a rounding
    boundary), almost every entry identical, and the matrix within the usual bound of the reference's."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    out = []
    for opts in ((), ("no_split_gather", "no_sym_gather")):
        eng = engine_from_problem(prob, man)
        eng.set_option("lazy_eval", lazy)
        for o in opts:
            eng.set_option(o, 1)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        out.append(eng.get_bsr())
        eng.close()
    (rp, cols, a), (rp2, cols2, b) = out
    assert (rp == rp2).all() and (cols == cols2).all()
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    step = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    assert (np.abs(a64 - b64) <= step).all()
    assert (a != b).mean() < 1e-3
    Sref = sp.coo_matrix((z["A_vals"], (z["A_rows"], z["A_cols"])), shape=(prob.ndofs, prob.ndofs)).tocsr()
    S = sp.bsr_matrix((a64, cols, rp), shape=(prob.ndofs, prob.ndofs)).tocsr()
    assert abs(S - Sref).max() <= 64 * np.finfo(np.float32).eps * abs(Sref).max()


@pytest.mark.parametrize("name", ["tetbeam_softrubber_6x2x2", "contactmix_t1", "rbchain"])
def test_pcg_fused_direction_variant(name):
    """Option fuse_dir: the search direction p = z + beta p formed inside the SpMV instead of by k_pcg_dir (a measured-slower variant kept as a
    cross-check of the solver's control flow): same iteration count, same solution."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    out = []
    for fuse in (0, 1):
        eng = engine_from_problem(prob, man)
        eng.set_option("fuse_dir", fuse)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        x, info = eng.pcg(man["pcg"]["abs_tol"])
        x2, info2 = eng.pcg(1e-10, 1e-9, 7)      # stops at the iteration cap: the last test runs without a following SpMV
        out.append((x, info.n_iterations, info.converged, x2, info2.n_iterations, info2.converged))
        eng.close()
    a, b = out
    assert a[1] == b[1] and a[2] == b[2] and a[4] == b[4] and a[5] == b[5]
    assert _rel(b[0], a[0]) < 1e-12 and _rel(b[3], a[3]) < 1e-12


def test_pcg_back_to_back_solves_do_not_see_each_others_lookahead():
    """Many solves in a row (the look-ahead batch of one solve publishes its "done" into a pinned slot the next solve watches, too: the
    published block carries the solve's epoch): every solve reports what it reported the first time, to the bit."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))
    eng = engine_from_problem(prob, man)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    tols = [1e-2, 1e-10, 1e-4, 1e-12, 1e-6, 1e-3, 1e-9]
    alone = []
    for t in tols:
        du, info = eng.pcg(t, 1e-14, 5000)
        alone.append((info.n_iterations, info.converged, du.copy()))
    assert len({a[0] for a in alone}) >= 4                      # the tolerances really give different iteration counts
    for rep in range(20):
        for t, (n_it, conv, du_ref) in zip(tols, alone):
            du, info = eng.pcg(t, 1e-14, 5000)
            assert (info.n_iterations, info.converged) == (n_it, conv)
            assert (du == du_ref).all()
    eng.close()


def test_sampled_spmv_launches_report_a_device_clock_duration_and_leave_the_solve_alone():
    """bench.py's roofline timing: every 32nd SpMV of a solve is bracketed by HIP events AND stamps per-workgroup start / end times of the
    device's constant clock into pinned memory. The solve's bits do not depend on the sampling, the clock figure is positive and below
    the event bracket around the same launches (which also contains the dispatch and the marker packets)."""
    import ctypes as C

    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))
    eng = engine_from_problem(prob, man)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    du0, info0 = eng.pcg(1e-300, 1e-300, 100)                    # (runs to the iteration cap)
    assert info0.n_iterations >= 64                              # several sampled launches
    eng.spmv_timing(reset=1)
    for rep in range(4):
        du, info = eng.pcg(1e-300, 1e-300, 100)
        assert info.n_iterations == info0.n_iterations and (du == du0).all()
    ms, n = C.c_double(), C.c_int64()
    assert eng.L.mistark_spmv_device_clock(eng.h, C.byref(ms), C.byref(n)) == 0
    ev_ms, ev_n, nbytes = eng.spmv_timing(reset=-1)
    assert n.value == ev_n == 4 * (info0.n_iterations // 32)
    assert 0.0 < ms.value < ev_ms < 1.0
    assert eng.L.mistark_spmv_device_clock(eng.h, C.byref(ms), C.byref(n)) == 0 and n.value == 0   # reset with the event timing
    eng.close()


@pytest.mark.parametrize("name", ["tetbeam_eo_4x1x1_big", "cloth_shells_6"])
def test_reduced_matrix_projection_with_mirroring(name):
    """project_to_pd_use_mirroring (negative eigenvalues become their mirror image instead of eps): the reduced-matrix kernel for
    translation-invariant elements against the full-size kernel on the same elements, and both against a numpy eigen-decomposition."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    out = []
    for variant in (2, 10):
        eng = engine_from_problem(prob, man)
        eng.set_option("proj_variant", variant)
        eng.eval(capi.EVAL_P_G_H)
        H0 = {pi: eng.element_hessians(pid, man["potentials"][pi]["n_elem"])[0] for pi, pid in eng.pot_ids.items()}
        eng.project(1e-10, True, None)
        H = {pi: eng.element_hessians(pid, man["potentials"][pi]["n_elem"])[0] for pi, pid in eng.pot_ids.items()}
        out.append((H0, H))
        eng.close()
    (H0, Ha), (_, Hb) = out
    checked = 0
    for pi in Ha:
        den = np.maximum(np.sqrt((H0[pi] ** 2).sum(axis=(1, 2))), 1e-300)
        assert (np.sqrt(((Ha[pi] - Hb[pi]) ** 2).sum(axis=(1, 2))) <= 1e-10 * den).all()
        w, V = np.linalg.eigh(0.5 * (H0[pi] + H0[pi].transpose(0, 2, 1)))
        wm = np.where(w < 1e-10, -w, w)
        ref = np.einsum("eik,ek,ejk->eij", V, wm, V)
        assert (np.sqrt(((Ha[pi] - ref) ** 2).sum(axis=(1, 2))) <= 1e-9 * den).all()
        checked += len(den)
    assert checked > 0


@pytest.mark.parametrize("n_bad", [1, -1])
def test_newton_reports_nan_state_instead_of_accepting_it(n_bad):
    """A NaN in one DoF (or in all of them) makes the gradient NaN: its largest magnitude must come back as non-finite (a max that drops
    NaNs would report 0 = "converged") and the Newton solve must end with a failure code, not Successful and not an endless loop."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))
    eng = engine_from_problem(prob, man)
    u = eng.get_dofs()
    if n_bad == 1:
        u[7] = np.nan
    else:
        u[:] = np.nan
    eng.set_dofs(u)
    E, g = eng.eval(capi.EVAL_P_G)
    assert not np.isfinite(g).all()
    res, st = eng.newton_solve()
    assert res == "LinearSystemSolveFailure" and st.newton_iterations <= 1
    eng.close()


def test_connectivity_entry_outside_its_array_is_an_error_not_a_memory_fault():
    """An element that points past the end of an array it is bound to: refused when the connectivity is uploaded (first evaluation), with the
    potential, the element and the column in the message."""
    import copy

    from gpu_util import engine_from_problem
    from stark_amd import capi
    from stark_amd.engine import EngineError

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_softrubber_6x2x2.npz"))
    bad = copy.deepcopy(prob)
    k = [i for i, p in enumerate(bad.potentials) if p.name == "EnergyTetStrain"][0]
    col = [b.conn for b in bad.potentials[k].bindings if b.conn >= 0][-1]
    bad.potentials[k].conn[3, col] = 10 ** 6
    eng = engine_from_problem(bad, man)
    with pytest.raises(EngineError) as ei:
        eng.eval(capi.EVAL_P_G_H)
    assert "EnergyTetStrain" in str(ei.value) and "element 3" in str(ei.value)
    eng.close()


def test_a_registration_only_context_does_not_disable_device_buffers_of_real_ones():
    """mistark_create_dry used to switch device allocation off for the whole process: a real context created afterwards ran
    its kernels on null buffers. The flag belongs to the context now; a dry context alive beside a real one changes nothing."""
    import ctypes as C

    from gpu_util import engine_from_problem
    from stark_amd import capi

    L = capi.lib()
    dry = C.c_void_p()
    assert L.mistark_create_dry(C.byref(dry)) == 0
    v = np.zeros(12)
    L.mistark_add_dof_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    assert L.mistark_add_dof_set(dry, b"soft.v1", v.ctypes.data, 12) >= 0
    try:
        prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "tetbeam_eo_4x1x1.npz"))
        eng = engine_from_problem(prob, man)
        E, grad = eng.eval(capi.EVAL_P_G_H)
        Eo, go, _ = ev.evaluate_all(prob)
        assert abs(E - Eo) <= 1e-11 * abs(Eo)
        assert np.abs(grad - go).max() <= 1e-11 * np.abs(go).max()
        # ... and the dry one still refuses to evaluate
        L.mistark_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        e = C.c_double()
        assert L.mistark_eval(dry, capi.EVAL_P, C.byref(e), None) != 0
        E2, _ = eng.eval(capi.EVAL_P_G_H)
        assert E2 == E
        eng.close()
    finally:
        L.mistark_destroy(dry)


@pytest.mark.parametrize("name", ["tetbeam_softrubber_6x2x2", "contactmix_t1", "rbchain", "cloth_shells_6"])
def test_chronopoulos_gear_iteration_on_one_gpu_matches_the_reference_loop(name):
    """option cg_variant = 1: the arithmetic the sharded solve uses between processes (u = M^-1 r, w = A u, s = A p by recurrence, p.Ap by
    expansion; kernels.hip pcg_cg / pcg_sharded_fused) on ONE GPU against the reference's loop (solve_pcg.h:170-225): same verdict, iteration
    count within +-1, same solution — at the fixture's tolerance, at a tight one on the projected matrix, and at the iteration cap."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))

    def solves(eng):
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        a = eng.pcg(man["pcg"]["abs_tol"])
        eng.project(1e-10)
        b = eng.pcg(1e-8, 1e-6, 5000)
        c = eng.pcg(1e-300, 1e-300, 9)
        return a, b, c

    eng = engine_from_problem(prob, man)
    ref = solves(eng)
    eng.set_option("cg_variant", 1)
    got = solves(eng)
    eng.close()
    for (x, i), (xr, ir), slack in zip(got, ref, (1, 2, 0)):
        assert i.converged == ir.converged and abs(i.n_iterations - ir.n_iterations) <= slack, (i.n_iterations, ir.n_iterations)
        assert np.abs(x - xr).max() <= 1e-4 * max(np.abs(xr).max(), 1e-300)


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("name", ["tetbeam_eo_4x1x1_big", "contactmix_t1", "cloth_shells_6", "rbchain"])
def test_projection_updates_the_matrix_in_order(name, lazy):
    """The matrix after a projection round is the matrix assembled from the projected Hessians, BIT FOR BIT: project() flags the blocks its
    elements contribute to and gathers them again from the pools in sorted-key order (ElementHessians.cpp:258-294 adds float deltas in thread
    order; the engine's first version added them with float atomics in arrival order). Progressive round, then everything; on the double pool
    and on the lazy float pool (projected blocks written back to it); twice the same bits. Option atomic_projection keeps the delta path:
    within float rounding of it."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    act = (np.abs(z["grad"]).reshape(-1, 3).max(axis=1) >= 0.3 * np.abs(z["grad"]).max()).astype(np.uint8)

    def run(atomic):
        eng = engine_from_problem(prob, man)
        eng.set_option("lazy_eval", lazy)
        eng.set_option("atomic_projection", atomic)
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        eng.project(1e-10, False, act)
        v1 = eng.get_bsr()[2].copy()
        eng.project(1e-10, False, None)
        v2 = eng.get_bsr()[2].copy()
        eng.assemble()                       # from the pools, which hold the projected Hessians
        v3 = eng.get_bsr()[2].copy()
        eng.close()
        return v1, v2, v3

    a, b, d = run(0), run(0), run(1)
    assert (a[1] == a[2]).all()                                    # patched == re-assembled
    assert all((x == y).all() for x, y in zip(a, b))               # run to run
    scale = np.abs(a[2]).max()
    assert np.abs(d[1] - a[1]).max() <= 64 * np.finfo(np.float32).eps * scale and np.abs(d[0] - a[0]).max() <= 64 * np.finfo(np.float32).eps * scale


@pytest.mark.parametrize("name", ["tetbeam_softrubber_6x2x2", "tetbeam_eo_8x2x2", "cloth_shells_6", "attachzoo"])
def test_lumped_inertia_one_lane_per_node_has_the_generic_kernels_bits(name):
    """k_eval_lumped_inertia (one lane per node: inputs gathered once, the hyper-dual expression run for the three diagonal pairs) against the
    generic kernel (six lanes per node; option generic_inertia): the SAME element energies, element Hessians and — on a problem holding only
    that potential's contribution per row beside others summed identically — the same energy and gradient, bit for bit. A hand-derived gradient
    differs in its last bit, and the attachment / contact trajectories amplify that into other Newton counts (round 6)."""
    from gpu_util import engine_from_problem
    from stark_amd import capi

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    pi = [p.name for p in prob.potentials].index("EnergyLumpedInertia")
    n_elem = prob.potentials[pi].conn.shape[0]
    out = []
    for generic in (0, 1):
        eng = engine_from_problem(prob, man)
        eng.set_option("generic_inertia", generic)
        E, g = eng.eval(capi.EVAL_P_G_H)
        H, rows = eng.element_hessians(eng.pot_ids[pi] if hasattr(eng, "pot_ids") else pi, n_elem)
        E2, g2 = eng.eval(capi.EVAL_P_G)
        out.append((E, g, H, rows, E2, g2))
        eng.close()
    a, b = out
    assert (a[3] == b[3]).all() and (a[2] == b[2]).all() and np.abs(a[2]).max() > 0
    # energy and gradient: the inertia's node gradients are single additions per row; the other potentials' sums are the same kernels in both runs
    assert a[0] == b[0] and a[4] == b[4]
    assert np.abs(a[1] - b[1]).max() <= 4 * np.finfo(np.float64).eps * np.abs(b[1]).max()
    assert np.abs(a[5] - b[5]).max() <= 4 * np.finfo(np.float64).eps * np.abs(b[5]).max()
