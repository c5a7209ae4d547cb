"""Pins the CPU oracle (oracle/*.py) against golden vectors produced by the UNMODIFIED reference
(tests/golden/*.npz, generator tests/golden/make_fixtures.py -> oracle/ref_harness.cpp).

Tolerances: element P/g/H 1e-11 relative (double, different operation order); projected Hessians 1e-9
relative; assembled blocks: float eps * contributions; PCG: same iteration count, solution within 10*rel_tol.
"""
import glob
import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import evaluator as ev

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fixture_list import stage_dumps  # noqa: E402

DUMPS = stage_dumps()


# EnergyDiscreteShells evaluates acos((1-1e-12) n0.n1) on (nearly) flat hinges: d acos/dx = -1/sqrt(1-x^2) with
# 1-x^2 ~ 2e-12 amplifies round-off of the argument by ~1e6, so any two double evaluations with different operation
# order (the reference's generated code vs any restatement) agree only to ~1e-9 relative there.
ELEMENT_TOL = {"EnergyDiscreteShells": 1e-8}


def _rel(a, b):
    s = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / s


@pytest.mark.parametrize("path", DUMPS, ids=[os.path.basename(p)[:-4] for p in DUMPS])
def test_oracle_matches_reference_stages(path):
    prob, man, z = ev.load_fixture(path)
    E, grad, outs = ev.evaluate_all(prob)
    assert abs(E - man["E"]) <= 1e-12 * max(1.0, sum(abs(p.get("E", 0.0)) for p in man["potentials"]))
    assert _rel(grad, z["grad"]) < max([1e-11] + [ELEMENT_TOL.get(p["name"], 0) for p in man["potentials"] if p["n_elem"] > 0])
    names = [p["name"] for p in man["potentials"]]
    n_h = 0
    for o in outs:
        pi = names.index(o.name)
        ref = man["potentials"][pi]
        assert len(o.E) == ref["n_hessians"]
        n_h += len(o.E)
        assert abs(o.E.sum() - ref["E"]) <= 1e-11 * max(1.0, np.abs(o.E).sum())
        assert abs(ref["E"] - ref["E_only"]) <= 1e-11 * max(1.0, np.abs(o.E).sum())
        assert (z["p%d_hrows" % pi] == o.block_rows).all()
        tol = ELEMENT_TOL.get(o.name, 1e-11)
        assert _rel(o.H, z["p%d_hvals" % pi]) < tol, o.name
        Hp, changed = ev.project_to_pd(o.H)
        assert _rel(Hp, z["p%d_hvals_proj" % pi]) < max(1e-9, 10 * tol), o.name
    assert n_h == man["n_hessians"]

    # assembly (float storage)
    A = ev.assemble(outs, prob.ndofs)
    S = A.to_scipy().tocsr()
    Sref = sp.coo_matrix((z["A_vals"], (z["A_rows"], z["A_cols"])), shape=S.shape).tocsr()
    assert S.nnz == Sref.nnz == man["nnz_scalar"]
    assert abs(S - Sref).max() <= 64 * np.finfo(np.float32).eps * abs(Sref).max()

    # SpMV / preconditioner probes
    x = np.sin(0.37 * np.arange(prob.ndofs))
    assert _rel(A.spmv(x), z["spmv_y"]) < 1e-5
    dinv = ev.block_diag_inverse(A)
    assert _rel(ev.apply_preconditioner(dinv, x), z["prec_z"]) < 1e-4

    # PCG with the Newton forcing tolerance
    assert abs(ev.forcing_abs_tol(man["residual"]) - man["pcg"]["abs_tol"]) < 1e-15
    xs, info = ev.solve_pcg(A, -grad, man["pcg"]["abs_tol"])
    assert info.converged == bool(man["pcg"]["converged"])
    assert abs(info.n_iterations - man["pcg"]["iterations"]) <= 1
    if info.n_iterations == man["pcg"]["iterations"]:
        assert _rel(xs, z["pcg_x"]) < 1e-3


# (rigid-body trajectories need the rigid-body state update and constraint hardening of the host layer: tests/test_gpu_scene.py)
TRAJ = [p for p in DUMPS if os.path.basename(p).startswith("traj_") and "rb" not in os.path.basename(p) and "box" not in os.path.basename(p) and "attach" not in os.path.basename(p) and "llt" not in os.path.basename(p)]  # (rigid-body and contact trajectories: scene tests)


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[:-4] for p in TRAJ])
def test_oracle_newton_trajectory(path):
    """Contact-free scenes: Newton / CG iteration counts equal the reference's, evaluation points within 1e-6 relative."""
    prob, man, z = ev.load_fixture(path)
    traj = json.loads(bytes(z["traj_json"]).decode())
    pts = []
    newton_its, cg_total = [], 0
    for s in range(len(traj["steps"])):
        st = ev.run_time_step(prob, on_eval_point=lambda u: pts.append(u.copy()))
        assert st.result == "Successful"
        newton_its.append(st.newton_iterations)
        cg_total += st.cg_iterations
    assert newton_its == traj["newton_iterations"]
    assert abs(cg_total - sum(traj["cg_iterations"])) <= max(2, 0.05 * sum(traj["cg_iterations"]))
    ref = z["iterates"]
    assert len(pts) == ref.shape[0]
    for a, b in zip(pts, ref):
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max())
    iv1, ix0, iv0 = ev.point_state_arrays(prob)
    assert np.abs(prob.arrays[ix0] - z["x_end"]).max() <= 1e-6 * np.abs(z["x_end"]).max()


@pytest.mark.parametrize("path", DUMPS, ids=[os.path.basename(p)[:-4] for p in DUMPS])
def test_symx_op_sequences_reproduce_reference_elements(path):
    """The op sequences the fixtures carry (the reference's own symx::Sequence of every potential's energy) evaluated by the oracle's
    interpreter give the reference's element Hessians: pins both the fixtures' sequences and the interpreter used as oracle for
    stark_amd/csrc/custom.hip."""
    from oracle import symx_ops

    prob, man, z = ev.load_fixture(path)
    n_checked = 0
    for pi, ref in enumerate(man["potentials"]):
        if ref["n_elem"] == 0 or ("p%d_ops" % pi) not in z:
            continue
        pot = prob.potentials[pi]
        cops = z["p%d_cops" % pi] if ("p%d_cops" % pi) in z else None
        o = symx_ops.evaluate(prob, pot, z["p%d_ops" % pi], z["p%d_opsc" % pi], cops, z["p%d_copsc" % pi] if cops is not None else None)
        assert len(o.E) == ref["n_hessians"], ref["name"]
        assert abs(o.E.sum() - ref["E"]) <= 1e-11 * max(1.0, np.abs(o.E).sum()), ref["name"]
        assert (z["p%d_hrows" % pi] == o.block_rows).all()
        assert _rel(o.H, z["p%d_hvals" % pi]) < ELEMENT_TOL.get(ref["name"], 1e-11), ref["name"]
        n_checked += 1
    assert n_checked > 0


def test_the_references_own_log_on_exact_ties_depends_on_its_compile_flags():
    """VERDICT r03 #5 / INTEGRATION.md section 4, the measured counter-example. configs[3] with the block CENTRED on the box has 78 of its 264
    edge-edge pairs exactly on a classification tie (closest point at an edge endpoint: edge-edge or edge-point is decided by the last bit of a
    product). The reference's answer there is not a property of its algorithm but of how its compiler contracted multiply-adds: the SAME
    unmodified sources built with -ffp-contract=off (make -C oracle nofma) log other Newton / solve / CG counts than the default build
    (GCC contracts under -mfma) from the very first time-step attempt on — while on the placement 1.4 mm off the axes (no ties) the two builds
    log IDENTICAL counts in all eight attempts, the counts tests/test_gpu_fullsize.py pins the engine to with `==`. An independent
    implementation can be bit-exact in contact-pair indexing to at most one of the two reference builds on the centred placement; what is
    pinned there instead: the engine given the reference's own tie decisions reproduces its log (test_gpu_fullsize.py, shim_check)."""
    import json

    import numpy as np

    g = GOLDEN
    nofma = np.load(os.path.join(g, "steplog_cfg3_nofma.npz"))
    log = lambda z, k: json.loads(bytes(z[k]).decode())
    centred_default = [log(np.load(os.path.join(g, "steplog_cfg3_blockbox_44x44x43.npz")), "time_t%d_json" % t)["per_step"] for t in (8, 4)]
    offset_default = [log(np.load(os.path.join(g, "steplog_cfg3_offset_44x44x43.npz")), "time_t%d_json" % t)["per_step"] for t in (8, 4)]
    centred_nofma = log(nofma, "centred_time_t8_json")
    offset_nofma = log(nofma, "offset_time_t8_json")
    assert centred_nofma["ndofs"] == offset_nofma["ndofs"] == 517050
    # off the ties: one answer, whatever the flags and the thread count
    assert offset_default[0] == offset_default[1] == offset_nofma["per_step"] and len(offset_nofma["per_step"]) == 8
    # on the ties: the default build reproduces itself across thread counts, the contraction-free build disagrees with it from the first attempt on
    assert centred_default[0] == centred_default[1]
    assert centred_nofma["per_step"][0][:2] == centred_default[0][0][:2] == [5, 6]            # same Newton iterations and solves in attempt 1 ...
    assert centred_nofma["per_step"][0][2] != centred_default[0][0][2]                        # ... with other CG counts (101 against 104)
    assert [p[:2] for p in centred_nofma["per_step"]] != [p[:2] for p in centred_default[0]]  # and other Newton / solve counts from attempt 2 on
