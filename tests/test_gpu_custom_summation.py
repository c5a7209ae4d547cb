"""GPU (-m gpu): the summation loop of user-defined potentials (MappedWorkspace::add_for_each, symx/src/compile/MappedWorkspace.h:123-130;
how the reference runs it: CompiledInLoop_run.h:375-400 — the kernel is called once per row of the summation data with the summation symbols
replaced, and the outputs are accumulated). Hand-written op sequence in SymX's encoding (rows {type, dst, a, b, cond}, Expr.h:12-43):
    E(v; k, w) = k * w0 * (v . v) + w1 * v_x^3          summed over the rows of `w`
against the closed form; and the same potential registered once per row WITHOUT the loop (sum of potentials) as a cross-check of the
accumulation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MUL, ADD, SYMBOL, POWN = 8, 6, 5, 10


def _program():
    # inputs: v (0, 1, 2), k (3), w0 w1 (4, 5); temporaries from 6
    ops = [
        (MUL, 6, 0, 0, -1), (MUL, 7, 1, 1, -1), (MUL, 8, 2, 2, -1), (ADD, 9, 6, 7, -1), (ADD, 10, 9, 8, -1),   # v . v
        (MUL, 11, 3, 4, -1), (MUL, 12, 11, 10, -1),                                                                # k w0 (v . v)
        (POWN, 13, 0, 3, -1), (MUL, 14, 5, 13, -1),                                                                # w1 v_x^3
        (ADD, 15, 12, 14, -1), (SYMBOL, 0, 15, -1, -1),
    ]
    return np.array(ops, dtype=np.int32), np.zeros(len(ops))


def _engine(n, W, looped):
    from stark_amd.engine import Engine

    rng = np.random.default_rng(7)
    v = rng.standard_normal((n, 3))
    k = np.array([1.7])
    eng = Engine(0)
    eng.add_dof_set("v", v)
    a_v = eng.L.mistark_dof_array(eng.h, 0, 3)
    a_k = eng.array(k, 1)
    conn = np.arange(n, dtype=np.int32).reshape(-1, 1)
    ops, cst = _program()
    pids = []
    if looped:
        w_slot = np.array(W[0], dtype=np.float64)   # a binding covers the summation symbols like any other input
        a_w = eng.array(w_slot, 2)
        pid = eng.potential_custom("UserSum", conn, [(a_v, 3, 0), (a_k, 1, -1), (a_w, 2, -1)], ops, cst, 6)
        eng.potential_custom_set_summation(pid, 4, W)
        pids.append(pid)
    else:
        for i, w in enumerate(W):
            a_w = eng.array(np.array(w, dtype=np.float64), 2)
            pids.append(eng.potential_custom("UserSum%d" % i, conn, [(a_v, 3, 0), (a_k, 1, -1), (a_w, 2, -1)], ops, cst, 6))
    return eng, v, k[0], pids


def test_summation_loop_equals_the_closed_form_and_the_sum_of_potentials():
    from stark_amd import capi

    n = 37
    W = np.array([[0.5, 0.1], [0.25, -0.3], [1.5, 0.02]])
    eng, v, k, (pid,) = _engine(n, W, looped=True)
    E, g = eng.eval(capi.EVAL_P_G_H)
    H, rows = eng.element_hessians(pid, n)
    a, b = k * W[:, 0].sum(), W[:, 1].sum()
    E_ref = (a * (v * v).sum(1) + b * v[:, 0] ** 3).sum()
    g_ref = 2 * a * v
    g_ref[:, 0] += 3 * b * v[:, 0] ** 2
    assert abs(E - E_ref) <= 1e-13 * abs(E_ref)
    assert np.abs(g.reshape(-1, 3) - g_ref).max() <= 1e-13 * np.abs(g_ref).max()
    H_ref = np.tile(2 * a * np.eye(3), (n, 1, 1))
    H_ref[:, 0, 0] += 6 * b * v[:, 0]
    assert np.abs(H - H_ref).max() <= 1e-13 * np.abs(H_ref).max()
    assert (rows[:, 0] == np.arange(n)).all()
    E_p, _ = eng.eval(capi.EVAL_P)
    assert abs(E_p - E_ref) <= 1e-13 * abs(E_ref)
    eng.close()
    eng2, _, _, pids = _engine(n, W, looped=False)
    E2, g2 = eng2.eval(capi.EVAL_P_G_H)
    assert abs(E2 - E) <= 1e-14 * abs(E) and np.abs(g2 - g).max() <= 1e-13 * np.abs(g).max()
    eng2.close()


def test_summation_outside_the_inputs_is_refused():
    from stark_amd.engine import EngineError

    eng, _, _, (pid,) = _engine(5, np.array([[0.5, 0.1]]), looped=True)
    with pytest.raises(EngineError, match="summation inputs outside"):
        eng.potential_custom_set_summation(pid, 5, np.array([[1.0, 2.0]]))
    eng.close()


def test_condition_with_a_summation_loop_is_summed_like_the_energy():
    """A conditional potential (add_potential with a DefWCondFunc) whose workspace has a summation loop: the reference compiles the condition
    over the same workspace and runs it through the loop too (SecondOrderCompiledPotential.cpp:185-197 with CompiledInLoop_run.h:375-400), so an
    element is active iff the SUM of the condition over the rows is > 0 — here c(v; w) = w0 * v_x summed over the rows = (sum w0) v_x, with
    sum w0 > 0 although the first row's w0 is negative (a condition evaluated on the first row alone would switch the opposite elements on)."""
    from stark_amd import capi
    from stark_amd.engine import Engine

    n = 41
    W = np.array([[-0.5, 0.1], [0.25, -0.3], [1.5, 0.02]])
    rng = np.random.default_rng(11)
    v = rng.standard_normal((n, 3))
    k = np.array([1.7])
    eng = Engine(0)
    eng.add_dof_set("v", v)
    a_v = eng.L.mistark_dof_array(eng.h, 0, 3)
    a_k = eng.array(k, 1)
    a_w = eng.array(np.array(W[0], dtype=np.float64), 2)
    conn = np.arange(n, dtype=np.int32).reshape(-1, 1)
    ops, cst = _program()
    cops = np.array([(MUL, 6, 4, 0, -1), (SYMBOL, 0, 6, -1, -1)], dtype=np.int32)   # w0 * v_x
    pid = eng.potential_custom("UserCondSum", conn, [(a_v, 3, 0), (a_k, 1, -1), (a_w, 2, -1)], ops, cst, 6, cops, np.zeros(len(cops)))
    eng.potential_custom_set_summation(pid, 4, W)
    E, g = eng.eval(capi.EVAL_P_G_H)
    on = (W[:, 0].sum() * v[:, 0]) > 0
    assert 5 < on.sum() < n - 5 and not (on == ((W[0, 0] * v[:, 0]) > 0)).all()
    a, b = k[0] * W[:, 0].sum(), W[:, 1].sum()
    e_el = a * (v * v).sum(1) + b * v[:, 0] ** 3
    E_ref = e_el[on].sum()
    g_ref = 2 * a * v
    g_ref[:, 0] += 3 * b * v[:, 0] ** 2
    g_ref[~on] = 0.0
    assert abs(E - E_ref) <= 1e-13 * abs(E_ref)
    assert np.abs(g.reshape(-1, 3) - g_ref).max() <= 1e-13 * np.abs(g_ref).max()
    eng.close()


def test_summation_over_a_dof_binding_is_refused():
    """inputs 0..2 are the DoF binding `v`: a summation loop would overwrite them with constants while their derivative seeds stay"""
    from stark_amd import capi
    from stark_amd.engine import EngineError

    eng, _, _, (pid,) = _engine(5, np.array([[0.5, 0.1]]), looped=True)
    eng.potential_custom_set_summation(pid, 1, np.array([[1.0, 2.0]]))   # inputs 1, 2 = v_y, v_z
    with pytest.raises(EngineError, match="overlap a binding on a DoF set"):
        eng.eval(capi.EVAL_P_G_H)
    eng.close()
