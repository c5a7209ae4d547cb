"""GPU (-m gpu): user-defined potentials as EMITTED kernels (stark_amd/csrc/custom.hip "the emitter": symx::Sequence -> HIP
source -> hipRTC, the mirror of the reference's scalar emitter symx/src/compile/Compilation.cpp:381-469) against the device interpreter on
the same sequence: same energy, gradient and element Hessians to rounding (the two call the same hyper-dual operations, custom_math.hpp; the
compiler may contract differently), and VERDICT r04 #9's bar: a 100 k-element user potential at least 5 times faster than interpreted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MUL, ADD, SUB, SYMBOL, POWN, RECIP, SQRT, CONST, BRANCH, EXP = 8, 6, 7, 5, 10, 9, 12, 4, 2, 15


def _magnet_program():
    """E = -k / |x0 + dt v - m| + 0.5 c (v . v) exp(-|v|^2) with a branch: if (k > 0) ... else 0  — the README's EnergyMagneticAttraction
    plus a damping term, written as SymX would emit it. inputs: v (0..2, DoF), x0 (3..5), dt (6), k (7), m (8..10), c (11); temporaries from 12."""
    ops, cst = [], []

    def op(*row, c=0.0):
        ops.append(row)
        cst.append(c)

    t = 12
    op(BRANCH, -1, 0, -1, 7)                                   # if (k > 0)
    for d in range(3):                                         # r_d = x0_d + dt v_d - m_d
        op(MUL, t + 3 * d, 6, d, -1)
        op(ADD, t + 3 * d + 1, 3 + d, t + 3 * d, -1)
        op(SUB, t + 3 * d + 2, t + 3 * d + 1, 8 + d, -1)
    r = [t + 2, t + 5, t + 8]
    u = t + 9
    op(MUL, u, r[0], r[0], -1); op(MUL, u + 1, r[1], r[1], -1); op(MUL, u + 2, r[2], r[2], -1)
    op(ADD, u + 3, u, u + 1, -1); op(ADD, u + 4, u + 3, u + 2, -1)
    op(SQRT, u + 5, u + 4, -1, -1); op(RECIP, u + 6, u + 5, -1, -1); op(MUL, u + 7, 7, u + 6, -1)
    op(CONST, u + 8, -1, -1, -1, c=-1.0); op(MUL, u + 9, u + 8, u + 7, -1)           # -k / d
    w = u + 10
    op(MUL, w, 0, 0, -1); op(MUL, w + 1, 1, 1, -1); op(MUL, w + 2, 2, 2, -1); op(ADD, w + 3, w, w + 1, -1); op(ADD, w + 4, w + 3, w + 2, -1)
    op(MUL, w + 5, u + 8, w + 4, -1); op(EXP, w + 6, w + 5, -1, -1)                      # exp(-|v|^2)
    op(CONST, w + 7, -1, -1, -1, c=0.5); op(MUL, w + 8, w + 7, 11, -1); op(MUL, w + 9, w + 8, w + 4, -1); op(MUL, w + 10, w + 9, w + 6, -1)
    op(ADD, w + 11, u + 9, w + 10, -1)
    op(BRANCH, -1, 1, -1, -1)                                  # else
    op(0, w + 11, -1, -1, -1)                                  # Zero
    op(BRANCH, -1, -1, -1, -2)                                 # endif
    op(SYMBOL, 0, w + 11, -1, -1)
    return np.array(ops, dtype=np.int32), np.array(cst)


def _engine(n, rtc):
    from stark_amd.engine import Engine

    rng = np.random.default_rng(3)
    v = 0.3 * rng.standard_normal((n, 3))
    x0 = rng.uniform(-1.0, 1.0, (n, 3))
    eng = Engine(0)
    eng.add_dof_set("v", v)
    a_v = eng.L.mistark_dof_array(eng.h, 0, 3)
    a_x0 = eng.array(x0, 3)
    a_dt, a_k, a_c = eng.array(np.array([1.0 / 30.0]), 1), eng.array(np.array([20.0]), 1), eng.array(np.array([0.7]), 1)
    a_m = eng.array(np.array([[0.3, 0.2, 2.6]]), 3)
    conn = np.arange(n, dtype=np.int32).reshape(-1, 1)
    ops, cst = _magnet_program()
    pid = eng.potential_custom("UserMagnet", conn, [(a_v, 3, 0), (a_x0, 3, 0), (a_dt, 1, -1), (a_k, 1, -1), (a_m, 3, -1), (a_c, 1, -1)], ops, cst, 12)
    eng.set_option("custom_rtc", rtc)
    return eng, pid, (v, x0)


def _closed_form(v, x0):
    dt, k, m, c = 1.0 / 30.0, 20.0, np.array([0.3, 0.2, 2.6]), 0.7
    r = x0 + dt * v - m
    d = np.linalg.norm(r, axis=1)
    vv = (v * v).sum(1)
    E = -k / d + 0.5 * c * vv * np.exp(-vv)
    g = (k / d ** 3)[:, None] * r * dt + (c * np.exp(-vv) * (1.0 - vv))[:, None] * v
    return E.sum(), g


def test_emitted_kernels_equal_the_interpreter_and_the_closed_form():
    from stark_amd import capi

    n = 5000
    res = {}
    for rtc in (0, 1):
        eng, pid, (v, x0) = _engine(n, rtc)
        E, g = eng.eval(capi.EVAL_P_G_H)
        H, rows = eng.element_hessians(pid, n)
        Ep, _ = eng.eval(capi.EVAL_P)
        Eg, gg = eng.eval(capi.EVAL_P_G)
        res[rtc] = (E, g.copy(), H.copy(), Ep, Eg, gg.copy())
        assert (eng.counter("rtc_launches") > 0) == bool(rtc) and eng.counter("rtc_builds") == rtc
        eng.close()
    E_ref, g_ref = _closed_form(v, x0)
    for rtc in (0, 1):
        E, g, H, Ep, Eg, gg = res[rtc]
        assert abs(E - E_ref) <= 1e-13 * abs(E_ref) and abs(Ep - E_ref) <= 1e-13 * abs(E_ref) and abs(Eg - E_ref) <= 1e-13 * abs(E_ref)
        assert np.abs(g.reshape(-1, 3) - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
        assert np.abs(gg - g).max() <= 1e-13 * np.abs(g).max()
    # emitted against interpreted: the same operations in the same order
    assert abs(res[1][0] - res[0][0]) <= 1e-14 * abs(res[0][0])
    assert np.abs(res[1][1] - res[0][1]).max() <= 1e-13 * np.abs(res[0][1]).max()
    assert np.abs(res[1][2] - res[0][2]).max() <= 1e-13 * np.abs(res[0][2]).max()
    assert np.abs(res[1][2] - np.transpose(res[1][2], (0, 2, 1))).max() == 0.0     # symmetric to the bit, like the interpreter's


def test_emitted_kernels_are_at_least_five_times_faster_on_100k_elements():
    from stark_amd import capi

    n = 100_000
    t = {}
    for rtc in (0, 1):
        eng, pid, _ = _engine(n, rtc)
        eng.eval(capi.EVAL_P_G_H)           # (build / first touch)
        eng.set_option("custom_timing", 1)  # HIP events around the potential's own launch (counter custom_kernel_us)
        for _ in range(10):
            eng.eval(capi.EVAL_P_G_H)
        t[rtc] = eng.counter("custom_kernel_us") / 10
        eng.close()
    print("user potential, 100 k elements, energy + gradient + Hessian: interpreted %.1f us, emitted %.1f us per launch (%.1fx)" % (t[0], t[1], t[0] / max(t[1], 1e-9)))
    assert t[1] > 0 and t[0] >= 5.0 * t[1], t
