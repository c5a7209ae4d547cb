"""oracle/symx_ops.py — TEST INFRASTRUCTURE ONLY (never imported by the product path stark_amd/).

CPU evaluation of a potential given as SymX's straight-line op sequence (symx::Sequence, symx/src/compile/Sequence.h:24-41; the
meaning of each op is the C++ the reference emits for it, symx/src/compile/Compilation.cpp:381-469), on oracle.ad.D2 numbers, so that
the sequences stored in the golden fixtures can be checked against the reference's own element outputs and serve as the oracle of the
engine's device interpreter (stark_amd/csrc/custom.hip).
"""
from __future__ import annotations

import numpy as np

from .ad import D2, where

# symx::ExprType (symx/src/symbol/Expr.h:12-43)
ZERO, ONE, BRANCH, CONST, SYMBOL, ADD, SUB, MUL, RECIP, POWN, POWF, SQRT, LN, LOG10, EXP, SIN, COS, TAN, ASIN, ACOS, ATAN, PRINT = 0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22


def _lift(x, like):
    return x if isinstance(x, D2) or like is None else like._lift(x)


def run(ops, consts, inputs, n_elem):
    """ops: [n, 5] rows {type, dst, a, b, cond}; inputs: list of n_inputs values (D2 for DoFs, ndarray[n_elem] otherwise).
    Returns output 0. Branches are evaluated on both sides and merged per element (the derivatives of the unselected side are discarded,
    exactly like the reference's `if (v > 0) {...} else {...}`)."""
    n_in = len(inputs)
    val = {}
    like = next((x for x in inputs if isinstance(x, D2)), None)

    def get(i):
        return inputs[i] if i < n_in else val[i]

    def num(x):
        return x.v if isinstance(x, D2) else np.broadcast_to(np.asarray(x, dtype=np.float64), (n_elem,))

    out = None
    mask = [np.ones(n_elem, dtype=bool)]   # execution mask stack
    taken = []

    def put(dst, r):
        m = mask[-1]
        if dst in val and not m.all():
            val[dst] = where(m, _lift(r, like), _lift(val[dst], like)) if like is not None else np.where(m, r, val[dst])
        else:
            val[dst] = r

    for (t, dst, a, b, cond), c in zip(np.asarray(ops).tolist(), np.asarray(consts).tolist()):
        if t == BRANCH:
            if cond == -2:
                mask.pop()
                taken.pop()
            elif a == 0:
                tk = num(get(cond)) > 0.0
                taken.append(tk)
                mask.append(mask[-1] & tk)
            else:
                mask[-1] = mask[-2] & ~taken[-1]
            continue
        with np.errstate(all="ignore"):
            if t == SYMBOL:
                r = get(a)
                m = mask[-1]
                out = r if out is None or m.all() else (where(m, _lift(r, like), _lift(out, like)) if like is not None else np.where(m, r, out))
                continue
            if t == ZERO: r = np.zeros(n_elem)
            elif t == ONE: r = np.ones(n_elem)
            elif t == CONST: r = np.full(n_elem, c)
            elif t == ADD: r = get(a) + get(b)
            elif t == SUB: r = get(a) - get(b)
            elif t == MUL: r = get(a) * get(b)
            elif t == RECIP:
                x = get(a)
                r = x.inv() if isinstance(x, D2) else 1.0 / x
            elif t == POWN:
                x = get(a)
                r = x.powN(b) if isinstance(x, D2) else x ** b
            elif t == SQRT:
                x = get(a)
                r = x.sqrt() if isinstance(x, D2) else np.sqrt(x)
            elif t == LN:
                x = get(a)
                r = x.log() if isinstance(x, D2) else np.where(x <= 0.0, -np.inf, np.log(np.where(x <= 0.0, 1.0, x)))
            elif t == SIN:
                x = get(a)
                r = x.sin() if isinstance(x, D2) else np.sin(x)
            elif t == COS:
                x = get(a)
                r = x.cos() if isinstance(x, D2) else np.cos(x)
            elif t == ACOS:
                x = get(a)
                r = x.acos() if isinstance(x, D2) else np.arccos(x)
            elif t == ATAN:
                x = get(a)
                r = x.atan() if isinstance(x, D2) else np.arctan(x)
            elif t == PRINT: r = np.zeros(n_elem)
            else:
                raise NotImplementedError("op type %d is not used by any fixture" % t)
        put(dst, r)
    return out


def evaluate(prob, pot, ops, consts, cond_ops=None, cond_consts=None):
    """Energy / gradient / Hessian of every element of `pot` from its op sequence; same output type as evaluator.evaluate_potential."""
    from .evaluator import ElementOutput, dof_layout

    n_elem = pot.conn.shape[0]
    order = dof_layout(pot)
    nb = len(order)
    n = 3 * nb
    local = {bi: k for k, bi in enumerate(order)}
    inputs = []
    for bi, b in enumerate(pot.bindings):
        data = prob.arrays[b.array]
        vals = data[pot.conn[:, b.conn]] if b.conn >= 0 else np.broadcast_to(data[0], (n_elem, b.stride))
        for c in range(b.stride):
            inputs.append(D2.var(vals[:, c], 3 * local[bi] + c, n) if bi in local else np.ascontiguousarray(vals[:, c]))
    like = next(x for x in inputs if isinstance(x, D2))
    out = like._lift(run(ops, consts, inputs, n_elem))
    active = np.ones(n_elem, dtype=bool)
    if cond_ops is not None and len(cond_ops):
        plain = [x.v if isinstance(x, D2) else x for x in inputs]
        active = np.asarray(run(cond_ops, cond_consts, plain, n_elem)) > 0
    rows = np.stack([prob.dof_offsets[pot.bindings[bi].dof_set] // 3 + pot.conn[:, pot.bindings[bi].conn] for bi in order], axis=1)
    return ElementOutput(pot.name, out.v[active], out.g[active], out.h[active], rows[active].astype(np.int64), active)
