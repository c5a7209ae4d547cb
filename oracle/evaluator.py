"""oracle/evaluator.py — TEST INFRASTRUCTURE ONLY (never imported by the product path stark_amd/).

CPU restatement (numpy) of the reference's second-order evaluation, PSD projection, blocked assembly, block-Jacobi
preconditioner, PCG and Newton loop for the hot path. Each function cites the reference code it follows.

Pinned against the golden fixtures produced by the unmodified reference (tests/golden/*.npz, generator:
tests/golden/make_fixtures.py) by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .ad import D2
from .energies import REGISTRY


# ----------------------------------------------------------------------------------------------------------------------
# Problem description: the same information the reference's GlobalPotential holds
# (symx/src/solver/GlobalPotential.h:24-31, symx/src/compile/data_maps.h:29-105)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class Binding:
    array: int      # index into Problem.arrays
    stride: int
    conn: int       # connectivity column providing the index, -1 = global value
    dof_set: int    # >= 0 if this array is a DoF set


@dataclass
class PotentialDesc:
    name: str
    conn: np.ndarray            # int32 [n_elem, stride]
    bindings: list
    has_condition: bool = False


@dataclass
class Problem:
    dt: float
    ndofs: int
    dof_offsets: list           # per dof set: first scalar dof
    dof_sizes: list
    arrays: list                # list of float64 [n_items, stride]
    potentials: list
    dof_arrays: dict = field(default_factory=dict)   # dof_set -> array index

    def get_dofs(self):
        u = np.zeros(self.ndofs)
        for s, a in self.dof_arrays.items():
            u[self.dof_offsets[s]:self.dof_offsets[s] + self.dof_sizes[s]] = self.arrays[a].reshape(-1)
        return u

    def set_dofs(self, u):
        for s, a in self.dof_arrays.items():
            self.arrays[a] = u[self.dof_offsets[s]:self.dof_offsets[s] + self.dof_sizes[s]].reshape(self.arrays[a].shape).copy()


def load_fixture(path):
    z = np.load(path)
    man = json.loads(bytes(z["manifest_json"]).decode())
    arrays = [np.array(z["a%d" % i], dtype=np.float64) for i in range(man["n_arrays"])]
    pots = []
    dof_arrays = {}
    for pi, p in enumerate(man["potentials"]):
        conn = z["p%d_conn" % pi] if p["n_elem"] > 0 else np.zeros((0, p["conn_stride"]), dtype=np.int32)
        bs = [Binding(b["array"], b["stride"], b["conn"], b["dof_set"]) for b in p["bindings"]]
        for b in bs:
            if b.dof_set >= 0 and b.array >= 0:
                dof_arrays[b.dof_set] = b.array
        pots.append(PotentialDesc(p["name"], conn, bs, bool(p["has_condition"])))
    prob = Problem(dt=man["dt"], ndofs=man["ndofs"], dof_offsets=[d["offset"] for d in man["dof_sets"]],
                   dof_sizes=[d["size"] for d in man["dof_sets"]], arrays=arrays, potentials=pots, dof_arrays=dof_arrays)
    return prob, man, z


# ----------------------------------------------------------------------------------------------------------------------
# Element evaluation  (symx/src/compile/CompiledInLoop_run.h:235-359 gather + kernel call;
#                      symx/src/solver/second_order/SecondOrderCompiledPotential.cpp:10-33 DoF ordering, :142-184 scatter)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class ElementOutput:
    name: str
    E: np.ndarray          # [n_active]
    g: np.ndarray          # [n_active, 3*nb]
    H: np.ndarray          # [n_active, 3*nb, 3*nb]
    block_rows: np.ndarray  # [n_active, nb] global block-row indices
    active: np.ndarray     # [n_elem] bool


def dof_layout(pot: PotentialDesc):
    """Local DoF order: for each dof set (in set order), the bindings on that set in binding order."""
    order = []
    for s in sorted({b.dof_set for b in pot.bindings if b.dof_set >= 0}):
        for bi, b in enumerate(pot.bindings):
            if b.dof_set == s:
                order.append(bi)
    return order


def evaluate_potential(prob: Problem, pot: PotentialDesc) -> ElementOutput | None:
    n_elem = pot.conn.shape[0]
    if n_elem == 0:
        return None
    order = dof_layout(pot)
    nb = len(order)
    n = 3 * nb
    local_of_binding = {bi: k for k, bi in enumerate(order)}
    inputs = []
    for bi, b in enumerate(pot.bindings):
        data = prob.arrays[b.array]
        vals = data[pot.conn[:, b.conn]] if b.conn >= 0 else np.broadcast_to(data[0], (n_elem, b.stride))
        if bi in local_of_binding:
            k = local_of_binding[bi]
            inputs.append([D2.var(vals[:, c], 3 * k + c, n) for c in range(b.stride)])
        else:
            inputs.append([np.ascontiguousarray(vals[:, c]) for c in range(b.stride)])
    out = REGISTRY[pot.name](inputs)
    active = np.ones(n_elem, dtype=bool)
    if isinstance(out, tuple):  # (energy, condition): evaluated only where condition > 0 (SecondOrderCompiledPotential.cpp:185-197)
        out, cond = out
        cv = cond.v if isinstance(cond, D2) else np.broadcast_to(cond, (n_elem,))
        active = cv > 0
    rows = np.stack([prob.dof_offsets[pot.bindings[bi].dof_set] // 3 + pot.conn[:, pot.bindings[bi].conn] for bi in order], axis=1)
    return ElementOutput(pot.name, out.v[active], out.g[active], out.h[active], rows[active].astype(np.int64), active)


def evaluate_all(prob: Problem):
    """E, grad, list of element outputs (SecondOrderCompiledGlobal.cpp:119-142)."""
    E = 0.0
    grad = np.zeros(prob.ndofs)
    outs = []
    for pot in prob.potentials:
        o = evaluate_potential(prob, pot)
        if o is None:
            continue
        E += float(np.sum(o.E))
        nb = o.block_rows.shape[1]
        idx = (3 * o.block_rows[:, :, None] + np.arange(3)[None, None, :]).reshape(len(o.E), 3 * nb)
        np.add.at(grad, idx.reshape(-1), o.g.reshape(-1))
        outs.append(o)
    return E, grad, outs


def evaluate_energy(prob: Problem):
    return evaluate_all(prob)[0]


# ----------------------------------------------------------------------------------------------------------------------
# PSD projection (symx/src/solver/second_order/project_to_PD.cpp:12-32)
# ----------------------------------------------------------------------------------------------------------------------
def project_to_pd(H, eps=1e-10, mirroring=False):
    """Returns (H_projected, changed[bool per element]). Matrices with no eigenvalue < eps are returned untouched."""
    w, V = np.linalg.eigh(H)
    changed = (w < eps).any(axis=1)
    w2 = np.where(w < eps, -w if mirroring else eps, w)
    Hp = np.einsum("eik,ek,ejk->eij", V, w2, V)
    out = H.copy()
    out[changed] = Hp[changed]
    return out, changed


# ----------------------------------------------------------------------------------------------------------------------
# Assembly into 3x3-blocked CSR with float storage (ElementHessians.cpp:224-256, BlockedSparseMatrix.h:781-895,380-593)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class BSR:
    n_block_rows: int
    row_ptr: np.ndarray     # int64 [nbr+1]
    cols: np.ndarray        # int32 [nnzb] block column
    vals: np.ndarray        # float32 [nnzb, 3, 3] (row-major within block here; the reference stores column-major)

    def to_scipy(self):
        return sp.bsr_matrix((self.vals.astype(np.float64), self.cols, self.row_ptr), shape=(3 * self.n_block_rows, 3 * self.n_block_rows))

    def spmv(self, x):
        # float storage, double arithmetic (BlockedSparseMatrix.h:986-1138)
        return self.to_scipy() @ x

    def diag_blocks(self):
        d = np.zeros((self.n_block_rows, 3, 3), dtype=np.float32)
        rows = np.repeat(np.arange(self.n_block_rows), np.diff(self.row_ptr))
        m = rows == self.cols
        d[rows[m]] = self.vals[m]
        return d


def assemble(outs, ndofs, hessians=None) -> BSR:
    """Sum of all element blocks per (block row, block col). Accumulated in double and rounded once to float; the
    reference accumulates in float in a thread-dependent order (tolerance: float eps * contributions)."""
    nbr = ndofs // 3
    keys, blocks = [], []
    for oi, o in enumerate(outs):
        H = o.H if hessians is None else hessians[oi]
        nb = o.block_rows.shape[1]
        m = len(o.E)
        if m == 0:
            continue
        Hb = H.reshape(m, nb, 3, nb, 3).transpose(0, 1, 3, 2, 4)  # [e, a, b, i, j]
        r = np.broadcast_to(o.block_rows[:, :, None], (m, nb, nb))
        c = np.broadcast_to(o.block_rows[:, None, :], (m, nb, nb))
        keys.append((r * nbr + c).reshape(-1))
        blocks.append(Hb.reshape(-1, 3, 3))
    keys = np.concatenate(keys)
    blocks = np.concatenate(blocks)
    uk, inv = np.unique(keys, return_inverse=True)
    acc = np.zeros((len(uk), 3, 3))
    np.add.at(acc, inv, blocks)
    rows = uk // nbr
    cols = (uk % nbr).astype(np.int32)
    row_ptr = np.zeros(nbr + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    return BSR(nbr, row_ptr, cols, acc.astype(np.float32))


def block_diag_inverse(A: BSR):
    """Closed-form symmetric 3x3 inverse in FLOAT with the reciprocal of the determinant via double
    (BlockedSparseMatrix.h:1198-1214)."""
    m = A.diag_blocks().astype(np.float32)
    f = np.float32
    tmp0 = m[:, 1, 1] * m[:, 2, 2]
    tmp1 = m[:, 1, 2] * m[:, 1, 2]
    tmp2 = m[:, 0, 2] * m[:, 1, 2]
    tmp3 = m[:, 0, 1] * m[:, 0, 1]
    tmp4 = m[:, 0, 2] * m[:, 0, 2]
    det = (m[:, 0, 0] * tmp0 - m[:, 0, 0] * tmp1 + f(2) * m[:, 0, 1] * tmp2 - m[:, 1, 1] * tmp4 - m[:, 2, 2] * tmp3).astype(np.float32)
    with np.errstate(divide="ignore"):
        tmp5 = (1.0 / det.astype(np.float64)).astype(np.float32)
    inv = np.zeros_like(m)
    inv[:, 2, 2] = tmp5 * (m[:, 0, 0] * m[:, 1, 1] - tmp3)
    inv[:, 1, 1] = tmp5 * (m[:, 0, 0] * m[:, 2, 2] - tmp4)
    inv[:, 0, 0] = tmp5 * (tmp0 - tmp1)
    inv[:, 1, 0] = -tmp5 * (m[:, 0, 1] * m[:, 2, 2] - tmp2)
    inv[:, 0, 1] = inv[:, 1, 0]
    inv[:, 2, 0] = tmp5 * (m[:, 0, 1] * m[:, 1, 2] - m[:, 1, 1] * m[:, 0, 2])
    inv[:, 0, 2] = inv[:, 2, 0]
    inv[:, 2, 1] = -tmp5 * (m[:, 0, 0] * m[:, 1, 2] - m[:, 0, 1] * m[:, 0, 2])
    inv[:, 1, 2] = inv[:, 2, 1]
    return inv


def apply_preconditioner(dinv, r):
    # float block x double vector (BlockedSparseMatrix.h:1315-1360)
    return np.einsum("bij,bj->bi", dinv.astype(np.float64), r.reshape(-1, 3)).reshape(-1)


# ----------------------------------------------------------------------------------------------------------------------
# PCG (BlockedSparseMatrix/solve_pcg.h:83-232)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class PCGInfo:
    converged: bool
    n_iterations: int
    error: float
    found_indefiniteness: bool


def solve_pcg(A: BSR, b, abs_tol, rel_tol=1e-4, max_iter=10000, stop_on_indef=True, dinv=None):
    n = len(b)
    S = A.to_scipy().tocsr()
    if dinv is None:
        dinv = block_diag_inverse(A)
    x = np.zeros(n)
    b_norm_sq = float(b @ b)
    if b_norm_sq < abs_tol * abs_tol:
        return x, PCGInfo(True, 0, 0.0, False)
    r = b - S @ x
    error = np.sqrt(float(r @ r) / b_norm_sq)
    error0 = error
    if error < abs_tol:
        return x, PCGInfo(True, 0, error, False)
    z = apply_preconditioner(dinv, r)
    p = z.copy()
    rz = float(r @ z)
    indef = False
    for it in range(1, max_iter + 1):
        Ap = S @ p
        pAp = float(p @ Ap)
        if pAp <= 0.0:
            indef = True
            if stop_on_indef:
                return x, PCGInfo(False, it, error, True)
        alpha = rz / pAp
        x += alpha * p
        r -= alpha * Ap
        error = np.sqrt(float(r @ r) / b_norm_sq)
        if error < abs_tol or error / error0 < rel_tol:
            return x, PCGInfo(True, it, error, indef)
        z = apply_preconditioner(dinv, r)
        rz_old = rz
        rz = float(r @ z)
        p = z + (rz / rz_old) * p
    return x, PCGInfo(False, max_iter, error, indef)


def forcing_abs_tol(residual, cg_abs_tolerance=1e-12):
    # NewtonsMethod.cpp:423-424
    return max(min(1e-2, residual * min(0.5, np.sqrt(residual))), cg_abs_tolerance)


# ----------------------------------------------------------------------------------------------------------------------
# Newton's method with progressive projection (symx/src/solver/NewtonsMethod.cpp:28-252,254-386,459-641)
# Contact-free subset: no validity callbacks, no step cap.
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class NewtonSettings:
    residual_tolerance_abs: float = 1e-6
    step_tolerance: float = 1e-3
    bailout_residual: float = 1e-10
    max_iterations: int = 1000
    projection_eps: float = 1e-10
    mirroring: bool = False
    ppn_tightening_factor: float = 0.5
    ppn_release_factor: float = 2.0
    cg_rel_tolerance: float = 1e-4
    cg_abs_tolerance: float = 1e-12
    cg_max_iterations: int = 10000
    armijo_beta: float = 1e-4
    max_armijo: int = 20
    projection_mode: str = "Progressive"


@dataclass
class NewtonStats:
    newton_iterations: int = 0
    cg_iterations: int = 0
    ls_bt_iterations: int = 0
    n_hessians: int = 0
    n_projected: int = 0
    result: str = "Running"


def newton_solve(prob: Problem, settings: NewtonSettings = NewtonSettings(), on_iterate=None):
    st = NewtonStats()
    ndofs = prob.ndofs
    nbr = ndofs // 3
    ppn_threshold = -1.0
    it = -1
    result = "Running"
    while result == "Running":
        it += 1
        if it == settings.max_iterations:
            result = "TooManyIterations"
            break
        E0, grad, outs = evaluate_all(prob)
        res = float(np.abs(grad).max())
        if res < settings.bailout_residual or res < settings.residual_tolerance_abs:
            result = "Successful"
            break
        hess = [o.H.copy() for o in outs]
        projected = [np.zeros(len(o.E), dtype=bool) for o in outs]
        A = None
        ok = False
        while not ok:
            all_projected = False
            if settings.projection_mode == "Progressive":
                if A is None:
                    A = assemble(outs, ndofs, hess)
                if ppn_threshold > 0.0:
                    if ppn_threshold < 1e-12:
                        ppn_threshold = 0.0
                    active = np.abs(grad).reshape(nbr, 3).max(axis=1) >= ppn_threshold
                    all_projected = bool(active.all())
                    for oi, o in enumerate(outs):
                        sel = (~projected[oi]) & active[o.block_rows].any(axis=1)
                        if sel.any():
                            hp, _ = project_to_pd(hess[oi][sel], settings.projection_eps, settings.mirroring)
                            hess[oi][sel] = hp
                            projected[oi][sel] = True
                    A = assemble(outs, ndofs, hess)
            elif settings.projection_mode == "ProjectedNewton":
                for oi in range(len(outs)):
                    hess[oi], _ = project_to_pd(hess[oi], settings.projection_eps, settings.mirroring)
                    projected[oi][:] = True
                all_projected = True
                A = assemble(outs, ndofs, hess)
            else:
                A = assemble(outs, ndofs, hess)
            du, info = solve_pcg(A, -grad, forcing_abs_tol(res, settings.cg_abs_tolerance), settings.cg_rel_tolerance, settings.cg_max_iterations)
            st.cg_iterations += info.n_iterations
            can_more = settings.projection_mode != "Newton" and not all_projected
            descends = False
            if not info.converged:
                if not can_more:
                    result = "LinearSystemSolveFailure"
                    break
            else:
                dg = float(du @ grad)
                descends = dg < 0.0
                if not descends and not can_more:
                    result = "StepDoesNotDescend"
                    break
            if info.converged and descends:
                ok = True
                break
            if ppn_threshold < 0.0:
                ppn_threshold = float(np.abs(grad).max())
            ppn_threshold *= settings.ppn_tightening_factor
        if result != "Running":
            break
        ppn_threshold *= settings.ppn_release_factor
        st.n_hessians += sum(len(o.E) for o in outs)
        st.n_projected += int(sum(p.sum() for p in projected))
        du_max = float(np.abs(du).max())
        if du_max < settings.step_tolerance:
            result = "Successful"
            break
        # line search (NewtonsMethod.cpp:459-641), Armijo only
        u0 = prob.get_dofs()
        step = 1.0
        prob.set_dofs(u0 + step * du)
        expected = settings.armijo_beta * dg
        k = 0
        while k < settings.max_armijo:
            E1 = evaluate_energy(prob)
            if E1 < E0 + expected * step:
                break
            step *= 0.5
            prob.set_dofs(u0 + step * du)
            st.ls_bt_iterations += 1
            k += 1
        if k == settings.max_armijo:
            result = "TooManyArmijoIterations"
            break
        if on_iterate is not None:
            on_iterate(prob.get_dofs())
    st.newton_iterations = it
    st.result = result
    return st


# ----------------------------------------------------------------------------------------------------------------------
# Time stepping for contact-free deformable scenes (stark/src/core/Stark.cpp:133-244;
# stark/src/models/deformables/PointDynamics.cpp:58-78)
# ----------------------------------------------------------------------------------------------------------------------
def point_state_arrays(prob: Problem):
    """(v1, x0, v0) array indices, read off the EnergyLumpedInertia bindings (EnergyLumpedInertia.cpp:17-21)."""
    for p in prob.potentials:
        if p.name == "EnergyLumpedInertia":
            return p.bindings[0].array, p.bindings[1].array, p.bindings[2].array
    raise RuntimeError("no EnergyLumpedInertia potential")


def run_time_step(prob: Problem, settings: NewtonSettings = NewtonSettings(), on_eval_point=None):
    iv1, ix0, iv0 = point_state_arrays(prob)
    prob.arrays[iv1] = np.zeros_like(prob.arrays[iv1])          # before_time_step: v1 <- 0
    if on_eval_point is not None:
        on_eval_point(prob.get_dofs())
    st = newton_solve(prob, settings, on_iterate=on_eval_point)
    if st.result == "Successful":                                # on_time_step_accepted
        prob.arrays[ix0] = prob.arrays[ix0] + prob.dt * prob.arrays[iv1]
        prob.arrays[iv0] = prob.arrays[iv1].copy()
    return st
