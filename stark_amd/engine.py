"""Thin object wrapper over the C ABI: mirrors the reference's registration calls
(symx::GlobalPotential::add_dof / add_potential, symx/src/solver/GlobalPotential.h:37-70) one to one."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class EngineError(RuntimeError):
    pass


class Engine:
    def __init__(self, device: int = 0):
        self.L = capi.lib()
        h = C.c_void_p()
        rc = self.L.mistark_create(device, C.byref(h))
        if rc != 0:
            raise EngineError("mistark_create failed (%d): no MI355X visible / HIP runtime error; the hot path has no CPU fallback" % rc)
        self.h = h
        self._keep = []          # host buffers must outlive the context (caller-owned memory contract)
        self._array_ids = {}
        self.potential_ids = {}

    def close(self):
        if self.h:
            self.L.mistark_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise EngineError(self.L.mistark_last_error(self.h).decode())
        return rc

    # ---- registration ------------------------------------------------------------------------------------------------
    def add_dof_set(self, label: str, arr: np.ndarray) -> int:
        assert arr.dtype == np.float64 and arr.flags.c_contiguous
        self._keep.append(arr)
        return self._ck(self.L.mistark_add_dof_set(self.h, label.encode(), arr.ctypes.data, arr.size))

    def array(self, arr: np.ndarray, stride: int | None = None) -> int:
        assert arr.dtype == np.float64 and arr.flags.c_contiguous
        if stride is None:
            stride = 1 if arr.ndim == 1 else arr.shape[-1]
        n_items = arr.size // stride
        self._keep.append(arr)
        return self._ck(self.L.mistark_array(self.h, arr.ctypes.data, n_items, stride))

    def potential(self, name: str, conn: np.ndarray, bindings) -> int:
        """bindings: list of (array_id, stride, conn_col) in the reference's mws.make_* order."""
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        if conn.ndim != 2:
            raise ValueError("conn must be [n_elem, stride]")
        b = (capi.Binding * len(bindings))(*[capi.Binding(a, s, c) for a, s, c in bindings])
        pid = self._ck(self.L.mistark_potential(self.h, name.encode(), conn.ctypes.data if conn.size else None, conn.shape[0], conn.shape[1], b, len(bindings)))
        self.potential_ids[name] = pid
        return pid

    def potential_custom(self, name: str, conn: np.ndarray, bindings, ops, constants, n_inputs: int, cond_ops=None, cond_constants=None) -> int:
        """A potential given as SymX's op sequence (rows {type, dst, a, b, cond} + one constant per op), interpreted on the device."""
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        b = (capi.Binding * len(bindings))(*[capi.Binding(a, s, c) for a, s, c in bindings])
        ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 5)
        cst = np.ascontiguousarray(constants, dtype=np.float64)
        nco = 0
        cops = ccst = None
        if cond_ops is not None and len(cond_ops):
            cops = np.ascontiguousarray(cond_ops, dtype=np.int32).reshape(-1, 5)
            ccst = np.ascontiguousarray(cond_constants, dtype=np.float64)
            nco = len(cops)
        self.L.mistark_potential_custom.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(capi.Binding), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        pid = self._ck(self.L.mistark_potential_custom(self.h, name.encode(), conn.ctypes.data if conn.size else None, conn.shape[0], conn.shape[1], b, len(bindings),
                                                       ops.ctypes.data, cst.ctypes.data, len(ops), int(n_inputs), cops.ctypes.data if nco else None,
                                                       ccst.ctypes.data if nco else None, nco))
        self.potential_ids[name] = pid
        return pid

    def potential_custom_set_summation(self, pid: int, first_input: int, data: np.ndarray):
        """Summation loop of a custom potential (MappedWorkspace::add_for_each): inputs [first_input, first_input + data.shape[1]) take the rows
        of `data` one after the other; energy, gradient and Hessian are summed over the rows."""
        d = np.ascontiguousarray(data, dtype=np.float64).reshape(len(data), -1)
        self.L.mistark_potential_custom_set_summation.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        self._ck(self.L.mistark_potential_custom_set_summation(self.h, pid, int(first_input), d.shape[1], d.shape[0], d.ctypes.data))

    def potential_id(self, name: str) -> int:
        pid = self.L.mistark_find_potential(self.h, name.encode())
        if pid < 0:
            raise KeyError(name)
        return pid

    def set_dynamic(self, pid: int, dynamic: bool = True):
        """Routes the potential's Hessian blocks to the dynamic (contact) part of the split matrix."""
        self._ck(self.L.mistark_potential_set_dynamic(self.h, pid, int(dynamic)))

    def update_connectivity(self, pid: int, conn: np.ndarray):
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        self._ck(self.L.mistark_potential_update_connectivity(self.h, pid, conn.ctypes.data if conn.size else None, conn.shape[0]))

    # ---- multi-GPU sharding ------------------------------------------------------------------------------------------------
    def dist_init_local(self, group, rank: int):
        self._ck(self.L.mistark_dist_init_local(self.h, group, rank))

    def dist_init_ipc(self, comm):
        """comm: a connected capi.IpcComm (one process per rank; include/mistark.h "IPC windows")."""
        self._comm = comm  # (must outlive the engine)
        self._ck(self.L.mistark_dist_init_ipc(self.h, comm.h))

    def dist_info(self):
        out = (C.c_int64 * 8)()
        self._ck(self.L.mistark_dist_info(self.h, out, 8))
        return list(out)

    def dist_row_owner(self) -> np.ndarray:
        o = np.zeros(self.ndofs // 3, dtype=np.int32)
        self._ck(self.L.mistark_dist_get_row_owner(self.h, o.ctypes.data))
        return o

    def dist_set_row_owner(self, owner):
        o = np.ascontiguousarray(owner, dtype=np.int32)
        self._ck(self.L.mistark_dist_set_row_owner(self.h, o.ctypes.data, len(o)))

    def dist_init_rccl(self, rank: int, world: int, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(self.L.mistark_dist_init_rccl(self.h, rank, world, buf))

    # ---- device contact detector (include/mistark_contact.h) ----------------------------------------------------------------
    def contact_init(self, **array_ids):
        """array_ids: engine array ids by role (capi.CONTACT_ROLES); absent roles are -1."""
        a = capi.ContactArrays(*[int(array_ids.get(r, -1)) for r in capi.CONTACT_ROLES])
        self._ck(self.L.mistark_contact_init(self.h, C.byref(a)))

    def contact_add_mesh(self, kind: str, idx_in_ps: int, verts, tris, edges) -> int:
        v = np.ascontiguousarray(verts, dtype=np.int32)
        t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        return self._ck(self.L.mistark_contact_add_mesh(self.h, 0 if kind == "d" else 1, idx_in_ps, v.ctypes.data, len(v), t.ctypes.data if t.size else None, len(t),
                                                        e.ctypes.data if e.size else None, len(e)))

    def contact_set_friction(self, a, b, mu):
        self._ck(self.L.mistark_contact_set_friction(self.h, a, b, mu))

    def contact_disable_collision(self, a, b):
        self._ck(self.L.mistark_contact_disable_collision(self.h, a, b))

    def contact_set_broad_phase(self, brute_force: bool):
        self._ck(self.L.mistark_contact_set_broad_phase(self.h, int(brute_force)))

    def contact_update(self, dt) -> int:
        n = C.c_int64()
        self._ck(self.L.mistark_contact_update(self.h, dt, C.byref(n)))
        return n.value

    def contact_update_friction(self) -> int:
        n = C.c_int64()
        self._ck(self.L.mistark_contact_update_friction(self.h, C.byref(n)))
        return n.value

    def contact_count_intersections(self, dt) -> int:
        n = C.c_int64()
        self._ck(self.L.mistark_contact_count_intersections(self.h, dt, C.byref(n)))
        return n.value

    def contact_table(self, name: str) -> np.ndarray:
        n, st = C.c_int32(), C.c_int32()
        self._ck(self.L.mistark_contact_get_table(self.h, name.encode(), None, C.byref(n), C.byref(st)))
        out = np.zeros((n.value, st.value), dtype=np.int32)
        if n.value:
            self._ck(self.L.mistark_contact_get_table(self.h, name.encode(), out.ctypes.data, C.byref(n), C.byref(st)))
        return out

    def contact_friction_data(self, name: str, n_rows: int) -> dict:
        nb = C.c_int32()
        self._ck(self.L.mistark_contact_get_friction_data(self.h, name.encode(), None, None, None, None, C.byref(nb)))
        T = np.zeros((n_rows, 6)); mu = np.zeros(n_rows); fn = np.zeros(n_rows); bary = np.zeros((n_rows, max(nb.value, 1)))
        self._ck(self.L.mistark_contact_get_friction_data(self.h, name.encode(), T.ctypes.data, mu.ctypes.data, fn.ctypes.data, bary.ctypes.data if nb.value else None, C.byref(nb)))
        out = dict(T=T, mu=mu, fn=fn)
        if nb.value:
            out["bary"] = bary
        return out

    def contact_vertices(self) -> np.ndarray:
        n = C.c_int64()
        self._ck(self.L.mistark_contact_get_vertices(self.h, None, C.byref(n)))
        x = np.zeros((n.value, 3))
        self._ck(self.L.mistark_contact_get_vertices(self.h, x.ctypes.data, C.byref(n)))
        return x

    def upload(self, array: int = -1):
        self._ck(self.L.mistark_upload(self.h, array))

    def download(self, array: int = -1):
        self._ck(self.L.mistark_download(self.h, array))

    def axpby(self, dst, a, x, b=0.0, y=-1):
        self._ck(self.L.mistark_array_axpby(self.h, dst, a, x, b, y))

    def fill(self, dst, v):
        self._ck(self.L.mistark_array_fill(self.h, dst, v))

    # ---- DoFs ------------------------------------------------------------------------------------------------------------
    @property
    def ndofs(self) -> int:
        return int(self.L.mistark_ndofs(self.h))

    def get_dofs(self) -> np.ndarray:
        u = np.zeros(self.ndofs)
        self._ck(self.L.mistark_get_dofs(self.h, u.ctypes.data))
        return u

    def set_dofs(self, u: np.ndarray):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.size == self.ndofs
        self._ck(self.L.mistark_set_dofs(self.h, u.ctypes.data))

    def dofs_to_host(self):
        self._ck(self.L.mistark_dofs_to_host_arrays(self.h))

    def dofs_from_host(self):
        self._ck(self.L.mistark_dofs_from_host_arrays(self.h))

    # ---- stages ------------------------------------------------------------------------------------------------------------
    def eval(self, mode=capi.EVAL_P_G_H, want_grad=True):
        E = C.c_double()
        g = np.zeros(self.ndofs) if (want_grad and mode != capi.EVAL_P) else None
        self._ck(self.L.mistark_eval(self.h, mode, C.byref(E), g.ctypes.data if g is not None else None))
        return E.value, g

    def element_hessians(self, pot: int, n_elem: int):
        nb = C.c_int32()
        self._ck(self.L.mistark_get_element_hessians(self.h, pot, None, None, C.byref(nb)))
        n = 3 * nb.value
        H = np.zeros((n_elem, n, n))
        rows = np.zeros((n_elem, nb.value), dtype=np.int32)
        self._ck(self.L.mistark_get_element_hessians(self.h, pot, H.ctypes.data, rows.ctypes.data, C.byref(nb)))
        return H, rows

    def element_energies(self, pot: int, n_elem: int):
        E = np.zeros(n_elem)
        self._ck(self.L.mistark_get_element_energies(self.h, pot, E.ctypes.data))
        return E

    def direct_llt(self, rhs):
        """x = A^-1 rhs by the device Cholesky (mistark_direct_llt_rhs); returns (x, success)."""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros(self.ndofs)
        ok = C.c_int()
        self._ck(self.L.mistark_direct_llt_rhs(self.h, rhs.ctypes.data, x.ctypes.data, C.byref(ok)))
        return x, bool(ok.value)

    def project(self, eps=1e-10, mirroring=False, active_blocks=None):
        npj, nch = C.c_int64(), C.c_int64()
        ab = None
        if active_blocks is not None:
            ab = np.ascontiguousarray(active_blocks, dtype=np.uint8)
        self._ck(self.L.mistark_project(self.h, eps, int(mirroring), ab.ctypes.data if ab is not None else None, C.byref(npj), C.byref(nch)))
        return npj.value, nch.value

    def assemble(self):
        self._ck(self.L.mistark_assemble(self.h))

    def get_bsr(self, with_vals=True):
        nbr, nnzb = C.c_int64(), C.c_int64()
        self._ck(self.L.mistark_get_bsr(self.h, C.byref(nbr), C.byref(nnzb), None, None, None))
        row_ptr = np.zeros(nbr.value + 1, dtype=np.int64)
        cols = np.zeros(nnzb.value, dtype=np.int32)
        vals = np.zeros((nnzb.value, 3, 3), dtype=np.float32)
        self._ck(self.L.mistark_get_bsr(self.h, C.byref(nbr), C.byref(nnzb), row_ptr.ctypes.data, cols.ctypes.data, vals.ctypes.data if with_vals else None))
        return row_ptr, cols, vals

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self._ck(self.L.mistark_spmv(self.h, x.ctypes.data, y.ctypes.data))
        return y

    def apply_preconditioner(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        z = np.zeros_like(x)
        self._ck(self.L.mistark_apply_preconditioner(self.h, x.ctypes.data, z.ctypes.data))
        return z

    def pcg(self, abs_tol, rel_tol=1e-4, max_iter=10000, stop_on_indef=True, rhs=None):
        info = capi.PcgInfo()
        x = np.zeros(self.ndofs)
        if rhs is None:
            self._ck(self.L.mistark_pcg(self.h, abs_tol, rel_tol, max_iter, int(stop_on_indef), x.ctypes.data, C.byref(info)))
        else:
            rhs = np.ascontiguousarray(rhs, dtype=np.float64)
            self._ck(self.L.mistark_pcg_rhs(self.h, rhs.ctypes.data, abs_tol, rel_tol, max_iter, int(stop_on_indef), x.ctypes.data, C.byref(info)))
        return x, info

    def default_newton_settings(self):
        s = capi.NewtonSettings()
        self.L.mistark_newton_default_settings(C.byref(s))
        return s

    def newton_solve(self, settings=None, callbacks=None):
        st = capi.NewtonStats()
        s = settings if settings is not None else self.default_newton_settings()
        rc = self._ck(self.L.mistark_newton_solve(self.h, C.byref(s), C.byref(callbacks) if callbacks is not None else None, C.byref(st)))
        return capi.SOLVER_RETURN[rc], st

    def newton_iteration_log(self):
        """Per-iteration records of the last newton_solve (capi.NewtonIteration)."""
        n = C.c_int32()
        self._ck(self.L.mistark_newton_iteration_log(self.h, None, 0, C.byref(n)))
        rec = (capi.NewtonIteration * max(n.value, 1))()
        self._ck(self.L.mistark_newton_iteration_log(self.h, rec, n.value, C.byref(n)))
        return list(rec)[:n.value]

    def counter(self, name: str) -> int:
        """mistark_get_counter: event counters of the context (tests assert that a feature under test actually ran)."""
        v = C.c_int64()
        self._ck(self.L.mistark_get_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, name: str, value: int):
        self._ck(self.L.mistark_set_option(self.h, name.encode(), int(value)))

    def spmv_timing(self, reset=0):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self.L.mistark_spmv_timing(self.h, reset, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value
