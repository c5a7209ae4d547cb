"""ctypes binding of libmistark.so (include/mistark.h). The library is built in-tree by `__graft_entry__.build()` /
`make -C stark_amd/csrc`; there is NO CPU fallback: loading fails loudly if the shared object is missing, and
`Engine()` fails loudly if no MI355X is visible."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmistark.so")

EVAL_P, EVAL_P_G, EVAL_P_G_H = 0, 1, 2
PROJ_NEWTON, PROJ_PROJECTED_NEWTON, PROJ_ON_DEMAND, PROJ_PROGRESSIVE = 0, 1, 2, 3
SOLVER_RETURN = ["Successful", "Running", "InvalidInitialState", "TooManyIterations", "TooManyArmijoIterations", "LinearSystemSolveFailure",
                 "TooManyInvalidIntermediateIterations", "StepDoesNotDescend", "InvalidConvergedState"]


class Binding(C.Structure):
    _fields_ = [("array", C.c_int32), ("stride", C.c_int32), ("conn_col", C.c_int32)]


CONTACT_ROLES = ["v1", "x0", "X", "dt", "k", "thickness", "epsv", "rb_xloc", "rb_v1", "rb_w1", "rb_t0", "rb_q0"]


class ContactArrays(C.Structure):
    """include/mistark_contact.h: mistark_contact_arrays (engine array ids, -1 = absent)."""
    _fields_ = [(r, C.c_int32) for r in CONTACT_ROLES]


class PcgInfo(C.Structure):
    _fields_ = [("converged", C.c_int32), ("n_iterations", C.c_int32), ("found_indefiniteness", C.c_int32), ("reserved", C.c_int32), ("error", C.c_double)]


class NewtonSettings(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("min_iterations", C.c_int32), ("residual_tolerance_abs", C.c_double), ("residual_tolerance_rel", C.c_double),
        ("step_tolerance", C.c_double), ("max_iterations_as_success", C.c_int32), ("step_cap", C.c_double), ("enable_armijo_backtracking", C.c_int32),
        ("line_search_armijo_beta", C.c_double), ("max_backtracking_armijo_iterations", C.c_int32), ("max_backtracking_invalid_state_iterations", C.c_int32),
        ("projection_mode", C.c_int32), ("projection_eps", C.c_double), ("project_to_pd_use_mirroring", C.c_int32), ("project_on_demand_countdown", C.c_int32),
        ("ppn_tightening_factor", C.c_double), ("ppn_release_factor", C.c_double), ("cg_max_iterations", C.c_int32), ("cg_abs_tolerance", C.c_double),
        ("cg_rel_tolerance", C.c_double), ("cg_stop_on_indefiniteness", C.c_int32), ("bailout_residual", C.c_double), ("linear_solver", C.c_int32),
    ]


class NewtonStats(C.Structure):
    _fields_ = [
        ("newton_iterations", C.c_int32), ("cg_iterations", C.c_int32), ("ls_cap_iterations", C.c_int32), ("ls_max_iterations", C.c_int32),
        ("ls_inv_iterations", C.c_int32), ("ls_bt_iterations", C.c_int32), ("n_hessians", C.c_int64), ("n_projected_hessians", C.c_int64),
        ("projected_hessians_ratio", C.c_double), ("n_linear_solves", C.c_int32), ("n_evaluations", C.c_int32), ("t_eval_pgh", C.c_double),
        ("t_eval_p", C.c_double), ("t_project", C.c_double), ("t_assembly", C.c_double), ("t_linear_solve", C.c_double), ("t_callbacks", C.c_double),
        ("t_total", C.c_double),
    ]


class NewtonIteration(C.Structure):
    """mistark_newton_iteration: one record per Newton iteration of the last solve (the reference's per-iteration Logger series)."""
    _fields_ = [
        ("residual", C.c_double), ("du_max", C.c_double), ("linear_solves", C.c_int32), ("cg_iterations_last", C.c_int32), ("cg_iterations_all", C.c_int32),
        ("logged", C.c_int32), ("n_hessians", C.c_int64), ("n_projected_hessians", C.c_int64), ("line_search", C.c_int32), ("ls_cap", C.c_int32),
        ("ls_max", C.c_int32), ("ls_inv", C.c_int32), ("ls_bt", C.c_int32), ("reserved", C.c_int32),
    ]


VOIDCB = C.CFUNCTYPE(None, C.c_void_p)
INTCB = C.CFUNCTYPE(C.c_int, C.c_void_p)
DBLCB = C.CFUNCTYPE(C.c_double, C.c_void_p)


class NewtonCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("before_energy_evaluation", VOIDCB), ("is_initial_state_valid", INTCB), ("is_intermediate_state_valid", INTCB),
                ("on_intermediate_state_invalid", VOIDCB), ("on_armijo_fail", VOIDCB), ("is_converged", INTCB), ("is_converged_state_valid", INTCB),
                ("max_allowed_step", DBLCB)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmistark.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the hot path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.mistark_version.restype = C.c_char_p
    L.mistark_last_error.restype = C.c_char_p
    L.mistark_last_error.argtypes = [p]
    L.mistark_create.argtypes = [C.c_int, C.POINTER(p)]
    L.mistark_destroy.argtypes = [p]
    L.mistark_destroy.restype = None
    L.mistark_supported_potential.restype = C.c_char_p
    L.mistark_supported_potential.argtypes = [C.c_int]
    L.mistark_add_dof_set.argtypes = [p, C.c_char_p, p, i64]
    L.mistark_resize_dof_set.argtypes = [p, C.c_int, p, i64]
    L.mistark_array.argtypes = [p, p, i64, C.c_int]
    L.mistark_array_rebind.argtypes = [p, C.c_int, p, i64]
    L.mistark_upload.argtypes = [p, C.c_int]
    L.mistark_download.argtypes = [p, C.c_int]
    L.mistark_array_axpby.argtypes = [p, C.c_int, dbl, C.c_int, dbl, C.c_int]
    L.mistark_array_fill.argtypes = [p, C.c_int, dbl]
    L.mistark_potential.argtypes = [p, C.c_char_p, p, i32, i32, C.POINTER(Binding), i32]
    L.mistark_ndofs.argtypes = [p]
    L.mistark_ndofs.restype = i64
    L.mistark_get_dofs.argtypes = [p, p]
    L.mistark_set_dofs.argtypes = [p, p]
    L.mistark_dofs_to_host_arrays.argtypes = [p]
    L.mistark_get_counter.argtypes = [p, C.c_char_p, C.POINTER(i64)]
    L.mistark_dofs_from_host_arrays.argtypes = [p]
    L.mistark_eval.argtypes = [p, C.c_int, C.POINTER(dbl), p]
    L.mistark_get_element_hessians.argtypes = [p, C.c_int, p, p, C.POINTER(i32)]
    L.mistark_get_element_energies.argtypes = [p, C.c_int, p]
    L.mistark_project.argtypes = [p, dbl, C.c_int, p, C.POINTER(i64), C.POINTER(i64)]
    L.mistark_project_by_gradient.argtypes = [p, dbl, C.c_int, dbl, C.POINTER(C.c_int), C.POINTER(i64)]
    L.mistark_assemble.argtypes = [p]
    L.mistark_potential_set_dynamic.argtypes = [p, C.c_int, C.c_int]
    L.mistark_find_potential.argtypes = [p, C.c_char_p]
    L.mistark_potential_update_connectivity.argtypes = [p, C.c_int, p, C.c_int32]
    L.mistark_contact_init.argtypes = [p, C.POINTER(ContactArrays)]
    L.mistark_contact_add_mesh.argtypes = [p, C.c_int, C.c_int, p, C.c_int32, p, C.c_int32, p, C.c_int32]
    L.mistark_contact_set_friction.argtypes = [p, C.c_int, C.c_int, C.c_double]
    L.mistark_contact_disable_collision.argtypes = [p, C.c_int, C.c_int]
    L.mistark_contact_enable.argtypes = [p, C.c_int, C.c_int]
    L.mistark_contact_set_broad_phase.argtypes = [p, C.c_int]
    L.mistark_contact_update.argtypes = [p, C.c_double, C.POINTER(i64)]
    L.mistark_contact_update_friction.argtypes = [p, C.POINTER(i64)]
    L.mistark_contact_count_intersections.argtypes = [p, C.c_double, C.POINTER(i64)]
    L.mistark_contact_get_table.argtypes = [p, C.c_char_p, p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mistark_contact_get_friction_data.argtypes = [p, C.c_char_p, p, p, p, p, C.POINTER(C.c_int32)]
    L.mistark_contact_get_vertices.argtypes = [p, p, C.POINTER(i64)]
    L.mistark_contact_recipe.argtypes = [C.c_char_p, C.POINTER(C.c_int32), p, p, p]
    L.mistark_cd_create.argtypes = [C.POINTER(p), C.c_int]
    L.mistark_cd_destroy.argtypes = [p]
    L.mistark_cd_destroy.restype = None
    L.mistark_cd_last_error.argtypes = [p]
    L.mistark_cd_last_error.restype = C.c_char_p
    L.mistark_cd_add_mesh.argtypes = [p, p, i32, p, i32, p, i32]
    L.mistark_cd_add_blacklist.argtypes = [p, i32, i32]
    L.mistark_cd_activate.argtypes = [p, C.c_int, C.c_int]
    L.mistark_cd_run_proximity.argtypes = [p, C.c_double, p]
    L.mistark_cd_get_proximity.argtypes = [p, C.c_int, p, p]
    L.mistark_cd_run_intersection.argtypes = [p, C.POINTER(i32)]
    L.mistark_cd_get_intersections.argtypes = [p, p]
    L.mistark_shard_range.argtypes = [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.mistark_dist_unique_id.argtypes = [p]
    L.mistark_dist_init_rccl.argtypes = [p, C.c_int, C.c_int, p]
    L.mistark_dist_rccl_selftest.argtypes = [p, p, i64]
    L.mistark_local_group_create.argtypes = [C.c_int]
    L.mistark_local_group_create.restype = p
    L.mistark_local_group_destroy.argtypes = [p]
    L.mistark_local_group_destroy.restype = None
    L.mistark_dist_init_local.argtypes = [p, p, C.c_int]
    L.mistark_ipc_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, i64, p]
    L.mistark_ipc_comm_create.restype = p
    L.mistark_ipc_comm_connect.argtypes = [p, p, i64]
    L.mistark_ipc_comm_last_error.argtypes = [p]
    L.mistark_ipc_comm_last_error.restype = C.c_char_p
    L.mistark_ipc_comm_destroy.argtypes = [p]
    L.mistark_ipc_comm_destroy.restype = None
    L.mistark_ipc_comm_selftest.argtypes = [p, i64, C.c_int, C.POINTER(dbl)]  # double[2]
    L.mistark_ipc_comm_preflight.argtypes = [p, C.c_int, dbl, C.POINTER(dbl)]  # double[world]
    L.mistark_rccl_allreduce_bench.argtypes = [C.c_int, C.c_int, C.c_int, p, i64, C.c_int, C.POINTER(dbl), p, C.c_int]  # double[4], char[err_len]
    L.mistark_dist_init_ipc.argtypes = [p, p]
    L.mistark_dist_fused_bench.argtypes = [p, C.c_int, C.POINTER(dbl)]
    L.mistark_dist_set_row_owner.argtypes = [p, p, i64]
    L.mistark_dist_add_shared_rows.argtypes = [p, p, i64]
    L.mistark_dist_set_row_coords.argtypes = [p, p, i64]
    L.mistark_dist_info.argtypes = [p, p, C.c_int]
    L.mistark_dist_get_row_owner.argtypes = [p, p]
    L.mistark_dist_move.argtypes = [p, p]
    L.mistark_sync.argtypes = [p]
    L.mistark_spmv_event_overhead.argtypes = [p, C.POINTER(C.c_double)]
    L.mistark_spmv_device_clock.argtypes = [p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.mistark_dof_array.argtypes = [p, C.c_int, C.c_int]
    L.mistark_create_dry.argtypes = [C.POINTER(p)]
    L.mistark_describe.argtypes = [p, p, i64]
    L.mistark_describe.restype = i64
    L.mistark_get_bsr.argtypes = [p, C.POINTER(i64), C.POINTER(i64), p, p, p]
    L.mistark_spmv.argtypes = [p, p, p]
    L.mistark_apply_preconditioner.argtypes = [p, p, p]
    L.mistark_pcg.argtypes = [p, dbl, dbl, C.c_int, C.c_int, p, C.POINTER(PcgInfo)]
    L.mistark_pcg_rhs.argtypes = [p, p, dbl, dbl, C.c_int, C.c_int, p, C.POINTER(PcgInfo)]
    L.mistark_direct_llt_rhs.argtypes = [p, p, p, C.POINTER(C.c_int)]
    L.mistark_newton_default_settings.argtypes = [C.POINTER(NewtonSettings)]
    L.mistark_newton_default_settings.restype = None
    L.mistark_newton_solve.argtypes = [p, C.POINTER(NewtonSettings), C.POINTER(NewtonCallbacks), C.POINTER(NewtonStats)]
    L.mistark_newton_iteration_log.argtypes = [p, C.POINTER(NewtonIteration), C.c_int32, C.POINTER(C.c_int32)]
    L.mistark_set_option.argtypes = [p, C.c_char_p, C.c_int]
    L.mistark_spmv_bench.argtypes = [p, C.c_int, C.POINTER(dbl)]
    L.mistark_spmv_timing.argtypes = [p, C.c_int, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl)]
    _lib = L
    return L


class IpcComm:
    """One rank's end of the IPC-window transport (include/mistark.h "IPC windows"): create -> all-gather the 64-byte handles through the
    launcher -> connect. `allgather_bytes(b)` must return the list of every rank's bytes in rank order (torch.distributed.all_gather_object)."""

    def __init__(self, device, rank, world, window_bytes, allgather_bytes=None):
        """With `allgather_bytes` the communicator is connected on return; without it the caller exchanges `self.handle` (64 bytes) itself
        and calls connect(handles) — the two-phase form lets a launcher agree on a fallback when a rank's creation failed, before any rank
        waits in the handle exchange (bench.py)."""
        L = lib()
        buf = C.create_string_buffer(64)
        self.h = L.mistark_ipc_comm_create(device, rank, world, int(window_bytes), buf)
        if not self.h:
            raise RuntimeError("mistark_ipc_comm_create failed (rank %d)" % rank)
        self.rank, self.world = rank, world
        self.handle = buf.raw
        if allgather_bytes is not None:
            self.connect(allgather_bytes(self.handle))

    def connect(self, handles):
        if len(handles) != self.world or any(not isinstance(x, (bytes, bytearray)) or len(x) != 64 for x in handles):
            raise RuntimeError("IPC handles: expected %d x 64 bytes" % self.world)
        blob = b"".join(handles)
        if lib().mistark_ipc_comm_connect(self.h, blob, len(blob)) != 0:
            raise RuntimeError("mistark_ipc_comm_connect: %s" % lib().mistark_ipc_comm_last_error(self.h).decode())

    def selftest(self, n=1024, iters=20):
        """Collective. Every value checked; returns (wall time of one all-gather of n doubles + stream synchronisation, of one all-gather in a
        train enqueued back to back) in microseconds."""
        us = (C.c_double * 2)()
        if lib().mistark_ipc_comm_selftest(self.h, n, iters, us) != 0:
            raise RuntimeError("IPC self-test: %s" % lib().mistark_ipc_comm_last_error(self.h).decode())
        return us[0], us[1]

    def preflight(self, iters=16, timeout_s=2.0):
        """Collective (host barrier in front). One tagged granule over every ordered pair of ranks; returns (peers that answered, [half the best
        round trip in microseconds per peer; 0 for the own rank, None where nothing came back])."""
        us = (C.c_double * self.world)()
        n = lib().mistark_ipc_comm_preflight(self.h, iters, float(timeout_s), us)
        if n < 0:
            raise RuntimeError("IPC pre-flight: %s" % lib().mistark_ipc_comm_last_error(self.h).decode())
        return n, [None if v < 0 else float(v) for v in us]

    def close(self):
        if self.h:
            lib().mistark_ipc_comm_destroy(self.h)
            self.h = None


def exported_symbols():
    """Every entry point include/*.h declares (used by the CPU-side load test)."""
    import re
    out = set()
    for h in ("mistark.h", "mistark_contact.h", "mistark_sim.h", "mistark_tmcd.h"):
        hdr = open(os.path.join(os.path.dirname(_HERE), "include", h)).read()
        out |= set(re.findall(r"\b(mistark_[a-z0-9_]+)\s*\(", hdr))
        out -= set(re.findall(r"struct\s+(mistark_[a-z0-9_]+)", hdr))  # (type names mentioned in comments)
    return sorted(out)


class CollisionDetector:
    """include/mistark_tmcd.h: the detector on host positions (what a replacement of the reference's tmcd::ProximityDetection /
    tmcd::IntersectionDetection binds). Meshes keep a reference to the position arrays handed over: update them in place between runs."""
    LISTS = ("pt_point_point", "pt_point_edge", "pt_point_triangle", "ee_point_point", "ee_point_edge", "ee_edge_edge")
    COLS = (8, 9, 7, 10, 9, 8)

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.mistark_cd_create(C.byref(h), device)
        if rc != 0:
            raise RuntimeError("mistark_cd_create failed (%d): no usable GPU" % rc)
        self.h = h
        self._keep = []

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.mistark_cd_last_error(self.h).decode())
        return rc

    def add_mesh(self, x, triangles, edges):
        import numpy as np
        assert x.dtype == np.float64 and x.flags.c_contiguous
        t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        e = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        self._keep.append(x)
        return self._ck(self.L.mistark_cd_add_mesh(self.h, x.ctypes.data, len(x), t.ctypes.data if len(t) else None, len(t), e.ctypes.data if len(e) else None, len(e)))

    def add_blacklist(self, a, b):
        self._ck(self.L.mistark_cd_add_blacklist(self.h, a, b))

    def add_blacklist_range(self, edge_edge, mesh_a, interval_a, mesh_b, interval_b):
        """tmcd add_blacklist_range_point_triangle (edge_edge False: points interval_a of mesh_a x triangles interval_b of mesh_b) /
        add_blacklist_range_edge_edge (True: edges x edges, the lower interval first); half-open local intervals."""
        self.L.mistark_cd_add_blacklist_range.argtypes = [C.c_void_p] + [C.c_int32] * 7
        self._ck(self.L.mistark_cd_add_blacklist_range(self.h, int(bool(edge_edge)), mesh_a, interval_a[0], interval_a[1], mesh_b, interval_b[0], interval_b[1]))

    def activate(self, point_triangle=True, edge_edge=True):
        self._ck(self.L.mistark_cd_activate(self.h, int(point_triangle), int(edge_edge)))

    def run_proximity(self, enlargement):
        import numpy as np
        counts = (C.c_int32 * 6)()
        self._ck(self.L.mistark_cd_run_proximity(self.h, enlargement, counts))
        out = {}
        for l, name in enumerate(self.LISTS):
            rows = np.zeros((counts[l], self.COLS[l]), dtype=np.int32)
            dist = np.zeros(counts[l])
            if counts[l]:
                self._ck(self.L.mistark_cd_get_proximity(self.h, l, rows.ctypes.data, dist.ctypes.data))
            out[name] = (rows, dist)
        return out

    def run_broad_phase(self, enlargement):
        """(point-triangle pairs, edge-edge pairs): rows of (first.set, first.idx, second.set, second.idx) — tmcd get_broad_phase_results()."""
        import numpy as np
        self.L.mistark_cd_run_broad_phase.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int32)]
        self.L.mistark_cd_get_broad_phase.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        counts = (C.c_int32 * 2)()
        self._ck(self.L.mistark_cd_run_broad_phase(self.h, enlargement, counts))
        out = []
        for l in range(2):
            rows = np.zeros((counts[l], 4), dtype=np.int32)
            if counts[l]:
                self._ck(self.L.mistark_cd_get_broad_phase(self.h, l, rows.ctypes.data))
            out.append(rows)
        return tuple(out)

    def run_intersection(self):
        import numpy as np
        n = C.c_int32()
        self._ck(self.L.mistark_cd_run_intersection(self.h, C.byref(n)))
        rows = np.zeros((n.value, 9), dtype=np.int32)
        if n.value:
            self._ck(self.L.mistark_cd_get_intersections(self.h, rows.ctypes.data))
        return rows

    def close(self):
        if self.h:
            self.L.mistark_cd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
