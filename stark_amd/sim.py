"""ctypes binding of the scene-level C facade (include/mistark_sim.h) over the C++ host mirror of stark::Simulation."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class SimSettings(C.Structure):
    _fields_ = [("gravity", C.c_double * 3), ("max_time_step_size", C.c_double), ("use_adaptive_time_step", C.c_int32),
                ("time_step_size_success_multiplier", C.c_double), ("time_step_size_lower_bound", C.c_double), ("device", C.c_int32),
                ("mirror_state_to_host", C.c_int32), ("enable_output", C.c_int32), ("init_frictional_contact", C.c_int32), ("newton", capi.NewtonSettings),
                ("enable_frame_writes", C.c_int32), ("fps", C.c_int32), ("output_directory", C.c_char * 256), ("simulation_name", C.c_char * 64),
                ("allowed_execution_time", C.c_double), ("end_simulation_time", C.c_double), ("end_frame", C.c_int32)]


class VolumeParams(C.Structure):
    _fields_ = [("density", C.c_double), ("inertia_damping", C.c_double), ("quasistatic", C.c_int32), ("elasticity_only", C.c_int32), ("scale", C.c_double),
                ("youngs_modulus", C.c_double), ("poissons_ratio", C.c_double), ("strain_damping", C.c_double), ("strain_limit", C.c_double),
                ("strain_limit_stiffness", C.c_double)]


class SurfaceParams(C.Structure):
    _fields_ = [("density", C.c_double), ("inertia_damping", C.c_double), ("quasistatic", C.c_int32), ("elasticity_only", C.c_int32), ("scale", C.c_double),
                ("thickness", C.c_double), ("youngs_modulus", C.c_double), ("poissons_ratio", C.c_double), ("strain_damping", C.c_double),
                ("strain_limit", C.c_double), ("strain_limit_stiffness", C.c_double), ("inflation", C.c_double), ("bending_stiffness", C.c_double),
                ("bending_damping", C.c_double), ("flat_rest_angle", C.c_int32)]


class LineParams(C.Structure):
    _fields_ = [("density", C.c_double), ("inertia_damping", C.c_double), ("quasistatic", C.c_int32), ("elasticity_only", C.c_int32), ("scale", C.c_double),
                ("section_radius", C.c_double), ("youngs_modulus", C.c_double), ("strain_damping", C.c_double), ("strain_limit", C.c_double),
                ("strain_limit_stiffness", C.c_double)]


class ContactGlobalParams(C.Structure):
    _fields_ = [("default_contact_thickness", C.c_double), ("min_contact_stiffness", C.c_double), ("max_contact_stiffness", C.c_double),
                ("friction_stick_slide_threshold", C.c_double), ("collisions_enabled", C.c_int32), ("friction_enabled", C.c_int32),
                ("triangle_point_enabled", C.c_int32), ("edge_edge_enabled", C.c_int32), ("intersection_test_enabled", C.c_int32)]


class SimInfo(C.Structure):
    _fields_ = [("current_time", C.c_double), ("dt", C.c_double), ("current_time_step", C.c_int32), ("last_newton_result", C.c_int32), ("n_points", C.c_int64),
                ("ndofs", C.c_int64), ("total_newton_iterations", C.c_int64), ("total_cg_iterations", C.c_int64), ("total_linear_solves", C.c_int64),
                ("failed_steps", C.c_int64), ("total_newton_time", C.c_double), ("total_linear_solve_time", C.c_double), ("total_eval_pgh_time", C.c_double),
                ("total_eval_p_time", C.c_double), ("total_project_time", C.c_double), ("total_assembly_time", C.c_double), ("total_callback_time", C.c_double),
                ("total_step_time", C.c_double), ("total_evaluations", C.c_int64), ("last_stats", capi.NewtonStats)]


_bound = False


def _lib():
    global _bound
    L = capi.lib()
    if not _bound:
        p = C.c_void_p
        L.mistark_sim_default_settings.argtypes = [C.POINTER(SimSettings)]
        L.mistark_sim_default_settings.restype = None
        L.mistark_volume_params_soft_rubber.argtypes = [C.POINTER(VolumeParams)]
        L.mistark_volume_params_soft_rubber.restype = None
        L.mistark_surface_params_cotton_fabric.argtypes = [C.POINTER(SurfaceParams)]
        L.mistark_surface_params_cotton_fabric.restype = None
        L.mistark_sim_rb_fix_set_transformation.argtypes = [p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mistark_sim_create.argtypes = [C.POINTER(SimSettings), C.POINTER(p)]
        L.mistark_sim_destroy.argtypes = [p]
        L.mistark_sim_destroy.restype = None
        L.mistark_sim_last_error.argtypes = [p]
        L.mistark_sim_last_error.restype = C.c_char_p
        L.mistark_sim_add_volume_grid.argtypes = [p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(VolumeParams)]
        L.mistark_sim_add_volume.argtypes = [p, C.c_char_p, p, C.c_int64, p, C.c_int64, C.POINTER(VolumeParams)]
        L.mistark_sim_add_surface_grid.argtypes = [p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(SurfaceParams)]
        L.mistark_sim_add_surface.argtypes = [p, C.c_char_p, p, C.c_int64, p, C.c_int64, C.POINTER(SurfaceParams)]
        L.mistark_sim_prescribe_inside_aabb.argtypes = [p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]
        L.mistark_sim_prescribe_outside_aabb.argtypes = [p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]
        L.mistark_sim_prescribe_points.argtypes = [p, C.c_int, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_line_params_elastic_rubberband.argtypes = [C.POINTER(LineParams)]
        L.mistark_line_params_elastic_rubberband.restype = None
        L.mistark_sim_add_line.argtypes = [p, C.c_char_p, p, C.c_int64, p, C.c_int64, C.POINTER(LineParams)]
        L.mistark_sim_add_line_as_segments.argtypes = [p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.POINTER(LineParams)]
        L.mistark_sim_attach_point_point.argtypes = [p, C.c_int, C.c_int, p, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_sim_attach_point_edge.argtypes = [p, C.c_int, C.c_int, p, p, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_sim_attach_point_triangle.argtypes = [p, C.c_int, C.c_int, p, p, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_sim_attach_edge_edge.argtypes = [p, C.c_int, C.c_int, p, p, p, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_sim_attach_rigid_body.argtypes = [p, C.c_int, C.c_int, p, p, C.c_int64, C.c_double, C.c_double]
        L.mistark_sim_attach_by_distance.argtypes = [p, C.c_int, C.c_int, p, C.c_int64, p, C.c_int64, C.c_double, C.c_double, C.c_double, p]
        L.mistark_sim_attach_rigid_body_by_distance.argtypes = [p, C.c_int, C.c_int, p, C.c_int64, p, C.c_int64, p, C.c_int64, C.c_double, C.c_double, C.c_double]
        L.mistark_sim_attachment_stiffness.argtypes = [p, C.c_int, C.POINTER(C.c_double)]
        L.mistark_sim_run_one_step.argtypes = [p]
        L.mistark_sim_set_newton_settings.argtypes = [p, C.POINTER(capi.NewtonSettings)]
        L.mistark_sim_add_max_allowed_step.argtypes = [p, capi.DBLCB, C.c_void_p]
        L.mistark_sim_prepare.argtypes = [p]
        L.mistark_sim_begin_time_step.argtypes = [p]
        L.mistark_sim_before_energy_evaluation.argtypes = [p]
        L.mistark_sim_engine.argtypes = [p]
        L.mistark_sim_engine.restype = p
        L.mistark_sim_get_info.argtypes = [p, C.POINTER(SimInfo)]
        L.mistark_sim_get_points.argtypes = [p, C.c_int, p]
        L.mistark_sim_set_points.argtypes = [p, C.c_int, p]
        D = C.POINTER(C.c_double)
        L.mistark_sim_point_set_add_displacement.argtypes = [p, C.c_int, D]
        L.mistark_sim_point_set_add_rotation.argtypes = [p, C.c_int, C.c_double, D, D]
        L.mistark_sim_add_rigid_box.argtypes = [p, C.c_char_p, C.c_double, D]
        L.mistark_sim_rb_set_translation.argtypes = [p, C.c_int, D]
        L.mistark_sim_rb_add_translation.argtypes = [p, C.c_int, D]
        L.mistark_sim_rb_add_rotation.argtypes = [p, C.c_int, C.c_double, D, D]
        L.mistark_sim_rb_set_velocity.argtypes = [p, C.c_int, D, D]
        L.mistark_sim_rb_set_default_constraint_params.argtypes = [p, C.c_double, C.c_double, C.c_double]
        L.mistark_sim_rb_add_constraint.argtypes = [p, C.c_char_p, C.c_int, C.c_int, D, C.c_int]
        L.mistark_sim_rb_get_state.argtypes = [p, C.c_int, p, p, p, p]
        L.mistark_sim_rb_add.argtypes = [p, C.c_double, D]
        L.mistark_inertia_tensor_box.argtypes = [C.c_double, D, D]
        L.mistark_inertia_tensor_box.restype = None
        L.mistark_sim_rb_add_force_at_centroid.argtypes = [p, C.c_int, D]
        L.mistark_sim_rb_add_torque.argtypes = [p, C.c_int, D]
        L.mistark_sim_rb_constraint_count.argtypes = [p, C.c_char_p]
        L.mistark_sim_rb_constraint_measure.argtypes = [p, C.c_char_p, C.c_int, C.c_int, D, D]
        L.mistark_sim_run.argtypes = [p, C.c_double]
        L.mistark_contact_default_global_params.argtypes = [C.POINTER(ContactGlobalParams)]
        L.mistark_contact_default_global_params.restype = None
        L.mistark_sim_set_contact_global_params.argtypes = [p, C.POINTER(ContactGlobalParams)]
        L.mistark_sim_contact_group.argtypes = [p, C.c_int, C.c_int]
        L.mistark_sim_set_friction.argtypes = [p, C.c_int, C.c_int, C.c_double]
        L.mistark_sim_disable_collision.argtypes = [p, C.c_int, C.c_int]
        L.mistark_sim_set_dist_rccl.argtypes = [p, C.c_int, C.c_int, p]
        L.mistark_sim_set_dist_local.argtypes = [p, p, C.c_int, C.c_int]
        L.mistark_sim_set_dist_ipc.argtypes = [p, p, C.c_int, C.c_int]
        L.mistark_sim_get_contact_info.argtypes = [p, D, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _bound = True
    return L


def default_settings() -> SimSettings:
    s = SimSettings()
    _lib().mistark_sim_default_settings(C.byref(s))
    return s


def soft_rubber() -> VolumeParams:
    p = VolumeParams()
    _lib().mistark_volume_params_soft_rubber(C.byref(p))
    return p


def cotton_fabric() -> SurfaceParams:
    p = SurfaceParams()
    _lib().mistark_surface_params_cotton_fabric(C.byref(p))
    return p


def elastic_rubberband() -> LineParams:
    p = LineParams()
    _lib().mistark_line_params_elastic_rubberband(C.byref(p))
    return p


def inertia_tensor_box(mass, size) -> np.ndarray:
    out = (C.c_double * 9)()
    _lib().mistark_inertia_tensor_box(float(mass), _d3(size), out)
    return np.array(out[:]).reshape(3, 3)


def generate_triangle_grid(center, dim, subdivisions):
    """stark::generate_triangle_grid: (vertices [n, 3], triangles [m, 3])."""
    L = _lib()
    L.mistark_generate_triangle_grid.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64)]
    nv, nt = C.c_int64(), C.c_int64()
    assert L.mistark_generate_triangle_grid(_d3(center), _d3(dim), _i3(subdivisions), None, C.byref(nv), None, C.byref(nt)) == 0
    V, T = np.zeros((nv.value, 3)), np.zeros((nt.value, 3), dtype=np.int32)
    assert L.mistark_generate_triangle_grid(_d3(center), _d3(dim), _i3(subdivisions), V.ctypes.data, C.byref(nv), T.ctypes.data, C.byref(nt)) == 0
    return V, T


def find_edges_from_triangles(triangles, n_vertices):
    """stark::find_edges_from_simplices for triangles: unique edges [k, 2] in the reference's order."""
    L = _lib()
    L.mistark_find_edges_from_triangles.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    T = np.ascontiguousarray(triangles, dtype=np.int32)
    ne = C.c_int64()
    assert L.mistark_find_edges_from_triangles(T.ctypes.data, len(T), n_vertices, None, C.byref(ne)) == 0
    E = np.zeros((ne.value, 2), dtype=np.int32)
    assert L.mistark_find_edges_from_triangles(T.ctypes.data, len(T), n_vertices, E.ctypes.data, C.byref(ne)) == 0
    return E


def contact_global_params() -> ContactGlobalParams:
    p = ContactGlobalParams()
    _lib().mistark_contact_default_global_params(C.byref(p))
    return p


def _d3(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def _i3(v):
    return (C.c_int32 * len(v))(*[int(x) for x in v])


class SimError(RuntimeError):
    pass


class Simulation:
    """Mirror of stark::Simulation for the hot-path subset (deformables + presets)."""

    def __init__(self, settings: SimSettings | None = None):
        self.L = _lib()
        h = C.c_void_p()
        s = settings if settings is not None else default_settings()
        if self.L.mistark_sim_create(C.byref(s), C.byref(h)) != 0:
            raise SimError("mistark_sim_create failed")
        self.h = h

    def close(self):
        if self.h:
            self.L.mistark_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise SimError(self.L.mistark_sim_last_error(self.h).decode())
        return rc

    def add_volume_grid(self, label, center, dim, subdivisions, params: VolumeParams) -> int:
        return self._ck(self.L.mistark_sim_add_volume_grid(self.h, label.encode(), _d3(center), _d3(dim), _i3(subdivisions), C.byref(params)))

    def add_surface_grid(self, label, dim, subdivisions, params: SurfaceParams) -> int:
        return self._ck(self.L.mistark_sim_add_surface_grid(self.h, label.encode(), _d3(dim), _i3(subdivisions), C.byref(params)))

    def add_volume(self, label, vertices, tets, params: VolumeParams) -> int:
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        t = np.ascontiguousarray(tets, dtype=np.int32)
        return self._ck(self.L.mistark_sim_add_volume(self.h, label.encode(), v.ctypes.data, len(v), t.ctypes.data, len(t), C.byref(params)))

    def add_surface(self, label, vertices, triangles, params: SurfaceParams) -> int:
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        t = np.ascontiguousarray(triangles, dtype=np.int32)
        return self._ck(self.L.mistark_sim_add_surface(self.h, label.encode(), v.ctypes.data, len(v), t.ctypes.data, len(t), C.byref(params)))

    def add_line(self, label, vertices, segments, params: LineParams) -> int:
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        t = np.ascontiguousarray(segments, dtype=np.int32)
        return self._ck(self.L.mistark_sim_add_line(self.h, label.encode(), v.ctypes.data, len(v), t.ctypes.data, len(t), C.byref(params)))

    def add_line_as_segments(self, label, begin, end, n_segments, params: LineParams) -> int:
        return self._ck(self.L.mistark_sim_add_line_as_segments(self.h, label.encode(), _d3(begin), _d3(end), int(n_segments), C.byref(params)))

    def prescribe_points(self, point_set, points, stiffness, tolerance=0.0) -> int:
        pts = np.ascontiguousarray(points, dtype=np.int32)
        return self._ck(self.L.mistark_sim_prescribe_points(self.h, point_set, pts.ctypes.data, len(pts), stiffness, tolerance))

    # EnergyAttachments::add overloads; indices are local to their point set, tolerance <= 0 = none; return the handler index
    def attach_point_point(self, set_0, set_1, points_0, points_1, stiffness, tolerance=0.0) -> int:
        a, b = np.ascontiguousarray(points_0, dtype=np.int32), np.ascontiguousarray(points_1, dtype=np.int32)
        return self._ck(self.L.mistark_sim_attach_point_point(self.h, set_0, set_1, a.ctypes.data, b.ctypes.data, len(a), stiffness, tolerance))

    def attach_point_edge(self, set_0, set_1, points, edges, bary, stiffness, tolerance=0.0) -> int:
        a, e = np.ascontiguousarray(points, dtype=np.int32), np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        w = np.ascontiguousarray(bary, dtype=np.float64).reshape(-1, 2)
        return self._ck(self.L.mistark_sim_attach_point_edge(self.h, set_0, set_1, a.ctypes.data, e.ctypes.data, w.ctypes.data, len(a), stiffness, tolerance))

    def attach_point_triangle(self, set_0, set_1, points, triangles, bary, stiffness, tolerance=0.0) -> int:
        a, t = np.ascontiguousarray(points, dtype=np.int32), np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        w = np.ascontiguousarray(bary, dtype=np.float64).reshape(-1, 3)
        return self._ck(self.L.mistark_sim_attach_point_triangle(self.h, set_0, set_1, a.ctypes.data, t.ctypes.data, w.ctypes.data, len(a), stiffness, tolerance))

    def attach_edge_edge(self, set_0, set_1, edges_0, edges_1, bary_0, bary_1, stiffness, tolerance=0.0) -> int:
        e0, e1 = np.ascontiguousarray(edges_0, dtype=np.int32).reshape(-1, 2), np.ascontiguousarray(edges_1, dtype=np.int32).reshape(-1, 2)
        w0, w1 = np.ascontiguousarray(bary_0, dtype=np.float64).reshape(-1, 2), np.ascontiguousarray(bary_1, dtype=np.float64).reshape(-1, 2)
        return self._ck(self.L.mistark_sim_attach_edge_edge(self.h, set_0, set_1, e0.ctypes.data, e1.ctypes.data, w0.ctypes.data, w1.ctypes.data, len(e0), stiffness, tolerance))

    def attach_rigid_body(self, rb, point_set, points, stiffness, tolerance=0.0, rb_points_loc=None) -> int:
        a = np.ascontiguousarray(points, dtype=np.int32)
        loc = None if rb_points_loc is None else np.ascontiguousarray(rb_points_loc, dtype=np.float64).reshape(-1, 3)
        return self._ck(self.L.mistark_sim_attach_rigid_body(self.h, rb, point_set, None if loc is None else loc.ctypes.data, a.ctypes.data, len(a), stiffness, tolerance))

    def attach_by_distance(self, set_0, set_1, points, triangles, distance, stiffness, tolerance=0.0):
        """EnergyAttachments::add_by_distance: returns the (point-point, point-edge, point-triangle) handlers."""
        a = np.ascontiguousarray(points, dtype=np.int32)
        t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        out = (C.c_int32 * 3)()
        self._ck(self.L.mistark_sim_attach_by_distance(self.h, set_0, set_1, a.ctypes.data, len(a), t.ctypes.data, len(t), distance, stiffness, tolerance, out))
        return list(out)

    def attach_rigid_body_by_distance(self, rb, point_set, loc_vertices, triangles, points, distance, stiffness, tolerance=0.0) -> int:
        v = np.ascontiguousarray(loc_vertices, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        a = np.ascontiguousarray(points, dtype=np.int32)
        return self._ck(self.L.mistark_sim_attach_rigid_body_by_distance(self.h, rb, point_set, v.ctypes.data, len(v), t.ctypes.data, len(t), a.ctypes.data, len(a), distance, stiffness,
                                                                         tolerance))

    def attachment_stiffness(self, handler) -> float:
        k = C.c_double()
        self._ck(self.L.mistark_sim_attachment_stiffness(self.h, handler, C.byref(k)))
        return k.value

    def point_set_add_displacement(self, ps, d):
        self._ck(self.L.mistark_sim_point_set_add_displacement(self.h, ps, _d3(d)))

    def point_set_add_rotation(self, ps, angle_deg, axis, pivot=(0.0, 0.0, 0.0)):
        self._ck(self.L.mistark_sim_point_set_add_rotation(self.h, ps, angle_deg, _d3(axis), _d3(pivot)))

    # ---- rigid bodies ---------------------------------------------------------------------------------------------------
    def add_rigid_box(self, label, mass, size) -> int:
        size = (size, size, size) if np.isscalar(size) else size
        return self._ck(self.L.mistark_sim_add_rigid_box(self.h, label.encode(), mass, _d3(size)))

    def rb_add(self, mass, inertia_local) -> int:
        """RigidBodies::add(mass, inertia) without a collision mesh."""
        return self._ck(self.L.mistark_sim_rb_add(self.h, float(mass), _d3(np.asarray(inertia_local, dtype=float).reshape(9))))

    def rb_add_force_at_centroid(self, rb, f):
        self._ck(self.L.mistark_sim_rb_add_force_at_centroid(self.h, rb, _d3(f)))

    def rb_add_torque(self, rb, t):
        self._ck(self.L.mistark_sim_rb_add_torque(self.h, rb, _d3(t)))

    def rb_add_fix(self, rb):
        """RigidBodies::add_constraint_fix; returns the handler (anchor point, z lock, x lock) for rb_fix_set_transformation."""
        g, d = self.rb_constraint_count("global_point"), self.rb_constraint_count("global_direction")
        self.rb_add_constraint("fix", rb)
        return (g, d, d + 1)

    def rb_fix_set_transformation(self, fix, translation, angle_deg=0.0, axis=(0.0, 0.0, 1.0)):
        """RBCFixHandler::set_transformation(translation, angle_deg, axis) (rigidbody_constraints_ui.h:376-379)."""
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        th = np.deg2rad(angle_deg)
        K = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
        R = np.ascontiguousarray(np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K))
        t = np.ascontiguousarray(translation, dtype=np.float64)
        self._ck(self.L.mistark_sim_rb_fix_set_transformation(self.h, int(fix[0]), int(fix[1]), int(fix[2]), t.ctypes.data_as(C.POINTER(C.c_double)),
                                                              R.ctypes.data_as(C.POINTER(C.c_double))))

    def rb_constraint_count(self, base_type) -> int:
        return self._ck(self.L.mistark_sim_rb_constraint_count(self.h, base_type.encode()))

    def rb_constraint_measure(self, base_type, idx, which=0):
        """(violation, force or torque, tolerance) of a base constraint, as the reference's constraint handlers report them."""
        out = (C.c_double * 2)()
        tol = C.c_double()
        self._ck(self.L.mistark_sim_rb_constraint_measure(self.h, base_type.encode(), idx, which, out, C.byref(tol)))
        return out[0], out[1], tol.value

    def run(self, duration) -> bool:
        return self._ck(self.L.mistark_sim_run(self.h, float(duration))) == 1

    def rb_set_translation(self, rb, t):
        self._ck(self.L.mistark_sim_rb_set_translation(self.h, rb, _d3(t)))

    def rb_add_translation(self, rb, t):
        self._ck(self.L.mistark_sim_rb_add_translation(self.h, rb, _d3(t)))

    def rb_add_rotation(self, rb, angle_deg, axis, pivot=(0.0, 0.0, 0.0)):
        self._ck(self.L.mistark_sim_rb_add_rotation(self.h, rb, angle_deg, _d3(axis), _d3(pivot)))

    def rb_set_default_constraint_params(self, stiffness=0.0, tolerance_in_m=0.0, tolerance_in_deg=0.0):
        self._ck(self.L.mistark_sim_rb_set_default_constraint_params(self.h, stiffness, tolerance_in_m, tolerance_in_deg))

    def rb_add_constraint(self, kind, a, b=-1, *params):
        flat = []
        for x in params:
            flat.extend([float(x)] if np.isscalar(x) else [float(v) for v in x])
        return self._ck(self.L.mistark_sim_rb_add_constraint(self.h, kind.encode(), a, b, _d3(flat) if flat else None, len(flat)))

    def rb_state(self, rb):
        t, q, v, w = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3)
        self._ck(self.L.mistark_sim_rb_get_state(self.h, rb, t.ctypes.data, q.ctypes.data, v.ctypes.data, w.ctypes.data))
        return t, q, v, w

    # ---- multi-GPU sharding ------------------------------------------------------------------------------------------------
    def set_dist_rccl(self, rank, world, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        self._ck(self.L.mistark_sim_set_dist_rccl(self.h, rank, world, buf))

    def set_dist_local(self, group, rank, world):
        self._ck(self.L.mistark_sim_set_dist_local(self.h, group, rank, world))

    def set_dist_ipc(self, comm, rank, world):
        """comm: a connected capi.IpcComm (IPC windows, one process per rank; include/mistark.h)."""
        self._comm = comm  # (must outlive the engine)
        self._ck(self.L.mistark_sim_set_dist_ipc(self.h, comm.h, rank, world))

    # ---- frictional contact ----------------------------------------------------------------------------------------------
    def set_contact_global_params(self, p: ContactGlobalParams):
        self._ck(self.L.mistark_sim_set_contact_global_params(self.h, C.byref(p)))

    def contact_group(self, kind, idx) -> int:
        return self._ck(self.L.mistark_sim_contact_group(self.h, 0 if kind == "d" else 1, idx))

    def set_friction(self, group_a, group_b, mu):
        self._ck(self.L.mistark_sim_set_friction(self.h, group_a, group_b, mu))

    def disable_collision(self, group_a, group_b):
        self._ck(self.L.mistark_sim_disable_collision(self.h, group_a, group_b))

    def contact_info(self):
        k, n, nf, nd = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.L.mistark_sim_get_contact_info(self.h, C.byref(k), C.byref(n), C.byref(nf), C.byref(nd)))
        return dict(contact_stiffness=k.value, n_contacts=n.value, n_friction_contacts=nf.value, n_detections=nd.value)

    def prescribe_inside_aabb(self, point_set, center, dim, stiffness, tolerance=0.0) -> int:
        return self._ck(self.L.mistark_sim_prescribe_inside_aabb(self.h, point_set, _d3(center), _d3(dim), stiffness, tolerance))

    def prescribe_outside_aabb(self, point_set, center, dim, stiffness, tolerance=0.0) -> int:
        return self._ck(self.L.mistark_sim_prescribe_outside_aabb(self.h, point_set, _d3(center), _d3(dim), stiffness, tolerance))

    def set_newton_settings(self, s: capi.NewtonSettings):
        self._ck(self.L.mistark_sim_set_newton_settings(self.h, C.byref(s)))

    def add_max_allowed_step(self, fn):
        """fn() -> fraction of the Newton step the line search may take (symx::SolverCallbacks::add_max_allowed_step)."""
        from . import capi

        cb = capi.DBLCB(lambda _u: float(fn()))
        self._keep_cb = getattr(self, "_keep_cb", []) + [cb]
        self._ck(self.L.mistark_sim_add_max_allowed_step(self.h, cb, None))

    def run_one_step(self) -> bool:
        return self._ck(self.L.mistark_sim_run_one_step(self.h)) == 1

    def begin_time_step(self):
        self._ck(self.L.mistark_sim_begin_time_step(self.h))

    def before_energy_evaluation(self):
        self._ck(self.L.mistark_sim_before_energy_evaluation(self.h))

    def prepare(self):
        self._ck(self.L.mistark_sim_prepare(self.h))

    def info(self) -> SimInfo:
        i = SimInfo()
        self._ck(self.L.mistark_sim_get_info(self.h, C.byref(i)))
        return i

    def points(self, which="x0") -> np.ndarray:
        sel = {"X": 0, "x0": 1, "v0": 2, "v1": 3}[which]
        n = self.info().n_points
        out = np.zeros((n, 3))
        self._ck(self.L.mistark_sim_get_points(self.h, sel, out.ctypes.data))
        return out

    def set_points(self, which, arr):
        sel = {"x0": 1, "v0": 2}[which]
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._ck(self.L.mistark_sim_set_points(self.h, sel, arr.ctypes.data))

    def engine_handle(self):
        return C.c_void_p(self.L.mistark_sim_engine(self.h))

    def newton_iteration_log(self):
        """Per-iteration records (capi.NewtonIteration) of the last Newton solve of the last time step attempt."""
        from . import capi

        n = C.c_int32()
        if self.L.mistark_newton_iteration_log(self.engine_handle(), None, 0, C.byref(n)) < 0:
            raise SimError("newton_iteration_log failed")
        rec = (capi.NewtonIteration * max(n.value, 1))()
        if self.L.mistark_newton_iteration_log(self.engine_handle(), rec, n.value, C.byref(n)) < 0:
            raise SimError("newton_iteration_log failed")
        return list(rec)[:n.value]

    def spmv_timing(self, reset=0):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        rc = self.L.mistark_spmv_timing(self.engine_handle(), reset, C.byref(ms), C.byref(n), C.byref(b))
        if rc < 0:
            raise SimError("spmv_timing failed")
        return ms.value, n.value, b.value
