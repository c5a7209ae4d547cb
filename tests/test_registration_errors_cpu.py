"""CPU (registration-only context, no GPU): what the C ABI does with wrong registrations — the reference prints and exit(-1)s on misuse
(GlobalPotential.cpp:8-13); here every entry point returns < 0 and mistark_last_error says what was wrong, and nothing is half-registered."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Binding(C.Structure):
    _fields_ = [("array", C.c_int32), ("stride", C.c_int32), ("conn_col", C.c_int32)]


@pytest.fixture()
def ctx():
    from stark_amd import capi

    L = capi.lib()
    L.mistark_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert L.mistark_create_dry(C.byref(h)) == 0
    yield L, h
    L.mistark_destroy(h)


def describe(L, h):
    n = L.mistark_describe(h, None, 0)
    buf = C.create_string_buffer(n)
    L.mistark_describe(h, buf, n)
    return json.loads(buf.value.decode())


def test_bad_dof_sets_and_arrays(ctx):
    L, h = ctx
    v = np.zeros(12)
    L.mistark_add_dof_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    assert L.mistark_add_dof_set(h, b"soft.v1", v.ctypes.data, 12) >= 0
    assert L.mistark_add_dof_set(h, b"bad", v.ctypes.data, -3) < 0 and L.mistark_last_error(h)
    L.mistark_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    assert L.mistark_array(h, v.ctypes.data, 4, 0) < 0            # stride 0
    assert L.mistark_array(h, v.ctypes.data, -1, 3) < 0           # negative item count
    L.mistark_dof_array.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert L.mistark_dof_array(h, 7, 3) < 0                        # no such DoF set
    assert len(describe(L, h)["dof_sets"]) == 1                    # the rejected set did not register


def test_bad_potentials(ctx):
    L, h = ctx
    v1 = np.zeros(12)
    x0 = np.zeros((4, 3))
    L.mistark_add_dof_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    assert L.mistark_add_dof_set(h, b"soft.v1", v1.ctypes.data, 12) >= 0
    L.mistark_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    a_v = L.mistark_array(h, v1.ctypes.data, 4, 3)
    a_x = L.mistark_array(h, x0.ctypes.data, 4, 3)
    assert a_v >= 0 and a_x >= 0
    conn = np.zeros((1, 6), dtype=np.int32)
    L.mistark_potential.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    bs = (Binding * 2)(Binding(a_v, 3, 2), Binding(a_x, 3, 2))
    # a name without a kernel: an explicit error (user energies go through mistark_potential_custom with their op sequence)
    assert L.mistark_potential(h, b"EnergyOfMyOwn", conn.ctypes.data, 1, 6, bs, 2) < 0
    assert b"EnergyOfMyOwn" in L.mistark_last_error(h)
    # a known name with the wrong binding list (the tet energy binds far more than two arrays)
    assert L.mistark_potential(h, b"EnergyTetStrain", conn.ctypes.data, 1, 6, bs, 2) < 0
    # a binding that points at an array that does not exist / a connectivity column outside the table
    bad = (Binding * 1)(Binding(99, 3, 2))
    assert L.mistark_potential(h, b"EnergyLumpedInertia", conn.ctypes.data, 1, 3, bad, 1) < 0
    assert describe(L, h)["potentials"] == []                      # nothing was half-registered
    # evaluation entry points refuse a registration-only context
    L.mistark_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert L.mistark_eval(h, 0, None, None) < 0 and b"registration-only" in L.mistark_last_error(h)


def test_unknown_option_and_null_context(ctx):
    L, h = ctx
    L.mistark_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    assert L.mistark_set_option(h, b"no_such_switch", 1) < 0 and b"no_such_switch" in L.mistark_last_error(h)
    assert L.mistark_set_option(h, b"no_eval_overlap", 1) == 0
    assert L.mistark_set_option(None, b"no_eval_overlap", 1) < 0
    assert L.mistark_last_error(None) == b"null context"


def test_scene_facade_rejects_bad_handles_and_indices():
    """The host mirror of stark::Simulation on a registration-only context: handles that name nothing and indices outside their point set
    are errors where the reference exits (Handler::exit_if_not_valid, IntervalVector::_assert_local_idx), not out-of-range reads later."""
    from stark_amd import sim as S

    st = S.default_settings()
    st.device = -1
    st.init_frictional_contact = 0
    sim = S.Simulation(st)
    a = sim.add_surface_grid("a", (1.0, 1.0), (2, 2), S.cotton_fabric())
    b = sim.add_surface_grid("b", (1.0, 1.0), (2, 2), S.cotton_fabric())
    for call in (lambda: sim.prescribe_points(99, [0], 1e6),
                 lambda: sim.prescribe_points(a, [1000], 1e6),
                 lambda: sim.prescribe_points(a, [-1], 1e6),
                 lambda: sim.attach_point_point(a, b, [0], [500], 1e3),
                 lambda: sim.attach_point_point(a, b, [0, 1], [0], 1e3),
                 lambda: sim.attach_by_distance(a, b, [0], [[0, 1, 77]], 0.1, 1e3),
                 lambda: sim.rb_add_translation(42, (0.0, 0.0, 1.0)),
                 lambda: sim.set_friction(5, 6, 0.5)):
        with pytest.raises(S.SimError):
            call()
    sim.prescribe_points(a, [0, 8], 1e6)          # (the valid calls still work afterwards)
    sim.attach_point_point(a, b, [0], [8], 1e3)
    sim.prepare()
    sim.close()
