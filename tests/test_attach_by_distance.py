"""EnergyAttachments::add_by_distance (stark/src/models/interactions/EnergyAttachments.cpp:229-297, 334-360) in the host mirror: the
attachment tables it registers (which points, which vertex / edge / face of the mesh, the barycentric weights, the rigid body's local
attachment points) against the tables the reference registered for the same scene (oracle/ref_harness.cpp: scene_attachdist) — on a
registration-only context, no GPU — and the trajectory of that scene on the GPU."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
from oracle import evaluator as ev  # noqa: E402


def box_mesh(s):
    """A cube of side s as a triangle mesh (any triangulation of the surface gives the same nearest points)."""
    h = 0.5 * s
    V = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    T = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], dtype=np.int32)
    return V, T


def build(S, sc, cloth_triangles, device):
    st = S.default_settings()
    st.device = device
    st.init_frictional_contact = 0
    sim = S.Simulation(st)
    n, d = sc["n"], sc["size"]
    hd = 0.5 * d
    cloth = sim.add_surface_grid("cloth", (d, d), (n, n), S.cotton_fabric())
    cV = sim.points("X").copy()
    corners = [i for i in range(len(cV)) if abs(abs(cV[i][0]) - hd) < 1e-9 and abs(cV[i][1] + hd) < 1e-9]
    sim.prescribe_points(cloth, corners, 1e6)
    patch = sim.add_surface_grid("patch", (0.45 * d, 0.45 * d), (3, 3), S.cotton_fabric())
    n_patch = len(sim.points("X")) - len(cV)
    sim.point_set_add_rotation(patch, sc["turn"], (0.0, 0.0, 1.0))
    sim.point_set_add_displacement(patch, (0.41 * d, 0.37 * d, sc["gap"]))
    h3 = sim.attach_by_distance(patch, cloth, list(range(n_patch)), cloth_triangles, sc["dist"], sc["k"], sc["tol"])
    bs = sc["box"]
    box = sim.add_rigid_box("box", sc["box_mass"], (bs, bs, bs))
    sim.rb_add_rotation(box, sc["box_turn"], (0.0, 0.0, 1.0))
    sim.rb_add_translation(box, (-0.11 * d, -0.07 * d, -0.5 * bs - sc["gap"]))
    V, T = box_mesh(bs)
    hb = sim.attach_rigid_body_by_distance(box, cloth, V, T, list(range(len(cV))), sc["box_dist"], sc["k"], sc["tol"])
    return sim, box, h3, hb


def _tables(L, h, names):
    """{potential name: (connectivity, [array of every binding as the caller holds it])} read back from a registration."""
    L.mistark_potential_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.mistark_potential_binding_data.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    n = L.mistark_describe(h, None, 0)
    buf = C.create_string_buffer(n)
    L.mistark_describe(h, buf, n)
    d = json.loads(buf.value.decode())
    out = {}
    for pi, p in enumerate(d["potentials"]):
        if p["name"] not in names:
            continue
        ne, stride = C.c_int64(), C.c_int32()
        assert L.mistark_potential_table(h, pi, None, C.byref(ne), C.byref(stride)) == 0
        conn = np.zeros((ne.value, stride.value), dtype=np.int32)
        assert L.mistark_potential_table(h, pi, conn.ctypes.data, None, None) == 0
        arrays = []
        for b in range(len(p["bindings"])):
            ptr, ni, st = C.c_void_p(), C.c_int64(), C.c_int32()
            assert L.mistark_potential_binding_data(h, pi, b, C.byref(ptr), C.byref(ni), C.byref(st)) == 0
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(ni.value * st.value,)).reshape(ni.value, st.value).copy() if ptr.value and ni.value else np.zeros((0, st.value))
            arrays.append(a)
        out[p["name"]] = (conn, arrays, p["bindings"])
    return out


def _rows(conn, arrays, bindings, cols_skip=(0,)):
    """One attachment per row as a comparable tuple: the connectivity without its running index, and the per-attachment data (bindings
    through connectivity column 0: barycentric weights, local points) — independent of row order."""
    per_row = [k for k, b in enumerate(bindings) if b[2] == 0 and not b[0].startswith("dof:")]
    out = []
    for r in range(conn.shape[0]):
        key = tuple(int(v) for c, v in enumerate(conn[r]) if c not in cols_skip)
        data = tuple(np.round(arrays[k][conn[r, 0]], 12).tolist().__repr__() for k in per_row)
        out.append((key, data))
    return sorted(out)


NAMES = ["EnergyAttachments_d_d_p_p", "EnergyAttachments_d_d_p_e", "EnergyAttachments_d_d_p_t", "EnergyAttachments_rb_d"]


def test_add_by_distance_registers_the_reference_tables():
    from stark_amd import capi
    from stark_amd import sim as S

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "attachdist.npz"))
    sc = json.loads(bytes(np.load(os.path.join(GOLDEN, "traj_attachdist.npz"))["traj_json"]).decode())["scene"]
    names = [q["name"] for q in man["potentials"]]
    tri = z["p%d_conn" % names.index("EnergyTriangleStrain")]
    cloth_tri = tri[tri[:, 2:5].max(axis=1) < (sc["n"] + 1) ** 2][:, 2:5]     # the cloth is the first point set: local = global indices
    sim, box, h3, hb = build(S, sc, cloth_tri, -1)
    sim.prepare()
    mine = _tables(capi.lib(), sim.engine_handle(), NAMES)
    for name in NAMES:
        pi = names.index(name)
        pot = prob.potentials[pi]
        ref_arrays = [prob.arrays[b.array] for b in pot.bindings]
        conn, arrays, bindings = mine[name]
        assert conn.shape == pot.conn.shape, name
        assert [(b.stride, b.conn) for b in pot.bindings] == [(b[1], b[2]) for b in bindings], name   # the same mws.make_* calls in the same order
        mine_rows = _rows(conn, arrays, bindings)
        ref_rows = _rows(pot.conn, ref_arrays, bindings)
        if name == "EnergyAttachments_d_d_p_e":
            # an edge shared by two triangles may be reported from either: (a, b) with weights (u, v) = (b, a) with (v, u)
            def canon(rows):
                out = []
                for key, data in rows:
                    g, p, a, b = key
                    w = [json.loads(x) for x in data]
                    if a > b:
                        a, b = b, a
                        w = [list(reversed(x)) if len(x) == 2 else x for x in w]
                    out.append(((g, p, a, b), tuple(repr(x) for x in w)))
                return sorted(out)
            mine_rows, ref_rows = canon(mine_rows), canon(ref_rows)
        assert mine_rows == ref_rows, name
    sim.close()


@pytest.mark.gpu
def test_add_by_distance_scene_trajectory():
    from stark_amd import sim as S

    z = np.load(os.path.join(GOLDEN, "traj_attachdist.npz"))
    traj = json.loads(bytes(z["traj_json"]).decode())
    man = json.loads(bytes(z["manifest_json"]).decode())
    sc = traj["scene"]
    names = [q["name"] for q in man["potentials"]]
    tri = z["p%d_conn" % names.index("EnergyTriangleStrain")]
    cloth_tri = tri[tri[:, 2:5].max(axis=1) < (sc["n"] + 1) ** 2][:, 2:5]
    def run():
        sim, box, h3, hb = build(S, sc, cloth_tri, 0)
        its = []
        for _ in traj["steps"]:
            assert sim.run_one_step()
            assert sim.info().last_newton_result == 0
            its.append(sim.info().last_stats.newton_iterations)
        x = sim.points("x0").copy()
        sim.close()
        return its, x

    its, x = run()
    # The first step pulls the patch onto the cloth from rest through k = 1e4 springs: linear solves of 300-400 CG iterations, where the float
    # rounding of the matrix decides the last Newton iterations. While the projection added its float deltas to the matrix with atomics (arrival
    # order) the engine took 13 or 14 iterations there from run to run (18 once, with the gradient rows summed by atomics as well) against the
    # reference's 15, and this test had to allow +-4. Since the touched blocks are gathered again in sorted order (project(), round 3) the
    # engine takes the reference's counts in every step, every time, with identical bits.
    # Round 5: whole-part assemblies sum the long contribution lists (the diagonal blocks) in another fixed order (k_assemble_gather_split); the
    # engine now takes [14, 7, 5, 6] — every time. The reference's own answer on this scene depends on ITS summation order, i.e. on its thread
    # count: `oracle/_ref/ref_harness traj attachdist steps=4 threads=T` gives 15 / 14 / 14 / 13 / 15 / 13 / 13 / 15 iterations in the first step
    # for T = 1, 2, 3, 4, 6, 8, 12, 16 (and [.., 8, 5, 6] behind it; the fixture holds one of those runs). The test asks for the reference's own
    # spread in the step that has one, one iteration of slack in the step that starts from its end state, the fixture's counts elsewhere, the
    # fixture's end state, and identical bits from run to run.
    ref = traj["newton_iterations"]
    REF_FIRST_STEP_SPREAD = (13, 15)
    assert REF_FIRST_STEP_SPREAD[0] <= its[0] <= REF_FIRST_STEP_SPREAD[1] and ref[0] == 15, (its, ref)
    assert abs(its[1] - ref[1]) <= 1 and its[2:] == ref[2:], (its, ref)
    # (end states of the reference's own runs at 4 / 8 / 16 threads against the fixture's: 1.5e-5 / 1.6e-5 / 2.5e-5 m, 4.1e-5 among themselves — a
    # different number of Newton iterations ends anywhere inside the solver tolerance; the engine: 2.4e-5. Bound: 2e-4 of the extent = 4.3e-5 m)
    assert np.abs(x - z["x_end"]).max() <= 2e-4 * np.abs(z["x_end"]).max()
    its2, x2 = run()
    assert its2 == its and (x2 == x).all()



@pytest.mark.gpu
def test_add_by_distance_scene_trajectory_with_the_lane_per_block_gather_takes_the_fixtures_counts():
    """the default whole-part gather (k_assemble_gather_split + mirrored pairs) sums the long lists in another fixed order and the
    test above had to widen to the reference's own thread-count spread. The lane-per-block gather (options no_split_gather, no_sym_gather: every
    block summed in list order, as through round 4) still takes the fixture's Newton counts exactly and its end state to 1e-4 — pinned here, so the
    wider bound above cannot hide a regression of anything else on this scene."""
    from stark_amd import sim as S

    z = np.load(os.path.join(GOLDEN, "traj_attachdist.npz"))
    traj = json.loads(bytes(z["traj_json"]).decode())
    man = json.loads(bytes(z["manifest_json"]).decode())
    sc = traj["scene"]
    names = [q["name"] for q in man["potentials"]]
    tri = z["p%d_conn" % names.index("EnergyTriangleStrain")]
    cloth_tri = tri[tri[:, 2:5].max(axis=1) < (sc["n"] + 1) ** 2][:, 2:5]
    sim, box, h3, hb = build(S, sc, cloth_tri, 0)
    sim.prepare()                      # (the engine exists from here on)
    L = sim.L
    L.mistark_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    for opt in (b"no_split_gather", b"no_sym_gather"):
        assert L.mistark_set_option(sim.engine_handle(), opt, 1) == 0
    its = []
    for _ in traj["steps"]:
        assert sim.run_one_step()
        assert sim.info().last_newton_result == 0
        its.append(sim.info().last_stats.newton_iterations)
    x = sim.points("x0").copy()
    sim.close()
    assert its == traj["newton_iterations"], (its, traj["newton_iterations"])
    assert np.abs(x - z["x_end"]).max() <= 1e-4 * np.abs(z["x_end"]).max()
