"""GPU, BASELINE.json's full sizes. (1) configs[3], [2] and [4] against stage outputs of the UNMODIFIED reference at full size
(tests/golden/slim_cfg*.npz, made by `ref_harness slimdump`: a closed-form DoF state at the initial configuration, so no evaluator inputs
need to travel): energy, the whole gradient, every contact and friction table as a row set, the pattern size, SpMV probes of the assembled
and of the PSD-projected matrix, and the PCG outcome (iteration count +-1, convergence / indefiniteness verdict) on both.
(2) size-independent properties of the hot path on configs[3]: symmetry of the assembled operator, the PCG answer checked by an
independent residual, idempotence of contact detection."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


class _Eng:
    """Engine view over the context a Simulation owns."""

    def __new__(cls, sim):
        import stark_amd
        from stark_amd import capi

        class E(stark_amd.Engine):
            def __init__(self, h):
                self.L = capi.lib()
                self.h = h
                self._keep = []
                self.potential_ids = {}

            def close(self):
                pass

            def __del__(self):
                pass

        return E(sim.engine_handle())


def _build_cfg(S, sc):
    """The scene of a slim fixture (oracle/ref_harness.cpp: scene_blockbox / scene_clothbox / scene_mixed) through the host mirror."""
    from stark_amd import sim as Sm

    st = Sm.default_settings()
    st.init_frictional_contact = 1
    st.mirror_state_to_host = 0
    sim = Sm.Simulation(st)
    gp = Sm.contact_global_params()
    gp.default_contact_thickness = sc["thickness"]
    gp.min_contact_stiffness = sc["kmin"]
    sim.set_contact_global_params(gp)
    if sc["kind"] == "blockbox":
        rb = sim.add_rigid_box("box", 1.0, (sc["bx"], sc["bx"], sc["bz"]))
        sim.rb_add_constraint("fix", rb)
        L = sc["L"]
        ps = sim.add_volume_grid("block", (sc.get("ox", 0.0), sc.get("oy", 0.0), 0.5 * sc["bz"] + sc["gap"] + 0.5 * L), (L, L, L), (sc["nx"], sc["ny"], sc["nz"]), Sm.soft_rubber())
        sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), sc["mu"])
    elif sc["kind"] == "clothbox":
        ps = sim.add_surface_grid("cloth", (sc["size"], sc["size"]), (sc["n"], sc["n"]), Sm.cotton_fabric())
        rb = sim.add_rigid_box("box", 1.0, sc["box"])
        sim.rb_add_translation(rb, (sc.get("ox", 0.0), sc.get("oy", 0.0), -0.5 * sc["box"] - sc["gap"]))
        sim.rb_add_constraint("fix", rb)
        sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), sc["mu"])
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_scene import _build_mixed

        _build_mixed(Sm, sim, sc)
    return sim


@pytest.mark.parametrize("name", ["slim_cfg3_blockbox_44x44x43", "slim_cfg2_clothbox_256", "slim_cfg4_mixed_26x26x25"])
def test_full_size_stages_match_reference(name):
    import json

    from stark_amd import capi
    from stark_amd import sim as S

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from contact_util import sorted_rows

    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    man = json.loads(bytes(z["slim_json"]).decode())
    sim = _build_cfg(S, man["scene"])
    sim.prepare()
    if man["xamp"]:                             # (the harness's closed-form displacement out of the exactly-parallel start geometry)
        x0 = sim.points("x0")
        k = np.arange(x0.size).reshape(x0.shape)
        sim.set_points("x0", x0 + man["xamp"] * np.sin(0.9 * k + 0.3))
    sim.begin_time_step()                       # friction tables at the start-of-step geometry, rigid-body caches, v1 = 0
    eng = _Eng(sim)
    n = eng.ndofs
    assert n == man["ndofs"]
    i = np.arange(n)
    eng.set_dofs(man["amp"] * np.sin(1.3 * i + 0.7))
    sim.before_energy_evaluation()              # contact tables at these DoFs
    # ---- the contact sets: every table of the reference, as a row set (bit-exact indices)
    ref_tables = {k[2:]: z[k] for k in z.files if k.startswith("t_")}

    def canonical(name, t):
        # The orientation of a deformable collision EDGE is the reference's find_surface hash order: rows are compared
        # with the two vertices of every edge sorted. Layouts (contact_and_friction_data.h:69-358): header, then per side its edge (2 columns)
        # followed by the closest point when that side's feature is a point. Friction rows start with their own running index: dropped.
        t = np.array(t, dtype=np.int64)
        parts = name.split("_")
        if parts[0] == "contact" and parts[3] == "ee":
            xa, xb = parts[4][0] == "p", parts[4][1] == "p"
            h = t.shape[1] - (4 + int(xa) + int(xb))
            for c0 in (h, h + 2 + int(xa)):
                t[:, c0:c0 + 2] = np.sort(t[:, c0:c0 + 2], axis=1)
        if parts[0] == "friction":
            t = t[:, 1:]
            kind = parts[3]
            pairs = {"ee": (t.shape[1] - 4, t.shape[1] - 2), "pe": (t.shape[1] - 2,), "ep": (t.shape[1] - 3,)}.get(kind, ())
            for c0 in pairs:
                t[:, c0:c0 + 2] = np.sort(t[:, c0:c0 + 2], axis=1)
        return sorted_rows(t)

    n_rows = 0
    for p in man["potentials"]:
        if p["name"].startswith(("contact_", "friction_")):
            ours = eng.contact_table(p["name"])
            assert len(ours) == p["n_elem"], (p["name"], len(ours), p["n_elem"])
            if p["n_elem"]:
                assert (canonical(p["name"], ours) == canonical(p["name"], ref_tables[p["name"]])).all(), p["name"]
            n_rows += p["n_elem"]
    assert n_rows > 1000
    # ---- energy and gradient (sums of up to 1.2 M element terms in another order: 1e-10)
    E, g = eng.eval(capi.EVAL_P_G_H)
    assert abs(E - man["E"]) <= 1e-9 * max(1.0, abs(man["E"]))
    gref = z["grad"]
    n_soft = 3 * sim.info().n_points        # (DoFs: soft.v1 | rigid.v1 | rigid.w1)
    gmax = np.abs(gref).max()
    # Friction rows with a deformable EDGE: the tangent basis follows the edge's direction, i.e. the orientation the reference's find_surface
    # hash order gave it, and the C0 friction energy adds the constant (1.13e-9, -1.07e-9) in THAT basis (EnergyFrictionalContact.cpp:
    # 1260-1278): the force on those nodes (and the reaction on the rigid body) moves by ~mu fn / eps_u * 1e-9 with the orientation
    # (DESIGN.md section 6). Everything else is pinned at 1e-11 of the largest entry.
    loose = np.zeros(n, dtype=bool)
    loose[n_soft:] = True
    for name, t in ref_tables.items():
        parts = name.split("_")
        if parts[0] == "friction" and parts[3] in ("ee", "pe", "ep"):
            fam = parts[1] + "_" + parts[2]
            cols = t[:, 2:] if fam == "rb_d" else t[:, 1:]   # (index, [rigid body,] vertices): rigid-local vertex ids name no soft DoF, but
            nodes = np.unique(cols[cols < n_soft // 3])      # marking a few more soft nodes than necessary only loosens the test there
            for d in range(3):
                loose[3 * nodes + d] = True
    assert loose[:n_soft].sum() < 0.35 * n_soft
    diff = np.abs(g - gref)
    assert diff[~loose].max() <= 1e-11 * gmax
    assert diff[loose].max() <= 1e-7 * gmax
    assert abs(np.abs(g).max() - man["residual"]) <= 1e-10 * man["residual"]
    # ---- assembled matrix: pattern size, SpMV probe (every 4th entry stored), PCG at the Newton forcing tolerance
    eng.assemble()
    x = np.sin(0.37 * i)

    def probe(tag):
        y = eng.spmv(x)[::4]
        ref = z[tag].astype(np.float64)
        soft = np.arange(0, n, 4) < n_soft
        assert np.abs(y - ref)[soft].max() <= 2e-5 * np.abs(ref[soft]).max(), tag
        # rigid-body rows: up to 10^5 float blocks (and, once projected, float deltas) accumulated in another order than the reference's
        assert np.abs(y - ref)[~soft].max() <= 1e-3 * np.abs(ref).max(), tag

    probe("spmv_y")
    row_ptr, cols, _ = eng.get_bsr(with_vals=False)
    assert 9 * len(cols) == man["nnz_scalar"]

    def solve(ref):
        du, info = eng.pcg(ref["abs_tol"], ref["rel_tol"], 10000)
        assert bool(info.converged) == bool(ref["converged"]) and bool(info.found_indefiniteness) == bool(ref["indefinite"])
        assert abs(info.n_iterations - ref["iterations"]) <= 1
        if info.n_iterations == ref["iterations"]:
            assert abs(du @ (-g) - ref["x_dot_rhs"]) <= 2e-3 * abs(ref["x_dot_rhs"]) and abs(np.linalg.norm(du) - ref["x_norm"]) <= 2e-3 * ref["x_norm"]

    solve(man["pcg"])
    # ---- every element Hessian projected to PSD (deltas patched into the assembled matrix): probe and solve again
    n_proj, _ = eng.project(1e-10, False, None)
    assert n_proj == man["n_hessians"]
    probe("spmv_y_proj")
    solve(man["pcg_projected"])
    sim.close()


@pytest.mark.parametrize("name", ["slim_cfg2_clothbox_256", "slim_cfg4_mixed_26x26x25"])
def test_full_size_contact_closed_forms_equal_the_generic_evaluator(name):
    """contact_closed.hpp (hand-derived gradient / Hessian of the 35 contact and friction potentials, one lane per contact) against the
    generic hyper-dual evaluation of the same expressions (option force_generic) on the contact sets of the full-size fixtures — 68 k
    point-triangle contacts under configs[2]'s cloth, every table of configs[4]: every element Hessian, element energy and the gradient
    at 1e-11 of the largest entry. (Against the reference itself: test_full_size_stages_match_reference above and the stage fixtures of
    tests/test_gpu_parity.py, which hold all 35 potentials.)"""
    import json

    from stark_amd import capi
    from stark_amd import sim as S

    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    man = json.loads(bytes(z["slim_json"]).decode())
    sim = _build_cfg(S, man["scene"])
    sim.prepare()
    if man["xamp"]:
        x0 = sim.points("x0")
        k = np.arange(x0.size).reshape(x0.shape)
        sim.set_points("x0", x0 + man["xamp"] * np.sin(0.9 * k + 0.3))
    sim.begin_time_step()
    eng = _Eng(sim)
    eng.set_dofs(man["amp"] * np.sin(1.3 * np.arange(eng.ndofs) + 0.7))
    sim.before_energy_evaluation()
    got = {}
    eng.set_option("contact_closed_min_lanes", 0)  # (every table in closed form, whatever its size)
    for generic in (0, 1):
        eng.set_option("force_generic", generic)
        E, g = eng.eval(capi.EVAL_P_G_H)
        per = {}
        for p in man["potentials"]:
            if p["name"].startswith(("contact_", "friction_")) and p["n_elem"]:
                pid = eng.potential_id(p["name"])
                H, rows = eng.element_hessians(pid, p["n_elem"])
                per[p["name"]] = (H, rows, eng.element_energies(pid, p["n_elem"]))
        got[generic] = (E, g, per)
    eng.set_option("force_generic", 0)
    (E0, g0, per0), (E1, g1, per1) = got[0], got[1]
    assert len(per0) >= 4 and sum(len(v[2]) for v in per0.values()) > 1000
    assert abs(E0 - E1) <= 1e-12 * max(1.0, abs(E1))
    assert np.abs(g0 - g1).max() <= 1e-11 * np.abs(g1).max()
    for nm, (H, rows, Ee) in per0.items():
        Hg, rows_g, Eg = per1[nm]
        assert (rows == rows_g).all()
        # point-line distances are evaluated as |u|^2 - (u.e)^2 / |e|^2 (distances.cpp:61-68): against a 2 m edge of the floor and a
        # distance of 1e-3 the subtraction cancels 6-7 digits, so ANY two orderings of the same derivative expression (the reference's
        # generated code, the hyper-dual pass, the closed form) differ by ~1e-16 (|u| / d)^2 = 4e-10 there
        tol = 2e-9 if nm.split("_")[-2] in ("pe", "ep") else 1e-11
        dev = np.abs(H - Hg).max() / np.abs(Hg).max()
        print("%-32s %6d elements, closed form vs generic: %.1e" % (nm, len(Ee), dev))
        assert dev <= tol, nm
        assert np.abs(Ee - Eg).max() <= 1e-12 * max(np.abs(Eg).max(), 1e-300), nm
    sim.close()


def test_full_size_properties():
    import bench
    from stark_amd import capi
    from stark_amd import sim as S

    sim = bench.build_scene(S, 44, 44, 43, 0, "contact")
    for _ in range(2):                       # two time steps: contact active, friction tables filled
        assert sim.run_one_step()
    info = sim.info()
    assert info.ndofs == 517050
    ci = sim.contact_info()
    assert ci["n_contacts"] > 1000 and ci["n_friction_contacts"] > 1000
    eng = _Eng(sim)
    n = eng.ndofs
    dt = info.dt
    # contact detection is a pure function of the state: twice the same tables, and the all-pairs search finds the same set
    # (a repeated request at an unchanged state is normally answered from the detector's cache: switched off here)
    assert eng.contact_update(dt) == ci["n_contacts"]            # cached answer
    eng.set_option("no_contact_cache", 1)
    n1 = eng.contact_update(dt)
    t1 = {name: eng.contact_table(name) for name in ("contact_rb_d_pt_tp_cubic", "contact_rb_d_ee_ee_cubic", "contact_rb_d_pt_pt_cubic")}
    eng.contact_set_broad_phase(True)
    n2 = eng.contact_update(dt)
    eng.contact_set_broad_phase(False)
    assert n1 == n2 == ci["n_contacts"]
    for name, t in t1.items():
        assert (eng.contact_table(name) == t).all(), name       # rows are sorted by pair id: identical arrays
    assert eng.contact_count_intersections(dt) == 0
    eng.set_option("no_contact_cache", 0)
    # assembled operator: symmetric, positive on the PCG search space once projected
    E0, g = eng.eval(capi.EVAL_P_G_H)
    eng.project(1e-10)
    eng.assemble()
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = eng.spmv(x), eng.spmv(y)
    assert abs(y @ Ax - x @ Ay) <= 1e-5 * max(abs(y @ Ax), 1.0)   # float storage: symmetric to rounding
    assert x @ Ax > 0 and y @ Ay > 0
    # linearity
    Axy = eng.spmv(2.0 * x - 3.0 * y)
    assert np.abs(Axy - (2.0 * Ax - 3.0 * Ay)).max() <= 1e-9 * np.abs(Axy).max()
    # PCG: residual of the returned solution, computed with an independent SpMV
    du, pinfo = eng.pcg(1e-8, 1e-6, 10000)
    assert pinfo.converged
    r = -g - eng.spmv(du)
    assert np.linalg.norm(r) <= 2e-6 * np.linalg.norm(g)
    assert du @ g < 0                                            # descent direction
    # one more step through the whole Newton loop: accepted, energy went down along every accepted iterate
    assert sim.run_one_step()
    st = sim.info().last_stats
    assert st.newton_iterations >= 1
    sim.close()


def _bsr_matvec(row_ptr, cols, vals, x):
    """y = A x from the exported block CSR (float blocks, double accumulate), row after row."""
    import scipy.sparse as sp

    A = sp.bsr_matrix((vals.astype(np.float64), cols, row_ptr), shape=(3 * (len(row_ptr) - 1),) * 2)
    return A @ x


@pytest.mark.parametrize("chunk_tiles", [0, 1, 4, 8])
@pytest.mark.parametrize("n_cloth", [24, 40])
def test_spmv_rows_longer_than_a_chunk(n_cloth, chunk_tiles):
    """The static matrix part is stored in row-aligned chunks of 2, 4 or 8 tiles by matrix size (8 tiles = 512 blocks; forced here, 0 =
    automatic); a rigid body attached to every point of a cloth owns two block rows with one block per point (625 / 1681 here): they fit
    a chunk of their own or take the over-long-row path depending on the chunk length. The device product must equal the product with
    the exported matrix either way."""
    from stark_amd import capi
    from stark_amd import sim as S

    st = S.default_settings()
    st.init_frictional_contact = 0
    sim = S.Simulation(st)
    cloth = sim.add_surface_grid("cloth", (1.0, 1.0), (n_cloth, n_cloth), S.cotton_fabric())
    box = sim.add_rigid_box("box", 1.0, (0.2, 0.2, 0.2))
    sim.rb_add_translation(box, (0.0, 0.0, 0.3))
    npts = (n_cloth + 1) ** 2
    sim.attach_rigid_body(box, cloth, list(range(npts)), 1e3)
    sim.prescribe_inside_aabb(cloth, (0.5, 0.5, 0.0), (0.001, 0.001, 0.001), 1e6)
    assert sim.run_one_step()
    eng = _Eng(sim)
    eng.set_option("spmv_chunk_tiles", chunk_tiles)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    row_ptr, cols, vals = eng.get_bsr()
    lens = np.diff(row_ptr)
    assert lens.max() >= npts     # the body's rows
    rng = np.random.default_rng(11)
    x = rng.standard_normal(eng.ndofs)
    y = eng.spmv(x)
    y_ref = _bsr_matvec(row_ptr, cols, vals, x)
    assert np.abs(y - y_ref).max() <= 1e-12 * np.abs(y_ref).max()
    # and the solver built on it still converges to the right answer
    du, pinfo = eng.pcg(1e-10, 1e-8, 20000, rhs=x)
    assert pinfo.converged
    assert np.linalg.norm(x - _bsr_matvec(row_ptr, cols, vals, du)) <= 1e-6 * np.linalg.norm(x)
    sim.close()


def _step_and_check(sim, n_steps, expect_ndofs):
    """A few whole time steps at full size: every step accepted, contact active, no edge-triangle intersection in the accepted
    state, finite state; returns Newton iterations and wall time (reported by -s / tools, not asserted)."""
    import time

    t0 = time.perf_counter()
    for _ in range(n_steps):
        assert sim.run_one_step()
    wall = time.perf_counter() - t0
    info = sim.info()
    assert info.ndofs == expect_ndofs
    ci = sim.contact_info()
    assert ci["n_contacts"] > 0
    eng = _Eng(sim)
    # (the accepted state itself: x0 with dt = 0; x0 + dt v1 would extrapolate the committed velocities one more step)
    assert eng.contact_count_intersections(0.0) == 0
    x = sim.points("x0")
    assert np.isfinite(x).all()
    return info.total_newton_iterations, wall, ci


def test_full_size_cloth_on_floor():
    """BASELINE configs[2]: 256 x 256 Cotton_Fabric cloth (discrete-shell bending off: the preset's flat bending) dropped on a fixed
    floor box with IPC contact + friction: 198 153 DoF."""
    from stark_amd import sim as S

    st = S.default_settings()
    st.mirror_state_to_host = 0
    st.init_frictional_contact = 1
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = 1e-3
    sim.set_contact_global_params(gp)
    floor = sim.add_rigid_box("floor", 1.0, (2.0, 2.0, 0.1))
    sim.rb_add_constraint("fix", floor)
    cloth = sim.add_surface_grid("cloth", (1.0, 1.0), (256, 256), S.cotton_fabric())
    sim.point_set_add_displacement(cloth, (0.0, 0.0, 0.05 + 0.0015))
    sim.set_friction(sim.contact_group("rb", floor), sim.contact_group("d", cloth), 0.5)
    its, wall, ci = _step_and_check(sim, 8, 3 * 257 * 257 + 6)
    x = sim.points("x0")
    assert x[:, 2].min() > 0.05                      # nothing went through the floor's top face
    assert ci["n_friction_contacts"] > 0
    # the floor's diagonal blocks collect > 10^5 contributions each: summed by 64 wavefronts + an ordered fold (k_assemble_vlong_*);
    # the scatter assembly with float atomics is an independent path to the same matrix
    from stark_amd import capi
    eng = _Eng(sim)
    eng.contact_update(sim.info().dt)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    row_ptr, cols, vals = eng.get_bsr()
    eng.set_option("atomic_assembly", 1)
    eng.eval(capi.EVAL_P_G_H)
    eng.assemble()
    row_ptr2, cols2, vals2 = eng.get_bsr()
    eng.set_option("atomic_assembly", 0)
    assert (row_ptr == row_ptr2).all() and (cols == cols2).all()
    n_rb_blocks = int(row_ptr[-1] - row_ptr[-3])     # the two block rows of the floor (v, w): dense against every touched node
    assert n_rb_blocks > 2 * 60000
    scale = np.abs(vals).max()
    assert np.abs(vals - vals2).max() <= 2e-4 * scale  # float atomics in arbitrary order over 1.4e5 terms vs double sums rounded once
    rb = slice(int(row_ptr[-3]), int(row_ptr[-1]))
    assert np.abs(vals[rb]).max() > 0
    print("configs[2]: %d Newton iterations in %.3f s (8 steps) = %.1f Newton-steps/s" % (its, wall, its / wall))
    sim.close()


def test_cfg2_cloth_dropped_from_5_cm_takes_the_references_attempts():
    """configs[2] as BASELINE / SURVEY 8d describe it: the 256 x 256 Cotton_Fabric cloth DROPPED from 5 cm on the fixed floor, dt = 1/30 (VERDICT
    r04 #5; fixture steplog_cfg2_clothbox_drop_256 = the reference's per-attempt log with 8 and with 4 threads, which agree with each other).
    The scene's first time step is a hard one for the reference as well — six Newton iterations, then FOUR failed attempts of 30-31 linear solves
    each (every one of them 10 000 CG iterations: cg_max_iterations) that halve dt down to 2 ms — and the engine takes exactly those attempts:
    Newton iterations and linear solves `==` for the first 24 attempts; the CG iterations of the counted Newton iterations from the
    ninth attempt on within a third of the reference's. What happens to the cloth afterwards is NOT pinned: impact comes at 1 m/s with steps of 5-30 mm against a 2 mm barrier range and
    the reference's contact model has no CCD, so whether the cloth is caught or passes through the floor is decided by where a step happens to
    end — the reference's own runs split (caught in its `traj` run of this scene, not caught in its `time` runs); see
    test_cfg2_cloth_dropped_with_1_ms_steps_lands for the well-posed variant."""
    import json

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from steplog_cfg2 import build

    z = np.load(os.path.join(ROOT, "tests", "golden", "steplog_cfg2_clothbox_drop_256.npz"))
    ref8 = json.loads(bytes(z["time_t8_json"]).decode())["per_step"]
    ref4 = json.loads(bytes(z["time_t4_json"]).decode())["per_step"]
    assert [r[:2] for r in ref8] == [r[:2] for r in ref4]
    sim = build(0.05)
    n_attempts = 24
    prev = (0, 0)
    mine = []
    for s in range(n_attempts):
        assert sim.run_one_step()
        i = sim.info()
        log = sim.newton_iteration_log()
        mine.append([i.total_newton_iterations - prev[0], i.total_linear_solves - prev[1], sum(r.cg_iterations_last for r in log if r.logged)])
        prev = (i.total_newton_iterations, i.total_linear_solves)
    sim.close()
    assert [m[:2] for m in mine] == [r[:2] for r in ref8[:n_attempts]], (mine, ref8[:n_attempts])
    assert [m[:2] for m in mine[:5]] == [[6, 7], [0, 31], [0, 30], [0, 30], [0, 30]]
    # CG iterations of the counted Newton iterations from the ninth attempt on (the three steps behind the failed attempts are tolerance-limited
    # solves of thousands of iterations whose leftovers decay differently from run to run — 37 / 277 / 1389 iterations in the seventh attempt of
    # the reference's two runs and of the engine, 760 / 779 / 45 in the eighth): within a third of the reference's
    cg_mine = sum(m[2] for m in mine[8:])
    cg_ref = [sum(r[2] for r in ref[8:n_attempts]) for ref in (ref8, ref4)]
    assert 0.67 * min(cg_ref) <= cg_mine <= 1.33 * max(cg_ref), (cg_mine, cg_ref)


def test_cfg2_tilted_cloth_lands_like_the_references_runs_that_land():
    """configs[2] as a well-posed dynamic scene (fixture steplog_cfg2_tilted_256): the 256 x 256 Cotton_Fabric cloth tilted 3
    degrees, its lowest edge 2 mm (the contact distance) above the fixed floor, released at dt = 1/30. A tilted cloth straddles the floor's
    surface whenever it penetrates, so the reference's intersection check catches it — unlike the flat 5 cm drop above — and the cloth lands and
    rests. HOW the reference gets there is not reproducible, not even at a fixed thread count: five runs of the unmodified reference (the
    fixture's `time` runs at 8 and 4 threads and its `traj` run at 4; two more while the scene was set up) fall into two families. In one the
    first three steps take 32 / 56-61 / 45-59 Newton iterations (53-63 / 95-103 / 92-118 linear solves), five steps of one iteration follow and
    the cloth is at rest from the ninth step on; in the other the first step takes 37-38 iterations and the next attempts FAIL (0 iterations, 17
    linear solves each) and halve dt. The engine — bit-reproducible — lands: no failed attempt, the three landing steps inside the landing runs'
    own spread (+- 10 % on Newton iterations, + 20 % on linear solves), one iteration per step while the cloth settles, none from the eighth or
    ninth step on at the rest state's CG count per solve, and the reference's cloth after 10 steps (its `traj` run) to a millimetre."""
    import json

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from steplog_cfg2 import build

    z = np.load(os.path.join(ROOT, "tests", "golden", "steplog_cfg2_tilted_256.npz"))
    logs = [json.loads(bytes(z[k]).decode())["per_step"] for k in ("time_t8_json", "time_t4_json")]
    traj = json.loads(bytes(z["traj_json"]).decode())["steps"]
    logs.append([[st["newton"], st["linear_solves"], None] for st in traj])
    failing = [lg for lg in logs if any(r[0] == 0 and r[1] >= 16 for r in lg[1:5])]        # a failed attempt right behind the first step
    landing = [lg for lg in logs if lg not in failing]
    assert len(landing) >= 2 and len(failing) >= 1, "the fixture holds both families of the reference's runs"
    for lg in landing:
        assert all(r[:2] == [1, 2] for r in lg[3:7]) and all(r[0] == 0 and r[1] == 1 for r in lg[9:10])
    sim = build(0.002, tilt=3.0)
    n_attempts = 12
    prev = (0, 0, 0)
    mine = []
    for s in range(n_attempts):
        assert sim.run_one_step()
        i = sim.info()
        assert abs(i.dt - 1.0 / 30.0) < 1e-12 and abs(i.current_time - (s + 1) / 30.0) < 1e-9, (s, i.dt, i.current_time)   # no failed attempt
        cur = (i.total_newton_iterations, i.total_linear_solves, i.total_cg_iterations)
        mine.append([c - p for c, p in zip(cur, prev)])
        prev = cur
        if s == 9:
            x, X = sim.points("x0")[::64], sim.points("X")[::64]
    sim.close()
    print("configs[2] tilted: engine", mine, "reference's landing runs", [lg[:n_attempts] for lg in landing])
    for k in range(3):   # the landing: each step inside the reference's own spread over its landing runs
        its = [lg[k][0] for lg in landing]
        sol = [lg[k][1] for lg in landing]
        assert 0.9 * min(its) - 2 <= mine[k][0] <= 1.1 * max(its) + 2, (k, mine[k], its)
        assert 0.8 * min(sol) - 2 <= mine[k][1] <= 1.2 * max(sol) + 2, (k, mine[k], sol)
    assert [m[:2] for m in mine[3:7]] == [[1, 2]] * 4, mine
    assert mine[7][:2] in ([1, 2], [1, 1], [0, 1]) and mine[8][:2] in ([1, 2], [1, 1], [0, 1]) and all(m[:2] == [0, 1] for m in mine[9:]), mine
    rest_ref = [lg[12][2] for lg in landing if len(lg) > 12 and lg[12][2]]   # (a step of the cloth at rest: one linear solve)
    assert rest_ref
    assert all(0.95 * min(rest_ref) <= m[2] <= 1.05 * max(rest_ref) for m in mine[9:]), (mine[9:], rest_ref)
    xr = z["x_end_every64"]
    dev = np.abs(x - xr).max(axis=0)
    print("configs[2] tilted: deviation from the reference's cloth after 10 steps (m, per axis):", dev, "travel:", np.abs(xr - X).max(axis=0))
    assert dev.max() <= 1e-3 and dev[2] <= 2e-4, dev


def test_full_size_mixed_scene():
    """BASELINE configs[4]: 202 800-tet Soft_Rubber block on a fixed floor + 128 x 128 cloth over it + a chain of 16 boxes joined by
    hinges (first link fixed) over the cloth; contact and friction between the layers (oracle/ref_harness.cpp scene_mixed)."""
    from stark_amd import sim as S
    from test_gpu_scene import _build_mixed

    st = S.default_settings()
    st.mirror_state_to_host = 0
    st.init_frictional_contact = 1
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = 1e-3
    gp.min_contact_stiffness = 1e8
    sim.set_contact_global_params(gp)
    sc = dict(nx=26, ny=26, nz=25, nc=128, nrb=16, L=1.0, gap=0.0015, bx=3.0, bz=0.1, link=0.05, cloth=1.2, mu=0.5)
    floor, links, block, cloth = _build_mixed(S, sim, sc)
    n_nodes = 27 * 27 * 26 + 26 * 26 * 25 + 129 * 129
    its, wall, ci = _step_and_check(sim, 3, 3 * n_nodes + 6 * 17)
    # the fixed first link stayed where it was, the hinges hold the chain together (1 mm default tolerance of the constraints)
    t0 = np.array(sim.rb_state(links[0])[0])
    assert np.abs(t0 - np.array([-0.5 * 15 * 0.075, 0.0, 0.05 + 0.0015 + 1.0 + 0.0015 + 0.0015 + 0.025])).max() < 2e-3
    for a, b in zip(links[:-1], links[1:]):
        d = np.linalg.norm(np.array(sim.rb_state(a)[0]) - np.array(sim.rb_state(b)[0]))
        assert abs(d - 0.075) < 5e-3
    print("configs[4]: %d Newton iterations in %.3f s (3 steps) = %.1f Newton-steps/s" % (its, wall, its / wall))
    sim.close()


def _step_log(sim, n_attempts, z=None, check_after=1):
    """(Newton iterations, linear solves, sum of the last solves' CG iterations) per time-step attempt + the per-iteration CG series; the
    state after attempt `check_after` against the fixture's sampled end state (relative to the step's displacement / velocity)."""
    per_step, series, prev, dev = [], [], (0, 0), None
    for s in range(n_attempts):
        assert sim.run_one_step()
        i = sim.info()
        cur = (i.total_newton_iterations, i.total_linear_solves)
        log = sim.newton_iteration_log()
        assert sum(r.linear_solves for r in log) == cur[1] - prev[1] and len(log) == cur[0] - prev[0] + 1
        per_step.append([cur[0] - prev[0], cur[1] - prev[1], sum(r.cg_iterations_last for r in log if r.logged)])
        series.append([r.cg_iterations_last for r in log if r.logged])
        prev = cur
        if z is not None and s == check_after:
            x, v, X = sim.points("x0")[::64], sim.points("v0")[::64], sim.points("X")[::64]
            xr, vr = z["x_end_every64"], z["v_end_every64"]
            dev = (np.abs(x - xr).max() / np.abs(xr - X).max(), np.abs(v - vr).max() / np.abs(vr).max(), i.current_time)
    return per_step, series, dev


def test_full_size_first_time_steps_equal_the_reference_log_off_the_degenerate_placement():
    """BASELINE configs[3] at full size (998 976 tets, IPC contact + friction on the rigid floor), the block moved 1.37 mm / -0.53 mm off
    the box's axes, over its first eight time-step attempts, against the log of the UNMODIFIED reference (fixture
    `steplog_cfg3_offset_44x44x43`: `ref_harness time` with 8 and 4 threads — identical — and `traj slim` of the first two attempts):
    Newton iterations, linear solves (progressive-projection retries included: 27 solves in the eighth attempt) and the CG iterations of
    every Newton iteration's last solve are EQUAL, attempt by attempt; the CG series of the first two attempts entry by entry; the state after
    the first accepted step to 1e-4 of the step's displacement and velocity."""
    import json

    import bench
    from stark_amd import sim as S

    z = np.load(os.path.join(ROOT, "tests", "golden", "steplog_cfg3_offset_44x44x43.npz"))
    ref = [json.loads(bytes(z["time_t%d_json" % t]).decode()) for t in (8, 4)]
    traj = json.loads(bytes(z["traj_json"]).decode())
    assert ref[0]["per_step"] == ref[1]["per_step"] and ref[0]["ndofs"] == 517050
    sc = ref[0]["scene"]
    sim = bench.build_scene(S, sc["nx"], sc["ny"], sc["nz"], 0, offset=(sc["ox"], sc["oy"]))
    per_step, series, dev = _step_log(sim, len(ref[0]["per_step"]), z)
    assert per_step == ref[0]["per_step"], (per_step, ref[0]["per_step"])
    n0 = len(series[0])
    assert series[0] + series[1] == traj["cg_iterations"] and [n0 - 1, len(series[1]) - 1] == traj["newton_iterations"]
    print("end state of the first accepted step: relative deviation x %.2e v %.2e" % dev[:2])
    assert abs(dev[2] - 1.0 / 30.0) < 1e-12 and dev[0] <= 1e-4 and dev[1] <= 1e-4
    sim.close()


def test_full_size_centred_placement_with_the_references_own_tie_decisions(tmp_path):
    """The centred placement bench.py times, with the reference's OWN classification of the pairs that sit on exact ties: oracle/_ref/shim_check
    is the unmodified reference (its scene classes, its EnergyFrictionalContact, its host collision detector tmcd with its tie decisions and
    edge orientations) on this engine through the SymX shim. With those decisions the engine takes the reference's Newton iterations and
    linear solves in ALL five attempts of the reference's log — [5, 6], [4, 5], [3, 4], [3, 21], [6, 17] — and its CG iterations in the
    first three (104, 67, 154; the fourth and fifth are where the reference's own runs differ from each other). The same
    binary with the detector on the GPU (shim_check_cd: this repo's decisions on the ties, equal pair sets otherwise) leaves the log at the
    first attempt's CG count: the ties are the only difference between the engine and the reference on this placement."""
    import json
    import subprocess

    exe = os.path.join(ROOT, "oracle", "_ref", "shim_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_check not built")
    z = np.load(os.path.join(ROOT, "tests", "golden", "steplog_cfg3_blockbox_44x44x43.npz"))
    ref = [json.loads(bytes(z["time_t%d_json" % t]).decode())["per_step"] for t in (8, 4)]
    assert ref[0] == ref[1] and len(ref[0]) == 5

    def run(binary):
        log = str(tmp_path / (os.path.basename(binary) + ".log"))
        env = dict(os.environ, SHIM_GRID="44,44,43", SHIM_THREADS="32", MISTARK_SHIM_SOLVELOG=log)
        r = subprocess.run([binary, "benchblock", "5"], capture_output=True, timeout=1200, env=env)
        assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-1500:])
        return [[int(v) for v in line.split()] for line in open(log).read().split("\n") if line.strip()]

    ours = run(exe)
    print("reference log           ", ref[0])
    print("engine, reference's ties", ours)
    assert [o[:2] for o in ours[:5]] == [r[:2] for r in ref[0]]
    assert [o[2] for o in ours[:3]] == [r[2] for r in ref[0][:3]]
    cd = exe + "_cd"
    if os.path.exists(cd):
        mine = run(cd)
        print("engine, its own ties    ", mine)
        assert mine[0][:2] == ref[0][0][:2] and mine[0][2] != ref[0][0][2]


def test_full_size_first_time_steps_on_the_centred_placement():
    """The placement bench.py measures (block centred on the box). Here the bottom-face nodes with x = -y lie EXACTLY above the diagonal
    edge of the box's top face, and 78 of the 264 edge-edge pairs have their closest point exactly at an edge endpoint: whether such a pair
    is an edge-edge or an edge-point contact is decided by the last bit of a product, i.e. by how a compiler contracted the multiply-adds
    (energy and gradient are the same either way — E and max |grad| of the first evaluation equal the reference's to all printed digits —
    the Hessian puts the curvature on another node). On this placement the reference does not reproduce itself either: the same build with
    the same thread count logged 21 linear solves in the fourth attempt in most runs and 17 in another (thread-order-dependent float sums
    next to exact ties); off the axes its logs are identical and the engine's equal them (test above). Replayed at the reference's OWN iterates the engine
    reproduces every later Newton step to 1e-5 (tools/steplog_cfg3.py). What is pinned here: the first attempt
    (6 Newton iterations, ends in "invalid converged state": the floor's constraint is hardened, the step redone) has the reference's
    counts and its first four solves the reference's CG iterations; the state after the first accepted step agrees to 1 % of the step
    (measured 0.3 %); the totals over five attempts stay within the spread of the reference's own two runs, widened.
    Round 4: the reference's answer here is not even a property of its sources — built with -ffp-contract=off the same sources log
    [5,6,101] [5,6,99] [3,16,173] [3,16,161] [9,24,557] instead of [5,6,104] [4,5,67] [3,4,154] [3,21,171] [6,17,357]
    (tests/test_oracle_golden.py::test_the_references_own_log_on_exact_ties_depends_on_its_compile_flags); the engine with its own tie decisions logs
    [5,6,97] [5,6,95] [3,12,159] [3,19,160] [3,20,202]: between the two builds in its first attempts, apart from both later — three answers to a
    question that has none; the pinned placement (test above) is where the answer is unique."""
    import json

    import bench
    from stark_amd import sim as S

    z = np.load(os.path.join(ROOT, "tests", "golden", "steplog_cfg3_blockbox_44x44x43.npz"))
    ref = [json.loads(bytes(z["time_t%d_json" % t]).decode()) for t in (8, 4)]
    traj = json.loads(bytes(z["traj_json"]).decode())
    assert [r["ndofs"] for r in ref] == [517050, 517050] and ref[0]["per_step"][:3] == ref[1]["per_step"][:3]
    sim = bench.build_scene(S, 44, 44, 43, 0)
    per_step, series, dev = _step_log(sim, 5, z)
    print("centred placement, engine:", per_step)
    assert per_step[0][:2] == ref[0]["per_step"][0][:2] == [5, 6]
    assert series[0][:4] == traj["cg_iterations"][:4] == [3, 19, 22, 4]
    assert abs(per_step[1][0] - ref[0]["per_step"][1][0]) <= 1 and abs(per_step[1][1] - ref[0]["per_step"][1][1]) <= 1
    assert abs(dev[2] - 1.0 / 30.0) < 1e-12 and dev[0] <= 1e-2 and dev[1] <= 1e-2
    newton, solves = sum(p[0] for p in per_step), sum(p[1] for p in per_step)
    ref_newton = [r["newton_iterations"] for r in ref]
    ref_solves = [r["linear_solves"] for r in ref]
    assert 0.75 * min(ref_newton) <= newton <= 1.25 * max(ref_newton), (per_step, ref_newton)
    assert 0.6 * min(ref_solves) <= solves <= 1.6 * max(ref_solves), (per_step, ref_solves)
    sim.close()


@pytest.mark.parametrize("grid", [(20, 5, 5), (52, 13, 13), (30, 30, 30)])
def test_multifrontal_cholesky_equals_the_band_cholesky_and_solves_the_system(grid, monkeypatch):
    """DirectLLT beyond the band (direct.hip: nested dissection, dense fronts, extend-add): the same solution as the block-tridiagonal
    path where both fit, and a residual at rounding level against the engine's own SpMV (float matrix, double vectors). 30 x 30 x 30
    hexahedra (89 373 unknowns) have a band of 8 GB; the multifrontal factor takes a fraction of it."""
    from stark_amd import capi
    from stark_amd import sim as S

    st = S.default_settings()
    st.mirror_state_to_host = 0
    sim = S.Simulation(st)
    nx, ny, nz = grid
    ps = sim.add_volume_grid("block", (0, 0, 0), (nx / 10.0, ny / 10.0, nz / 10.0), grid, S.soft_rubber())
    sim.prescribe_inside_aabb(ps, (-0.5 * nx / 10.0, 0, 0), (2e-3, 10.0, 10.0), 1e7)
    sim.prepare()
    sim.begin_time_step()
    eng = _Eng(sim)
    n = eng.ndofs
    eng.set_dofs(1e-3 * np.sin(1.3 * np.arange(n) + 0.7))
    eng.eval(capi.EVAL_P_G_H)
    eng.project(1e-10, False, None)
    eng.assemble()
    b = np.cos(0.11 * np.arange(n))
    sols = {}
    for mode in ((1, 2, -1) if n < 80000 else (1, 2)):   # 1: separators from the positions, 2: from breadth-first level sets, -1: band
        eng.set_option("llt_no_coords", int(mode == 2))
        eng.set_option("llt_multifrontal", 1 if mode == 2 else mode)
        x, ok = eng.direct_llt(b)
        assert ok
        r = eng.spmv(x) - b
        print(grid, {1: "multifrontal (positions)", 2: "multifrontal (level sets)", -1: "band"}[mode], "relative residual %.1e" % (np.linalg.norm(r) / np.linalg.norm(b)))
        assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)
        sols[mode] = x
    eng.set_option("llt_multifrontal", 0)
    eng.set_option("llt_no_coords", 0)
    assert np.abs(sols[1] - sols[2]).max() <= 1e-9 * np.abs(sols[2]).max()
    if -1 in sols:
        assert np.abs(sols[1] - sols[-1]).max() <= 1e-9 * np.abs(sols[-1]).max()
    # a matrix that is not positive definite is reported by every path (SimplicialLLT's info() != Success), not factored: the unprojected
    # Hessian of a badly distorted state (the PCG's own verdict on it: indefinite)
    if n < 80000:
        eng.set_dofs(8.0 * np.sin(1.3 * np.arange(n) + 0.7))
        eng.eval(capi.EVAL_P_G_H)
        eng.assemble()
        _, info = eng.pcg(1e-8, 1e-8, 2000, stop_on_indef=True)
        assert info.found_indefiniteness
        for mode in (1, 2, -1):
            eng.set_option("llt_no_coords", int(mode == 2))
            eng.set_option("llt_multifrontal", 1 if mode == 2 else mode)
            _, ok = eng.direct_llt(b)
            assert not ok, mode
        eng.set_option("llt_multifrontal", 0)
        eng.set_option("llt_no_coords", 0)
    sim.close()
