"""GPU: scenes built through the C++ host mirror (stark::Simulation-style API, include/mistark_sim.h) reproduce the
reference's trajectories (same meshes, same registration, same Newton / CG iteration counts, positions <= 1e-6 rel)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, json.loads(bytes(z["traj_json"]).decode()), json.loads(bytes(z["manifest_json"]).decode())


def test_tetbeam_scene_trajectory():
    from stark_amd import sim as S

    z, traj, man = _load("traj_tetbeam_eo_8x2x2")
    sc = traj["scene"]
    sim = S.Simulation()
    p = S.soft_rubber()
    p.elasticity_only = sc["eo"]
    ps = sim.add_volume_grid("beam", (0, 0, 0), (sc["lx"], sc["ly"], sc["lz"]), (sc["nx"], sc["ny"], sc["nz"]), p)
    sim.prescribe_inside_aabb(ps, (-0.5 * sc["lx"], 0, 0), (2e-3, 2 * sc["ly"], 2 * sc["lz"]), 1e7)
    # identical mesh (generate_tet_grid restatement): rest positions equal the reference's X array
    X = sim.points("X")
    Xref = z["a%d" % man["potentials"][[q["name"] for q in man["potentials"]].index("EnergyTetStrain_Elasticity_Only")]["bindings"][8]["array"]]
    assert X.shape == Xref.shape and np.abs(X - Xref).max() == 0.0
    its = []
    cg = 0
    for _ in traj["steps"]:
        assert sim.run_one_step()
        i = sim.info()
        assert i.last_newton_result == 0
        its.append(i.last_stats.newton_iterations)
        cg += i.last_stats.cg_iterations
    assert its == traj["newton_iterations"]
    assert abs(cg - sum(traj["cg_iterations"])) <= 2
    x = sim.points("x0")
    assert np.abs(x - z["x_end"]).max() <= 1e-6 * np.abs(z["x_end"]).max()
    v = sim.points("v0")
    assert np.abs(v - z["v_end"]).max() <= 1e-6 * max(np.abs(z["v_end"]).max(), 1.0)
    sim.close()


def test_cloth_scene_trajectory():
    from stark_amd import sim as S

    z, traj, man = _load("traj_cloth_flat_8")
    sc = traj["scene"]
    sim = S.Simulation()
    p = S.cotton_fabric()
    p.elasticity_only = sc["eo"]
    p.flat_rest_angle = sc["flat"]
    d = sc["size"]
    ps = sim.add_surface_grid("cloth", (d, d), (sc["n"], sc["n"]), p)
    sim.prescribe_inside_aabb(ps, (0.5 * d, 0.5 * d, 0), (0.001, 0.001, 0.001), 1e6)
    sim.prescribe_inside_aabb(ps, (-0.5 * d, 0.5 * d, 0), (0.001, 0.001, 0.001), 1e6)
    its = []
    for _ in traj["steps"]:
        assert sim.run_one_step()
        its.append(sim.info().last_stats.newton_iterations)
    assert its == traj["newton_iterations"]
    x = sim.points("x0")
    assert np.abs(x - z["x_end"]).max() <= 1e-6 * np.abs(z["x_end"]).max()
    sim.close()


def test_rigid_body_chain_trajectory():
    """The rbchain scene of oracle/ref_harness.cpp (9 boxes, all 13 rigid-body potentials) through the host layer's
    RigidBodies API: same Newton iteration counts and the same converged velocities as the reference, step by step."""
    from stark_amd import sim as S

    z, traj, man = _load("traj_rbchain")
    st = S.default_settings()
    st.init_frictional_contact = 0
    sim = S.Simulation(st)
    sim.rb_set_default_constraint_params(stiffness=1e4)
    axis = np.array([1.0, 0.5, -0.3])
    axis /= np.linalg.norm(axis)

    def box(x, y, zz):
        b = sim.add_rigid_box("box", 1.0 + 0.1 * x, (0.1, 0.12, 0.08))
        sim.rb_set_translation(b, (x, y, zz))
        sim.rb_add_rotation(b, 20.0 * x + 5.0, axis)
        return b

    def unit(v):
        v = np.asarray(v, dtype=float)
        return v / np.linalg.norm(v)

    b0 = box(0.0, 0.0, 0.0)
    sim.rb_add_constraint("fix", b0)
    b1 = box(0.15, 0.0, 0.0)
    sim.rb_add_constraint("point", b0, b1, (0.07, 0.02, 0.01))
    b2 = box(0.30, 0.02, 0.0)
    sim.rb_add_constraint("point_on_axis", b1, b2, (0.22, 0.0, 0.01), unit((1.0, 0.2, 0.0)))
    b3 = box(0.45, 0.0, 0.03)
    sim.rb_add_constraint("distance", b2, b3, (0.33, 0.0, 0.0), (0.42, 0.01, 0.02))
    b4 = box(0.60, 0.0, 0.0)
    sim.rb_add_constraint("distance_limits", b3, b4, (0.48, 0.0, 0.0), (0.57, 0.0, 0.0), 0.08995, 0.09005)
    sim.rb_add_constraint("direction", b3, b4, unit((0.0, 1.0, 0.2)))
    b5 = box(0.75, 0.0, 0.0)
    sim.rb_add_constraint("point", b4, b5, (0.67, 0.0, 0.0))
    sim.rb_add_constraint("angle_limit", b4, b5, (1.0, 0.0, 0.0), 0.5)
    b6 = box(0.90, 0.0, 0.0)
    sim.rb_add_constraint("spring", b5, b6, (0.78, 0.0, 0.0), (0.87, 0.01, 0.0), 200.0, 3.0)
    b7 = box(1.05, 0.0, 0.0)
    sim.rb_add_constraint("point_on_axis", b6, b7, (0.97, 0.0, 0.0), (1.0, 0.0, 0.0))
    sim.rb_add_constraint("linear_velocity", b6, b7, (1.0, 0.0, 0.0), 0.3, 5.0, 0.05)
    b8 = box(1.20, 0.0, 0.0)
    sim.rb_add_constraint("hinge", b7, b8, (1.12, 0.0, 0.0), (0.0, 1.0, 0.0))
    sim.rb_add_constraint("angular_velocity", b7, b8, (0.0, 1.0, 0.0), 1.0, 2.0, 0.1)

    iterates, iter_step = z["iterates"], np.array(traj["iter_step"])
    nb = 9
    its = []
    for step in range(len(traj["steps"])):
        assert sim.run_one_step()
        i = sim.info()
        # the reference's first call ends with "invalid converged state" (the fix anchor sags beyond its 1 mm tolerance, the
        # constraint stiffness is doubled and the step is redone): the recorded time after each call tells which calls advanced
        assert abs(i.current_time - traj["steps"][step]["time"]) < 1e-12, (step, i.last_newton_result)
        its.append(i.last_stats.newton_iterations)
        ref = iterates[np.nonzero(iter_step == step)[0][-1]]   # last evaluation point of the step = converged DoFs
        v_ref, w_ref = ref[:3 * nb].reshape(nb, 3), ref[3 * nb:].reshape(nb, 3)
        v = np.array([sim.rb_state(b)[2] for b in range(nb)])
        w = np.array([sim.rb_state(b)[3] for b in range(nb)])
        assert np.abs(v - v_ref).max() <= 2e-5 * max(np.abs(v_ref).max(), 1e-3), step
        assert np.abs(w - w_ref).max() <= 2e-5 * max(np.abs(w_ref).max(), 1e-3), step
    assert its == traj["newton_iterations"]
    sim.close()


def _contact_sim(S, sc):
    st = S.default_settings()
    st.init_frictional_contact = 1
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = sc["thickness"]
    gp.min_contact_stiffness = sc["kmin"]
    sim.set_contact_global_params(gp)
    return sim


def _run_and_compare(sim, z, traj, tol=1e-6, its_slack=0):
    its = []
    for step in range(len(traj["steps"])):
        assert sim.run_one_step()
        i = sim.info()
        assert abs(i.current_time - traj["steps"][step]["time"]) < 1e-12, (step, i.last_newton_result)
        its.append(i.last_stats.newton_iterations)
    if its_slack == 0:
        assert its == traj["newton_iterations"]
    else:
        assert all(abs(a - b) <= its_slack for a, b in zip(its, traj["newton_iterations"])), (its, traj["newton_iterations"])
    x = sim.points("x0")
    assert np.abs(x - z["x_end"]).max() <= tol * np.abs(z["x_end"]).max()
    v = sim.points("v0")
    assert np.abs(v - z["v_end"]).max() <= 10 * tol * max(np.abs(z["v_end"]).max(), 1.0)
    return its


def test_cloth_on_box_contact_trajectory():
    """cfg 1 (README hello world, no spin) at fixture size: IPC contact + friction between a cloth and a fixed rigid box, device
    detection inside the Newton loop; same Newton iteration counts and end state as the reference."""
    from stark_amd import sim as S

    z, traj, man = _load("traj_clothbox_8")
    sc = traj["scene"]
    sim = _contact_sim(S, sc)
    ps = sim.add_surface_grid("cloth", (sc["size"], sc["size"]), (sc["n"], sc["n"]), S.cotton_fabric())
    rb = sim.add_rigid_box("box", 1.0, sc["box"])
    sim.rb_add_translation(rb, (0.0, 0.0, -0.5 * sc["box"] - sc["gap"]))
    sim.rb_add_constraint("fix", rb)
    sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), sc["mu"])
    # inexact Newton (forcing-sequence CG on a float matrix) only pins the iterates to the solver tolerance; the free in-plane
    # modes of the cloth carry that difference from step to step: 1e-4 of the cloth size after 4 steps
    _run_and_compare(sim, z, traj, tol=1e-4)
    info = sim.contact_info()
    assert info["n_contacts"] > 0 and info["n_friction_contacts"] > 0 and info["n_detections"] > 0
    sim.close()


@pytest.mark.parametrize("name", ["traj_blockbox_3", "traj_blockbox_3_nofriction"])
def test_block_on_box_contact_trajectory(name):
    """cfg 4 at fixture size: Soft_Rubber tet block landing on a fixed rigid box (collision surface from find_surface), with
    friction (box registered first) and without (block first)."""
    from stark_amd import sim as S

    z, traj, man = _load(name)
    sc = traj["scene"]
    sim = _contact_sim(S, sc)
    L = sc["L"]

    def add_box():
        rb = sim.add_rigid_box("box", 1.0, (sc["bx"], sc["bx"], sc["bz"]))
        sim.rb_add_constraint("fix", rb)
        return rb

    def add_block():
        return sim.add_volume_grid("block", (0.0, 0.0, 0.5 * sc["bz"] + sc["gap"] + 0.5 * L), (L, L, L), (sc["nx"], sc["ny"], sc["nz"]), S.soft_rubber())

    if sc["boxfirst"]:
        rb = add_box()
        ps = add_block()
    else:
        ps = add_block()
        rb = add_box()
    if sc["mu"] > 0:
        sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), sc["mu"])
    if sc["mu"] > 0:
        _run_and_compare(sim, z, traj, tol=1e-4)
    else:
        # frictionless: the block's lateral rigid motion is a neutral mode (nothing restores it), so solver-tolerance differences
        # integrate freely (observed 2e-4 m lateral vs 2.5e-5 m vertical after 6 steps) and convergence tests sit on the threshold
        _run_and_compare(sim, z, traj, tol=1e-3, its_slack=2)
    assert sim.contact_info()["n_contacts"] > 0
    sim.close()
