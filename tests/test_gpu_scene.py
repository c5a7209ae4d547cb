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
