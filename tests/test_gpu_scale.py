"""GPU: the other BASELINE.json configurations at their full sizes, as robustness / property runs (no oracle finishes at these sizes;
their small versions are the trajectory tests of tests/test_gpu_scene.py):
  configs[2]  256x256 Cotton_Fabric cloth dropped on a fixed rigid box, IPC contact + friction
  configs[4]  200k-tet soft block + 128x128 cloth + chain of 16 hinged rigid boxes, everything in contact with everything"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _contact_sim(S, thickness, kmin=None):
    st = S.default_settings()
    st.init_frictional_contact = 1
    st.mirror_state_to_host = 0
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = thickness
    if kmin:
        gp.min_contact_stiffness = kmin
    sim.set_contact_global_params(gp)
    return sim


def _run(sim, n_steps):
    its = 0
    for _ in range(n_steps):
        assert sim.run_one_step(), "simulation stopped"
        i = sim.info()
        assert i.last_newton_result in (0, 8), i.last_newton_result   # accepted, or redone after a constraint / stiffness hardening
        its += i.last_stats.newton_iterations
    return its


def test_cloth_256_on_box():
    from stark_amd import sim as S

    sim = _contact_sim(S, 1e-3)
    cloth = sim.add_surface_grid("cloth", (1.0, 1.0), (256, 256), S.cotton_fabric())
    sim.point_set_add_displacement(cloth, (0.0, 0.0, 0.05 + 0.0015))
    box = sim.add_rigid_box("box", 1.0, (2.0, 2.0, 0.1))
    sim.rb_add_constraint("fix", box)
    sim.set_friction(sim.contact_group("d", cloth), sim.contact_group("rb", box), 0.5)
    x_start = sim.points("x0")
    its = _run(sim, 5)
    x = sim.points("x0")
    assert np.isfinite(x).all()
    assert x[:, 2].min() > 0.05 - 1e-6                      # never below the top face of the box (no tunnelling)
    assert x[:, 2].max() < 0.05 + 2.5e-3                    # and rests inside the barrier range (it starts 1.5 mm above, d_hat = 2 mm)
    ci = sim.contact_info()
    assert ci["n_contacts"] > 10000 and ci["n_friction_contacts"] > 10000, ci
    assert its >= 1
    sim.close()


def test_mixed_scene_block_cloth_chain():
    from stark_amd import sim as S

    sim = _contact_sim(S, 1e-3, kmin=1e7)
    floor = sim.add_rigid_box("floor", 1.0, (4.0, 4.0, 0.1))
    sim.rb_add_constraint("fix", floor)
    block = sim.add_volume_grid("block", (0.0, 0.0, 0.05 + 0.0015 + 0.25), (0.5, 0.5, 0.5), (26, 26, 25), S.soft_rubber())   # 202 800 tets
    cloth = sim.add_surface_grid("cloth", (0.8, 0.8), (128, 128), S.cotton_fabric())
    sim.point_set_add_displacement(cloth, (0.0, 0.0, 0.05 + 0.0015 + 0.5 + 0.0015))
    links = []
    for k in range(16):
        b = sim.add_rigid_box("link%d" % k, 0.2, (0.08, 0.04, 0.04))
        sim.rb_set_translation(b, (-0.9 + 0.1 * k, 1.0, 0.4))
        links.append(b)
    sim.rb_add_constraint("fix", links[0])
    for k in range(15):
        sim.rb_add_constraint("hinge", links[k], links[k + 1], (-0.85 + 0.1 * k, 1.0, 0.4), (0.0, 1.0, 0.0))
    gf, gb, gc = sim.contact_group("rb", floor), sim.contact_group("d", block), sim.contact_group("d", cloth)
    sim.set_friction(gf, gb, 0.5)
    sim.set_friction(gb, gc, 0.5)
    its = _run(sim, 4)
    x = sim.points("x0")
    assert np.isfinite(x).all()
    info = sim.info()
    assert info.ndofs == 3 * info.n_points + 6 * 17
    ci = sim.contact_info()
    assert ci["n_contacts"] > 500, ci
    # the chain hangs from its fixed first link: the last link moved down, the first did not
    t0 = sim.rb_state(links[0])[0]
    t15 = sim.rb_state(links[15])[0]
    assert abs(t0[2] - 0.4) < 2e-3 and t15[2] < 0.4 - 1e-3
    assert its >= 1
    sim.close()
