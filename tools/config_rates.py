#!/usr/bin/env python3
"""Newton-steps/s and host stage timers of the BASELINE configs other than the bench's (configs[0], [1], [2], [4]) on one GPU.
Usage (GPU box): python tools/config_rates.py [cfg0] [cfg1] [cfg2] [cfg2tilt] [cfg4]   -> one JSON line per config."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from stark_amd import sim as S  # noqa: E402


def contact_sim(thickness, kmin=None):
    st = S.default_settings()
    st.mirror_state_to_host = 0
    st.init_frictional_contact = 1
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = thickness
    if kmin is not None:
        gp.min_contact_stiffness = kmin
    sim.set_contact_global_params(gp)
    return sim


def cfg0():
    """README hello world: 32 x 32 cloth on a box turned by the per-step script."""
    sim = contact_sim(0.0025)
    sim.add_surface_grid("cloth", (0.4, 0.4), (32, 32), S.cotton_fabric())
    box = sim.add_rigid_box("box", 1.0, (0.08, 0.08, 0.08))
    anchor = (0.0, 0.0, -0.08)
    sim.rb_add_translation(box, anchor)
    fix = sim.rb_add_fix(box)
    sim.script = lambda: sim.rb_fix_set_transformation(fix, anchor, 90.0 * sim.info().current_time, (0.0, 0.0, 1.0))
    return sim, 10


def cfg1():
    st = S.default_settings()
    st.mirror_state_to_host = 0
    st.init_frictional_contact = 0
    sim = S.Simulation(st)
    ps = sim.add_volume_grid("beam", (0.0, 0.0, 0.0), (4.0, 1.0, 1.0), (52, 13, 13), S.soft_rubber())
    sim.prescribe_inside_aabb(ps, (-2.0, 0.0, 0.0), (2e-3, 2.0, 2.0), 1e7)
    return sim, 6


def cfg2():
    sim = contact_sim(1e-3)
    floor = sim.add_rigid_box("floor", 1.0, (2.0, 2.0, 0.1))
    sim.rb_add_constraint("fix", floor)
    cloth = sim.add_surface_grid("cloth", (1.0, 1.0), (256, 256), S.cotton_fabric())
    sim.point_set_add_displacement(cloth, (0.0, 0.0, 0.05 + 0.0015))
    sim.set_friction(sim.contact_group("rb", floor), sim.contact_group("d", cloth), 0.5)
    return sim, 10


def cfg2tilt():
    """configs[2] as a well-posed dynamic scene (round 6): the cloth tilted 3 degrees, lowest edge at the contact distance, released; the warm-up
    step is the first landing step, the timed ones the two that follow and the settling (fixture steplog_cfg2_tilted_256)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from steplog_cfg2 import build

    return build(0.002, tilt=3.0), 7


def cfg4():
    from test_gpu_scene import _build_mixed

    sim = contact_sim(1e-3, 1e8)
    _build_mixed(S, sim, dict(nx=26, ny=26, nz=25, nc=128, nrb=16, L=1.0, gap=0.0015, bx=3.0, bz=0.1, link=0.05, cloth=1.2, mu=0.5))
    return sim, 3


def main():
    names = sys.argv[1:] or ["cfg0", "cfg1", "cfg2", "cfg2tilt", "cfg4"]
    for name in names:
        sim, steps = globals()[name]()
        script = getattr(sim, "script", lambda: None)
        script()
        assert sim.run_one_step()  # warm-up: pattern construction, first detection
        a = sim.info()
        t0 = time.perf_counter()
        for _ in range(steps):
            script()
            assert sim.run_one_step()
        wall = time.perf_counter() - t0
        b = sim.info()
        d = lambda f: getattr(b, f) - getattr(a, f)  # noqa: E731
        n = max(d("total_newton_iterations"), 1)
        print(json.dumps({
            "config": name, "ndofs": b.ndofs, "time_steps": steps, "newton_iterations": d("total_newton_iterations"), "wall_s": round(wall, 4),
            "newton_steps_per_s": round(d("total_newton_iterations") / wall, 2), "linear_solves": d("total_linear_solves"),
            "cg_iterations_per_solve": round(d("total_cg_iterations") / max(d("total_linear_solves"), 1), 1),
            "ms_per_linear_solve": round(1e3 * d("total_linear_solve_time") / max(d("total_linear_solves"), 1), 3),
            # per-iteration costs: what stays comparable from round to round when another summation order sends a long contact run down another path
            "ms_per_newton_iteration": round(1e3 * wall / n, 3), "us_per_cg_iteration": round(1e6 * d("total_linear_solve_time") / max(d("total_cg_iterations"), 1), 2),
            "cg_iterations": d("total_cg_iterations"),
            "ms_per_newton": {k: round(1e3 * d("total_%s_time" % k) / n, 3) for k in ("newton", "linear_solve", "eval_pgh", "eval_p", "project", "assembly", "callback", "step")},
            "evaluations": d("total_evaluations"), "contact": sim.contact_info() if name != "cfg1" else None}))
        sim.close()


if __name__ == "__main__":
    main()
