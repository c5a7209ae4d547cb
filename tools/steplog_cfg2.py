"""Developer tool: configs[2] as BASELINE describes it — a 256 x 256 Cotton_Fabric cloth DROPPED from 5 cm on a fixed floor (the scene of
`oracle/_ref/ref_harness time clothbox n=256 size=1 box=2 gap=0.05 thickness=0.001 mu=0.5`, fixture steplog_cfg2_clothbox_drop_256) — on the
engine: per time-step attempt [Newton iterations, linear solves, CG iterations], then a JSON summary. usage: steplog_cfg2.py [attempts] [gap] [dt|-] [tilt]
(round 6: gap = 0.002, tilt = 3 is the well-posed dynamic variant of configs[2], fixture steplog_cfg2_tilted_256)"""
import json
import sys
import time

sys.path.insert(0, ".")
from stark_amd import sim as S


def build(gap=0.05, n=256, size=1.0, box=2.0, thickness=1e-3, mu=0.5, dt=None, tilt=0.0):
    st = S.default_settings()
    if dt is not None:
        st.max_time_step_size = dt
    st.mirror_state_to_host = 0
    st.init_frictional_contact = 1
    sim = S.Simulation(st)
    gp = S.contact_global_params()
    gp.default_contact_thickness = thickness
    sim.set_contact_global_params(gp)
    ps = sim.add_surface_grid("cloth", (size, size), (n, n), S.cotton_fabric())
    if tilt != 0.0:   # (oracle/ref_harness.cpp scene_clothbox `tilt`: lowest edge `gap` above the floor, the cloth rising from there)
        import math
        sim.point_set_add_rotation(ps, tilt, (0.0, 1.0, 0.0))
        sim.point_set_add_displacement(ps, (0.0, 0.0, 0.5 * size * math.sin(abs(tilt) * math.pi / 180.0)))
    rb = sim.add_rigid_box("box", 1.0, (box, box, box))
    sim.rb_add_translation(rb, (0.0, 0.0, -0.5 * box - gap))
    sim.rb_add_constraint("fix", rb)
    sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), mu)
    return sim


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    dt = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    tilt = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    sim = build(gap, dt=dt, tilt=tilt)
    prev = (0, 0, 0)
    per_step = []
    t0 = time.perf_counter()
    t_first = None
    walls = []
    for s in range(n):
        t_a = time.perf_counter()
        ok = sim.run_one_step()
        walls.append(time.perf_counter() - t_a)
        i = sim.info()
        cur = (i.total_newton_iterations, i.total_linear_solves, i.total_cg_iterations)
        per_step.append([c - p for c, p in zip(cur, prev)])
        print(s, per_step[-1], "t=%.4f dt=%.5f" % (i.current_time, i.dt), "" if ok else "STOPPED", flush=True)
        prev = cur
        if t_first is None:
            t_first = time.perf_counter()
        if not ok:
            break
    wall = time.perf_counter() - t_first
    i = sim.info()
    newton = sum(p[0] for p in per_step[1:])
    x = sim.points("x0")
    # the steady part of the run: attempts 8.. (behind the first time step's failed attempts and the two badly conditioned steps after them)
    tail_newton = sum(p[0] for p in per_step[8:])
    tail_wall = sum(walls[8:])
    print(json.dumps({"config": "cfg2drop" if tilt == 0.0 else "cfg2tilt", "gap": gap, "tilt": tilt, "newton_steps_per_s_attempts_8_on": round(tail_newton / tail_wall, 2) if tail_wall > 0 else None,
                      "ms_per_attempt_8_on": round(1e3 * tail_wall / max(len(walls) - 8, 1), 3), "cg_per_solve_attempts_8_on": round(sum(p[2] for p in per_step[8:]) / max(sum(p[1] for p in per_step[8:]), 1), 1), "dt_max": dt, "z_min": float(x[:, 2].min()), "z_max": float(x[:, 2].max()), "attempts": len(per_step), "time": i.current_time, "newton_iterations": sum(p[0] for p in per_step), "linear_solves": sum(p[1] for p in per_step),
                      "cg_iterations": sum(p[2] for p in per_step), "wall_s_after_first": round(wall, 4), "newton_steps_per_s_after_first": round(newton / wall, 2) if wall > 0 else None,
                      "ms_per_attempt_after_first": round(1e3 * wall / max(len(per_step) - 1, 1), 3), "per_step": per_step, "contact": sim.contact_info()}))
    sim.close()
