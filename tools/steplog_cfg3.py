"""Developer tool: per-time-step (Newton iterations, linear solves, CG iterations of every Newton iteration's last solve) of configs[3] on the
engine, to lay beside `oracle/_ref/ref_harness time blockbox ... per_step` (MISTARK_NEWTON_TRACE=1: per-iteration lines like the
reference's Verbosity::Full). usage: steplog_cfg3.py [attempts] [ox oy]"""
import sys

sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import sim as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
off = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 0.0)
sim = build_scene(S, 44, 44, 43, 0, offset=off)
prev = (0, 0)
for s in range(n):
    print("--- time step", s, file=sys.stderr, flush=True)
    assert sim.run_one_step()
    i = sim.info()
    cur = (i.total_newton_iterations, i.total_linear_solves)
    log = sim.newton_iteration_log()
    print(s, [c - p for c, p in zip(cur, prev)] + [sum(r.cg_iterations_last for r in log if r.logged)], [r.cg_iterations_last for r in log], flush=True)
    prev = cur
