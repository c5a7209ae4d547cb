#!/usr/bin/env python3
"""Longer run of the bench scene (configs[3]): N time steps, every step must be accepted; prints per-step Newton counts and the rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from stark_amd import sim as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sim = bench.build_scene(S, 44, 44, 43, 0, "contact")
its, t0 = [], time.perf_counter()
for k in range(n):
    assert sim.run_one_step(), k
    its.append(sim.info().last_stats.newton_iterations)
wall = time.perf_counter() - t0
i = sim.info()
x = sim.points("x0")
print("steps", n, "failed", i.failed_steps, "newton", sum(its), "per step", its)
print("wall %.2f s, %.1f Newton-steps/s, z range %.4f..%.4f, finite %s, contacts %s" % (wall, sum(its) / wall, x[:, 2].min(), x[:, 2].max(), np.isfinite(x).all(), sim.contact_info()))
