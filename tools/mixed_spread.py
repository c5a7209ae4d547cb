"""Run-to-run spread of the small mixed scene (tests/golden/traj_cfg4_mixed_small.npz) on the GPU path."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from stark_amd import sim as S
from test_gpu_scene import _build_mixed, _contact_sim, _load
z, traj, man = _load("traj_cfg4_mixed_small")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    sim = _contact_sim(S, traj["scene"])
    _build_mixed(S, sim, traj["scene"])
    its = []
    for step in range(len(traj["steps"])):
        assert sim.run_one_step()
        its.append(sim.info().last_stats.newton_iterations)
    print(its, "ref", traj["newton_iterations"], "dx %.2e dv %.2e" % (np.abs(sim.points("x0") - z["x_end"]).max(), np.abs(sim.points("v0") - z["v_end"]).max()))
    sim.close()
