import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from stark_amd import capi
from stark_amd import sim as S
from test_gpu_fullsize import _Eng
st = S.default_settings(); st.mirror_state_to_host = 0
sim = S.Simulation(st)
grid = (44, 44, 43)
ps = sim.add_volume_grid("block", (0, 0, 0), (1.0, 1.0, 1.0), grid, S.soft_rubber())
sim.prescribe_inside_aabb(ps, (-0.5, 0, 0), (2e-3, 10.0, 10.0), 1e7)
sim.prepare(); sim.begin_time_step()
eng = _Eng(sim); n = eng.ndofs
eng.set_dofs(1e-3 * np.sin(1.3 * np.arange(n) + 0.7))
eng.eval(capi.EVAL_P_G_H); eng.project(1e-10, False, None); eng.assemble()
b = np.cos(0.11 * np.arange(n))
for rep in range(2):
    t = time.time(); x, ok = eng.direct_llt(b); dt = time.time() - t
    r = eng.spmv(x) - b
    print(n, ok, "residual %.1e" % (np.linalg.norm(r) / np.linalg.norm(b)), "%.2f s" % dt)
