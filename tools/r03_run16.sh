timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "projection_updates" 2>&1 | tail -5
for i in 1 2 3 4 5 6; do timeout 300 python - <<'PY'
import json, os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_attach_by_distance as T
from stark_amd import sim as S
z = np.load(os.path.join(T.GOLDEN, "traj_attachdist.npz"))
traj = json.loads(bytes(z["traj_json"]).decode()); man = json.loads(bytes(z["manifest_json"]).decode()); sc = traj["scene"]
names = [q["name"] for q in man["potentials"]]
tri = z["p%d_conn" % names.index("EnergyTriangleStrain")]
cloth_tri = tri[tri[:, 2:5].max(axis=1) < (sc["n"] + 1) ** 2][:, 2:5]
sim, box, h3, hb = T.build(S, sc, cloth_tri, 0)
its = []
for _ in traj["steps"]:
    assert sim.run_one_step(); its.append(sim.info().last_stats.newton_iterations)
x = sim.points("x0").copy()
print(its, traj["newton_iterations"], float(np.abs(x - z["x_end"]).max() / np.abs(z["x_end"]).max()), float(x.sum()))
PY
done
