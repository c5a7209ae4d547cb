import os, sys
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
t = torch.ones(4, device="cuda"); torch.cuda.synchronize()
import numpy as np, ctypes as C
import stark_amd
eng = stark_amd.Engine(0)
x = np.linspace(-3.0, 5.0, 1000); y = x.copy()
rc = eng.L.mistark_dist_rccl_selftest(eng.h, y.ctypes.data, len(y))
print("selftest rc", rc, (x == y).all(), eng.L.mistark_last_error(eng.h))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "amdhip64" in l or "rccl" in l or "hsa-runtime" in l})
print("\n".join(libs))
dist.destroy_process_group()
