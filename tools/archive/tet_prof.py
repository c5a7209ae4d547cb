"""Developer tool: the tet evaluation alone under rocprofv3 (usage: rocprofv3 --kernel-trace --stats -d DIR -- python tools/tet_prof.py <kernel_dbg>)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import capi
from stark_amd import sim as S

sim = build_scene(S, 44, 44, 43, 0, scene="clamped")
sim.run_one_step()
L = capi.lib()
h = sim.engine_handle()
E = C.c_double()
L.mistark_set_option(h, b"lazy_eval", 1)
L.mistark_set_option(h, b"kernel_dbg", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(30):
    L.mistark_eval(h, 2, C.byref(E), None)
