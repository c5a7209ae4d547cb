"""Developer tool: evaluation + assembly of the clamped 1M-tet block in a loop (usage: rocprofv3 --kernel-trace --stats -d DIR -- python tools/gather_prof.py)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import capi
from stark_amd import sim as S

sim = build_scene(S, 44, 44, 43, 0, scene="clamped")
L = capi.lib()
sim.prepare()
h = sim.engine_handle()
E = C.c_double()
L.mistark_set_option(h, b"lazy_eval", 1)
for _ in range(20):
    assert L.mistark_eval(h, 2, C.byref(E), None) == 0
    assert L.mistark_assemble(h) == 0
