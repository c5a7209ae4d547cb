"""Developer tool: where the time of the tet evaluation + assembly goes (1M-tet block, run on the GPU box).
Times mistark_eval(P+g+H) and mistark_assemble back to back with the measurement switches of option "kernel_dbg"
(1: no gradient atomics, 2: no global stores of the Hessian blocks), on the lazy (float upper-triangle pool) and the full (double pool) path."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import capi
from stark_amd import sim as S

sim = build_scene(S, 44, 44, 43, 0, scene="clamped")
sim.run_one_step()
L = capi.lib()
h = sim.engine_handle()
E = C.c_double()


def t_eval(n=20):
    L.mistark_eval(h, 2, C.byref(E), None)
    t0 = time.perf_counter()
    for _ in range(n):
        L.mistark_eval(h, 2, C.byref(E), None)
    return 1e6 * (time.perf_counter() - t0) / n


def t_asm(n=20):
    L.mistark_assemble(h)
    L.mistark_sync(h)
    t0 = time.perf_counter()
    for _ in range(n):
        L.mistark_assemble(h)
    L.mistark_sync(h)
    return 1e6 * (time.perf_counter() - t0) / n


for lazy in (0, 1):
    L.mistark_set_option(h, b"lazy_eval", lazy)
    for dbg in (0, 1, 2, 3):
        L.mistark_set_option(h, b"kernel_dbg", dbg)
        te = t_eval()
        ta = t_asm() if dbg == 0 else float("nan")
        print("lazy=%d kernel_dbg=%d  eval(all potentials + read-back) %.1f us   assemble %.1f us" % (lazy, dbg, te, ta))
