"""Developer tool: SpMV micro-benchmark sweep on the 1M-tet block matrix (run on the GPU box); GRID=88,88,86 in the environment: the 8M-tet
matrix (790 MB: out of reach of the Infinity Cache).

Variants (mistark_set_option "spmv_variant"): 0 = the solver's launch (static chunks + contact part, contact rows left to the consumer),
1 = the same without the row reduction (loads + block products), 3 = without the matrix value loads, 9 = plain float4 stream of the value
buffer (floor of the memory system for this matrix), 11 = loads + gather + products in the simplest possible loop, 12 = the solver's
kernel on the static part with the input vector in SoA layout (x[n] | y[n] | z[n]) instead of interleaved (north_star's "SoA node/DoF arrays")."""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
from bench import build_scene
from stark_amd import capi
from stark_amd import sim as S

nx, ny, nz = [int(v) for v in os.environ.get("GRID", "44,44,43").split(",")]
sim = build_scene(S, nx, ny, nz, 0)
sim.run_one_step()
L = capi.lib()
h = sim.engine_handle()
_, _, nbytes = sim.spmv_timing()
for variant in [int(v) for v in os.environ.get("VARIANTS", "0,1,3,9,11").split(",")]:
  for cap in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]:
    L.mistark_set_option(h, b"spmv_grid_cap", cap)
    L.mistark_set_option(h, b"spmv_variant", variant)
    for nt in [int(v) for v in os.environ.get("NT", "-1").split(",")]:
     L.mistark_set_option(h, b"spmv_nt", nt)
     us = C.c_double()
     L.mistark_spmv_bench(h, 200, C.byref(us))
     print("variant=%d nt=%d grid_cap=%d  %.2f us  %.0f GB/s  (%.1f%% of 8 TB/s)" % (variant, nt, cap, us.value, nbytes / us.value / 1e3, 100 * nbytes / us.value / 1e3 / 8000))
