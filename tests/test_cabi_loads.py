"""CPU: libmistark.so loads without a GPU and exports every entry point the three public headers declare; entry points that need
a device fail loudly instead of falling back to anything."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_exports_every_declared_symbol():
    from stark_amd import capi

    L = capi.lib()
    names = capi.exported_symbols()
    assert len(names) > 80
    missing = [s for s in names if not hasattr(L, s)]
    assert not missing, missing


def test_host_only_entry_points_work_without_gpu():
    from stark_amd import capi

    L = capi.lib()
    b, e = C.c_int64(), C.c_int64()
    covered = 0
    for r in range(8):
        assert L.mistark_shard_range(998976, r, 8, C.byref(b), C.byref(e)) == 0
        assert b.value == covered
        covered = e.value
    assert covered == 998976
    assert L.mistark_shard_range(10, 3, 2, C.byref(b), C.byref(e)) < 0
    assert L.mistark_n_supported_potentials() == 63
    stride = C.c_int32()
    assert L.mistark_contact_recipe(b"contact_rb_d_pt_tp_cubic", C.byref(stride), None, None, None) == 14 and stride.value == 7


def test_no_silent_cpu_fallback():
    """Without a visible MI355X the engine cannot be created: there is no CPU path behind the C ABI."""
    import torch

    from stark_amd import capi

    if torch.cuda.is_available():
        return  # (on the GPU box this is covered by the -m gpu tests actually running kernels)
    L = capi.lib()
    h = C.c_void_p()
    assert L.mistark_create(0, C.byref(h)) != 0
    # ... nor behind the standalone collision detector (include/mistark_tmcd.h)
    d = C.c_void_p()
    assert L.mistark_cd_create(C.byref(d), 0) != 0 and not d.value


def test_headers_are_plain_c():
    """The drop-in boundary is a C ABI: every header under include/ must compile as C99 on its own (plain pointers and sizes only)."""
    import glob
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for h in sorted(glob.glob(os.path.join(root, "include", "*.h"))):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I" + os.path.join(root, "include"), h], capture_output=True)
        assert r.returncode == 0, (h, r.stderr.decode())
