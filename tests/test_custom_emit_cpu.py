"""CPU: the emitter of user-defined potentials (stark_amd/csrc/custom.hip; include/mistark.h mistark_custom_emit) without a GPU — the HIP source
it writes for the reference's own op sequences (symx::Sequence of EnergyTriangleStrain: two branches; EnergyDiscreteShells: acos; a conditional
rigid-body constraint) and that hipRTC compiles it for gfx950 (hipcc's runtime compiler needs no device). What the kernels compute is the GPU
suite's business (tests/test_gpu_custom_rtc.py, the custom_rtc variant of tests/test_gpu_parity.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import evaluator as ev

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _emit(pot, prob, z, pi, compile_it):
    from stark_amd import capi

    L = capi.lib()
    L.mistark_custom_emit.restype = C.c_int64
    L.mistark_custom_emit.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_char_p, C.c_int64]
    ops = np.ascontiguousarray(z["p%d_ops" % pi], dtype=np.int32).reshape(-1, 5)
    cst = np.ascontiguousarray(z["p%d_opsc" % pi], dtype=np.float64)
    has_c = ("p%d_cops" % pi) in z
    cops = np.ascontiguousarray(z["p%d_cops" % pi], dtype=np.int32).reshape(-1, 5) if has_c else None
    ccst = np.ascontiguousarray(z["p%d_copsc" % pi], dtype=np.float64) if has_c else None
    strides = np.array([b.stride for b in pot.bindings], dtype=np.int32)
    n_in = int(strides.sum())
    set_of = {a: s for s, a in prob.dof_arrays.items()}
    in_dof = -np.ones(n_in, dtype=np.int32)
    blk = 0
    for s in sorted(prob.dof_arrays):   # local DoF blocks: DoF sets in registration order, then binding order (SecondOrderCompiledPotential.cpp:10-33)
        o = 0
        for b in pot.bindings:
            if set_of.get(b.array) == s:
                in_dof[o:o + 3] = 3 * blk + np.arange(3)
                blk += 1
            o += b.stride
    out = C.create_string_buffer(1 << 21)
    r = L.mistark_custom_emit(pot.name.encode(), strides.ctypes.data, len(strides), in_dof.ctypes.data, ops.ctypes.data, cst.ctypes.data, len(ops), n_in,
                              cops.ctypes.data if has_c else None, ccst.ctypes.data if has_c else None, len(cops) if has_c else 0, blk, 1 if compile_it else 0, out, 1 << 21)
    return r, out.value.decode(), len(ops), has_c


@pytest.mark.parametrize("fixture,name", [("cloth_shells_6", "EnergyTriangleStrain"), ("cloth_shells_6", "EnergyDiscreteShells"), ("rbchain", "rb_constraint_angle_limits"),
                                          ("tetbeam_full_4x1x1", "EnergyTetStrain")])
def test_emitted_source_has_one_statement_per_op_and_compiles(fixture, name, tmp_path, monkeypatch):
    monkeypatch.setenv("MISTARK_RTC_CACHE", str(tmp_path))   # (a build, not a cache hit)
    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, fixture + ".npz"))
    pi = [p.name for p in prob.potentials].index(name)
    n, src, n_ops, has_c = _emit(prob.potentials[pi], prob, z, pi, False)
    assert n > 0 and "prog_energy" in src and ("prog_condition" in src) == has_c
    for k in range(3):
        assert 'extern "C" __global__ __launch_bounds__(256) void mistark_custom_k%d(PotArgs a' % k in src
    body = src[src.index("HDual prog_energy"):]
    body = body[:body.index("return out;")]
    ops = np.ascontiguousarray(z["p%d_ops" % pi], dtype=np.int32).reshape(-1, 5)
    n_branch_if = int(((ops[:, 0] == 2) & (ops[:, 4] != -2) & (ops[:, 2] == 0)).sum())
    n_assign = int(((ops[:, 0] != 2) & (ops[:, 0] != 5)).sum())
    assert body.count(".v > 0.0) {") == n_branch_if                       # Branch markers became real branches
    assert body.count(" = ") - body.count("HDual out(0.0)") >= n_assign  # one assignment per op (+ the input seeds)
    if name == "EnergyTriangleStrain":
        assert n_branch_if >= 2 and "} else {" in body
    size, msg, _, _ = _emit(prob.potentials[pi], prob, z, pi, True)
    assert size > 4096, msg                                               # a gfx950 code object came out of hipRTC
    assert len(os.listdir(tmp_path)) == 1 and os.listdir(tmp_path)[0].endswith("_gfx950.hsaco")
    # the cached file is bound to its source: magic, source length, then hashes; the code object (ELF) follows the 40-byte header
    raw = open(os.path.join(tmp_path, os.listdir(tmp_path)[0]), "rb").read()
    assert raw[:8] == b"MISRTC02" and int.from_bytes(raw[8:16], "little") > 1000 and raw[40:44] == b"\x7fELF"
    assert int.from_bytes(raw[32:40], "little") == len(raw) - 40 == size
    # the same source again is answered from the cache
    size2, _, _, _ = _emit(prob.potentials[pi], prob, z, pi, True)
    assert size2 == size


def test_the_code_object_cache_only_trusts_what_is_provably_the_users_own(tmp_path, monkeypatch):
    """ADVICE r05 (custom.hip rtc_build): a code object found in the cache is run on the device. A planted file under the predictable name (no
    valid header / another source), a symlink in its place, or a cache directory that group or others can write must never be loaded."""
    import stat

    prob, man, z = ev.load_fixture(os.path.join(GOLDEN, "cloth_shells_6.npz"))
    pi = [p.name for p in prob.potentials].index("EnergyDiscreteShells")
    pot = prob.potentials[pi]
    good = tmp_path / "good"
    monkeypatch.setenv("MISTARK_RTC_CACHE", str(good))
    size, msg, _, _ = _emit(pot, prob, z, pi, True)                     # (creates the directory 0700 before the lookup)
    assert size > 4096, msg
    assert stat.S_IMODE(os.stat(good).st_mode) == 0o700
    (name,) = os.listdir(good)
    real = (good / name).read_bytes()
    assert stat.S_IMODE(os.stat(good / name).st_mode) == 0o600
    # 1. a planted file under the right name: garbage, an ELF without the header, a header of another source -> ignored, rebuilt, replaced
    other = bytearray(real)
    other[8:16] = (int.from_bytes(real[8:16], "little") + 1).to_bytes(8, "little")        # another source length
    for planted in (b"\x7fELF" + b"\0" * 5000, real[40:], bytes(other)):
        (good / name).write_bytes(planted)
        size2, msg, _, _ = _emit(pot, prob, z, pi, True)
        assert size2 == size, msg
        assert (good / name).read_bytes()[:8] == b"MISRTC02" and (good / name).read_bytes() != planted
    # 2. a symlink in the file's place is not followed
    target = tmp_path / "elsewhere.hsaco"
    target.write_bytes(real)
    os.remove(good / name)
    os.symlink(target, good / name)
    size3, msg, _, _ = _emit(pot, prob, z, pi, True)
    assert size3 == size, msg
    assert not os.path.islink(good / name)                                # (rename() replaced the link by the freshly built file)
    # 3. a directory open to others is not used at all: nothing is read from it and nothing written into it
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    (loose / name).write_bytes(b"\x7fELF" + b"\0" * 5000)
    monkeypatch.setenv("MISTARK_RTC_CACHE", str(loose))
    size4, msg, _, _ = _emit(pot, prob, z, pi, True)
    assert size4 == size, msg
    assert os.listdir(loose) == [name] and (loose / name).read_bytes()[:8] != b"MISRTC02"
    # 4. a symlinked directory likewise
    link = tmp_path / "link"
    os.symlink(good, link)
    monkeypatch.setenv("MISTARK_RTC_CACHE", str(link))
    before = sorted(os.listdir(good))
    size5, msg, _, _ = _emit(pot, prob, z, pi, True)
    assert size5 == size and sorted(os.listdir(good)) == before


def test_a_sequence_that_overwrites_an_input_is_refused_with_a_message():
    from stark_amd import capi

    L = capi.lib()
    L.mistark_custom_emit.restype = C.c_int64
    ops = np.array([[8, 0, 0, 0, -1], [5, 0, 0, -1, -1]], dtype=np.int32)   # Mul writes input 0
    cst = np.zeros(2)
    strides = np.array([1], dtype=np.int32)
    in_dof = np.array([-1], dtype=np.int32)
    out = C.create_string_buffer(4096)
    L.mistark_custom_emit.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_char_p, C.c_int64]
    r = L.mistark_custom_emit(b"Bad", strides.ctypes.data, 1, in_dof.ctypes.data, ops.ctypes.data, cst.ctypes.data, 2, 1, None, None, 0, 0, 0, out, 4096)
    assert r < 0 and b"overwrites an input" in out.value
