"""CPU (build container only: needs oracle/_ref/shim_check, which `make -C oracle shim` builds from /root/reference): the B-upper
boundary of SURVEY.md 8(b). The UNMODIFIED reference (stark/src/** compiled in place against shim/include/symx, i.e. with
symx::NewtonsMethod replaced by the shim over the C ABI) builds a scene through its own stark::Simulation; its first time step makes
the shim register every DoF set, array and potential with a registration-only libmistark context (no GPU). What arrives at the C ABI
must be what this repo's host mirror (stark_amd/csrc/host/*) registers for the same scene: potential names in the same order,
connectivity strides, element counts, and for every binding the array (a DoF set by label, or the same array wherever the reference
binds the same array), its stride, connectivity column and item count. The engine itself checks every binding list against its kernel
(mistark_potential fails on a wrong count or stride), so a pass also means: every potential of a stock Simulation has its kernel."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHIM_CHECK = os.path.join(ROOT, "oracle", "_ref", "shim_check")

pytestmark = pytest.mark.skipif(not os.path.exists(SHIM_CHECK), reason="oracle/_ref/shim_check not built (needs /root/reference: make -C oracle shim)")


def _describe(h):
    from stark_amd import capi

    L = capi.lib()
    n = L.mistark_describe(h, None, 0)
    buf = C.create_string_buffer(n)
    L.mistark_describe(h, buf, n)
    return json.loads(buf.value.decode())


def _mirror(scene):
    """The same scene through this repo's host mirror on a registration-only context (settings.device = -1)."""
    from stark_amd import sim as S

    st = S.default_settings()
    st.device = -1
    st.init_frictional_contact = 1 if scene in ("blockbox", "mixed") else 0
    sim = S.Simulation(st)
    if scene == "mixed":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_scene import _build_mixed

        gp = S.contact_global_params()
        gp.default_contact_thickness = 0.002
        gp.min_contact_stiffness = 1e6
        sim.set_contact_global_params(gp)
        _build_mixed(S, sim, dict(nx=2, ny=2, nz=2, nc=4, nrb=3, L=0.4, gap=0.003, bx=1.2, bz=0.1, link=0.04, cloth=1.2, mu=0.5))
    elif scene == "blockbox":
        gp = S.contact_global_params()
        gp.default_contact_thickness = 1e-3
        sim.set_contact_global_params(gp)
        rb = sim.add_rigid_box("box", 1.0, (1.0, 1.0, 0.1))
        sim.rb_add_constraint("fix", rb)
        ps = sim.add_volume_grid("block", (0.0, 0.0, 0.6), (1.0, 1.0, 1.0), (2, 2, 2), S.soft_rubber())
        sim.set_friction(sim.contact_group("d", ps), sim.contact_group("rb", rb), 0.5)
    elif scene == "tetbeam":
        ps = sim.add_volume_grid("beam", (0.0, 0.0, 0.0), (4.0, 1.0, 1.0), (4, 1, 1), S.soft_rubber())
        sim.prescribe_inside_aabb(ps, (-2.0, 0.0, 0.0), (2e-3, 2.0, 2.0), 1e7)
    else:
        ps = sim.add_surface_grid("cloth", (0.4, 0.4), (4, 4), S.cotton_fabric())
        sim.prescribe_inside_aabb(ps, (-0.2, 0.0, 0.0), (2e-3, 2.0, 2.0), 1e7)
    sim.prepare()
    d = _describe(sim.engine_handle())
    sim.close()
    return d


def _normalise(d):
    """Per potential: bindings with arrays renamed by first appearance WITHIN the potential (DoF sets keep their label); the global
    identity of arrays across potentials is compared separately."""
    out = {}
    for p in d["potentials"]:
        local = {}
        bs = []
        for arr, stride, col, n in p["bindings"]:
            key = arr if arr.startswith("dof:") else "arr%d" % local.setdefault(arr, len(local))
            bs.append((key, stride, col, n))
        out[p["name"]] = dict(conn_stride=p["conn_stride"], n_elem=p["n_elem"], dynamic=p["dynamic"], bindings=bs)
    return out


@pytest.mark.parametrize("scene", ["blockbox", "mixed", "tetbeam", "cloth"])
def test_reference_classes_register_what_the_mirror_registers(scene, tmp_path):
    out = str(tmp_path / "shim.json")
    env = dict(os.environ, MISTARK_SHIM_DRY="1", MISTARK_SHIM_DESCRIBE=out, SHIM_SCRATCH=str(tmp_path))
    r = subprocess.run([SHIM_CHECK, scene], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    ref = json.load(open(out))
    ours = _mirror(scene)
    # DoF sets: same labels, sizes and order (Simulation.cpp:88-89: soft.v1 | rigid.v1 | rigid.w1)
    assert ref["dof_sets"] == ours["dof_sets"]
    R, O = _normalise(ref), _normalise(ours)
    # the reference registers all its potentials, used or not; the mirror the ones its scene can use: every potential with elements on
    # either side must be on both, identically; the mirror must not invent names
    assert set(O) <= set(R)
    used = {n for n, p in R.items() if p["n_elem"] > 0} | {n for n, p in O.items() if p["n_elem"] > 0}
    assert len(used) >= (15 if scene == "mixed" else 3)
    assert used and used <= set(O), sorted(used - set(O))
    for name in O:
        r_, o_ = R[name], O[name]
        assert r_["conn_stride"] == o_["conn_stride"], name
        if not name.startswith(("contact_", "friction_")):
            assert r_["n_elem"] == o_["n_elem"], name
        if name.startswith(("contact_", "friction_")):
            # (the reference's host-side detection has already filled its tables when the first solve registers them; the mirror's tables
            # are filled on the device)
            assert r_["dynamic"] == o_["dynamic"] == 1, name
            # the friction data arrays (T, mu, fn, bary) are empty until the first detection on both sides; item counts of the
            # mirror's device-filled arrays are 0 at registration
            strip = lambda bs: [(a, s, c) for a, s, c, _ in bs]
            assert strip(r_["bindings"]) == strip(o_["bindings"]), name
        else:
            assert r_["bindings"] == o_["bindings"], name
    # array identity across potentials: wherever the reference binds ONE array in two potentials, so does the mirror
    def identity(d):
        seen = {}
        for p in d["potentials"]:
            if p["name"] in O:
                for k, (arr, stride, col, n) in enumerate(p["bindings"]):
                    if not arr.startswith("dof:"):
                        seen.setdefault(arr, []).append((p["name"], k))
        return sorted(tuple(v) for v in seen.values())
    assert identity(ref) == identity(ours)


@pytest.mark.parametrize("scene", ["blockbox", "mixed"])
def test_the_collision_detection_stand_in_changes_nothing_that_is_registered(scene, tmp_path):
    """oracle/_ref/shim_check_cd: the unmodified stark/src/** compiled against shim/include_cd/TriangleMeshCollisionDetection (the reference's
    detector dependency replaced by include/mistark_tmcd.h) instead of the dependency's own header. In registration-only mode (no GPU: the
    stand-in returns empty lists) EnergyFrictionalContact registers the same DoF sets and the same potentials with the same bindings as with
    its own detector; only the number of rows its host detection had already found differs."""
    exe = SHIM_CHECK + "_cd"
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_check_cd not built (make -C oracle shim_cd)")
    got = []
    for binary in (SHIM_CHECK, exe):
        out = str(tmp_path / (os.path.basename(binary) + ".json"))
        env = dict(os.environ, MISTARK_SHIM_DRY="1", MISTARK_SHIM_DESCRIBE=out, SHIM_SCRATCH=str(tmp_path))
        r = subprocess.run([binary, scene], env=env, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got.append(json.load(open(out)))
    assert got[0]["dof_sets"] == got[1]["dof_sets"]
    A, B = _normalise(got[0]), _normalise(got[1])
    assert set(A) == set(B) and len(A) >= 35
    strip = lambda bs: [(a, s, c) for a, s, c, _ in bs]
    for name in A:
        assert A[name]["conn_stride"] == B[name]["conn_stride"] and A[name]["dynamic"] == B[name]["dynamic"], name
        assert strip(A[name]["bindings"]) == strip(B[name]["bindings"]), name
        if not name.startswith(("contact_", "friction_")):
            assert A[name]["n_elem"] == B[name]["n_elem"] and A[name]["bindings"] == B[name]["bindings"], name


@pytest.mark.parametrize("scene,name,n_bindings", [("magnetic", "EnergyMagneticAttraction", 5), ("foreach", "EnergyMultipoleAttraction", 5)])
def test_user_defined_potential_takes_the_sequence_branch_of_the_shim(scene, name, n_bindings, tmp_path):
    """A name the engine has no kernel for (README.md:109-126 of the reference): the shim flattens the expression into SymX's op sequence and
    registers it with mistark_potential_custom; `foreach` carries a summation loop (MappedWorkspace::add_for_each), whose symbols get a binding
    of their own (a global slot of 4 doubles: the pole) at the place the summation vector was made."""
    out = str(tmp_path / "shim.json")
    env = dict(os.environ, MISTARK_SHIM_DRY="1", MISTARK_SHIM_DESCRIBE=out, SHIM_SCRATCH=str(tmp_path))
    r = subprocess.run([SHIM_CHECK, scene], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.load(open(out))
    p = [q for q in d["potentials"] if q["name"] == name]
    assert len(p) == 1 and p[0]["n_elem"] == 4 ** 3 + 3 ** 3 and p[0]["conn_stride"] == 1
    bs = p[0]["bindings"]
    assert len(bs) == n_bindings
    assert bs[0][0].startswith("dof:") and bs[0][1:3] == [3, 0]           # v1 of the point
    assert [b[1] for b in bs] == ([3, 3, 1, 1, 3] if scene == "magnetic" else [3, 3, 1, 1, 4])
    assert [b[2] for b in bs][2:] == [-1, -1, -1]                          # dt, k and the magnet centre / the pole slot are global
