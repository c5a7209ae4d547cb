"""GPU (-m gpu): SURVEY 8(f)-2 end to end through the drop-in. oracle/_ref/shim_check is the UNMODIFIED reference compiled against the shim
(shim/include/symx) and linked with libmistark.so; its scenes `magnetic` and `foreach` register user-defined potentials exactly as the
reference's README.md:109-126 / examples/main.cpp:666-692 do — names the engine has no kernel for, so the shim takes its `symx::Sequence`
branch (shim/src/NewtonsMethod.cpp, "no hand-written kernel under this name") and the energy runs on the device interpreter (csrc/custom.hip),
`foreach` with a summation loop (MappedWorkspace::add_for_each). Expected values: the reference itself on the same scenes
(tests/golden/traj_user_*.npz, written by oracle/ref_harness `traj magnetic|foreach`).
Also here: what the shim re-sends when a Newton callback rewrites a LARGE bound array in place."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "shim_check")
needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/shim_check not built (needs /root/reference: make -C oracle shim)")


def _run(scene, steps, tmp_path, env=None, tag=""):
    out = str(tmp_path / ("%s%s.json" % (scene, tag)))
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([EXE, scene, str(steps), out], capture_output=True, timeout=900, env=e)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-1500:])
    return json.load(open(out)), r.stderr.decode()


@needs_exe
@pytest.mark.parametrize("scene", ["magnetic", "foreach"])
def test_user_defined_potential_through_the_shim_equals_the_reference(scene, tmp_path):
    z = np.load(os.path.join(ROOT, "tests", "golden", "traj_user_%s_3.npz" % scene))
    man = json.loads(bytes(z["traj_json"]).decode())
    desc = str(tmp_path / "describe.json")
    got, _ = _run(scene, len(man["steps"]), tmp_path, {"SHIM_MAGNET_K": "20", "MISTARK_SHIM_DESCRIBE": desc})
    # the potential arrived as an op sequence under its own name
    names = [p["name"] for p in json.load(open(desc))["potentials"]]
    assert {"magnetic": "EnergyMagneticAttraction", "foreach": "EnergyMultipoleAttraction"}[scene] in names
    assert got["newton_iterations"] == man["newton_iterations"]
    assert abs(got["time"] - man["steps"][-1]["time"]) <= 1e-12
    x = np.array(got["x"])
    assert np.abs(x - z["x_end"]).max() <= 1e-6 * max(1.0, np.abs(z["x_end"]).max())
    # and the user potential does something: the free top of the block is pulled up against gravity, towards the magnet / the poles
    plain, _ = _run(scene, len(man["steps"]), tmp_path, {"SHIM_MAGNET_K": "0"}, "_k0")
    assert np.abs(np.array(plain["x"]) - x).max() > 1e-4


@needs_exe
def test_large_array_rewritten_in_place_by_a_newton_callback_reaches_the_engine(tmp_path):
    """scene `inplace`: a before_energy_evaluation callback moves all 12 691 targets (38 073 doubles > the 32 768 the shim reads in full at
    every evaluation) of a user potential at the 4th evaluation of the run — same address, same size. Default mode (sampled fingerprints) must
    end where MISTARK_SHIM_STRICT=1 (everything read in full at every evaluation) ends; MISTARK_SHIM_FAST=1 (address and size only, the
    round-3 behaviour) does not."""
    strict, _ = _run("inplace", 2, tmp_path, {"MISTARK_SHIM_STRICT": "1"}, "_strict")
    default, err = _run("inplace", 2, tmp_path, {"MISTARK_SHIM_STATS": "1"}, "_default")
    fast, _ = _run("inplace", 2, tmp_path, {"MISTARK_SHIM_FAST": "1"}, "_fast")
    xs, xd, xf = (np.array(r["x"]) for r in (strict, default, fast))
    assert default["newton_iterations"] == strict["newton_iterations"]
    assert np.array_equal(xd, xs)                       # the same bytes reached the engine at the same evaluation: identical runs
    assert "WARNING" not in err and " found by the sampled check inside the Newton loop" in err
    assert np.abs(xf - xs).max() > 1e-4                 # (the fast path solved the first step against the old targets)


@needs_exe
def test_sparse_in_place_edit_is_reported_and_the_solve_redone(tmp_path):
    """scene `inplace_sparse`: the callback moves only three of the 12 691 targets, at the 4th evaluation of the run (in the middle of the first
    solve's second line search). The sampled check cannot see that; the full pass at the end of the solve does. A result computed from stale
    inputs is never handed out: the shim says so on stderr, switches this solver to reading everything in full at every evaluation
    (what MISTARK_SHIM_STRICT=1 selects from the start) and REDOES the solve from its initial DoFs on the caller's data as it is now. That is
    the run in which the targets were moved before anything was evaluated (SHIM_EDIT_AT=1, strict): same Newton iterations, same end state.
    MISTARK_SHIM_NO_RESOLVE=1 keeps the stale result and only warns; under MISTARK_SHIM_STRICT=1 nothing is missed and nothing is said."""
    default, err = _run("inplace_sparse", 2, tmp_path, None, "_default")
    assert "WARNING" in err and "REDONE" in err and err.count("WARNING") == 1      # (once: the solver is in strict mode from then on)
    consistent, err_s = _run("inplace_sparse", 2, tmp_path, {"MISTARK_SHIM_STRICT": "1", "SHIM_EDIT_AT": "1"}, "_consistent")
    assert "WARNING" not in err_s
    warn_only, err_w = _run("inplace_sparse", 2, tmp_path, {"MISTARK_SHIM_NO_RESOLVE": "1"}, "_noresolve")
    assert "WARNING" in err_w and "REDONE" not in err_w and "stale result is kept" in err_w
    xd, xs, xw = np.array(default["x"]), np.array(consistent["x"]), np.array(warn_only["x"])
    assert default["newton_iterations"] == consistent["newton_iterations"]
    assert np.abs(xd - xs).max() < 1e-9, np.abs(xd - xs).max()
    # the warn-only run solved its first step against the old targets
    assert np.abs(xw - xs).max() > 1e-4
