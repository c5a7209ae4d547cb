"""GPU (run with `-m gpu`): the N > 1 launch path of bench.py with REAL OS processes — torch.distributed.run, gloo rendezvous, exchange of the
IPC window handles, mistark_dist_init_ipc, row-sharded evaluation / assembly / PCG — on the ONE MI355X of the test box (all ranks on device
0: MISTARK_BENCH_DEVICE; RCCL refuses two ranks on one device, the IPC-window transport does not care). The sharded run must take the
single-rank run's decisions: the same Newton iterations, linear solves (progressive-projection retries included) and contact counts, and the
same CG iterations up to the +-1 per solve the parity rule allows (the reductions are solve_pcg.h:180,201,217)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRID = "12,12,12"
ARGS = ["--steps", "8", "--warmup", "2", "--grid", GRID, "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env["MISTARK_BENCH_DEVICE"] = "0"
    env["MISTARK_IPC_TIMEOUT_S"] = "20"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MISTARK_BENCH_TRANSPORT", None)
    return env


def _torchrun(n, script, args, extra_env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, script)] + args
    env = _env()
    env.update(extra_env or {})
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, timeout=900)


def _launch(n, script, args):
    r = _torchrun(n, script, args)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def _worker(n, args, extra_env=None):
    r = _torchrun(n, "tests/mp_sharded_worker.py", args, extra_env)
    out = r.stdout.decode()
    assert r.returncode == 0 and "MP-OK" in out, (out[-2000:], r.stderr.decode()[-3000:])
    return out


@pytest.fixture(scope="module")
def single():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT, env=_env(), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("n", [2, 4])
def test_transport_selftest_between_processes(n):
    out = _launch(n, "tools/ipc_selftest.py", [])
    assert out["world"] == n and out["one_device"]
    assert all(a > 0 and b > 0 for a, b in out["allgather_us_by_doubles [synchronised, in a train]"].values())


@pytest.mark.parametrize("n", [2, 4])
def test_bench_launch_path_with_n_processes_takes_the_single_rank_decisions(single, n):
    out = _launch(n, "bench.py", ["--gpus", str(n)] + ARGS)
    assert out["n_gpus"] == n and out["config"]["transport"] == "ipc" and out["config"]["ranks_on_one_device"]
    assert out["newton_iterations"] == single["newton_iterations"] == 8
    assert out["linear_solves"] == single["linear_solves"]
    assert abs(out["cg_iterations"] - single["cg_iterations"]) <= single["linear_solves"]
    for k in ("n_contacts", "n_friction_contacts", "n_detections"):
        assert out["contact"][k] == single["contact"][k], k
    assert out["roofline"]["achieved"] > 0


@pytest.mark.parametrize("n", [2, 4])
def test_plain_bench_command_launches_its_own_ranks(single, n):
    """`python bench.py --gpus N` with NO launcher around it (what the round-end driver runs): bench.py starts the N ranks itself and rank 0
    prints the one line; the line says how many ranks really ran, on which devices, over which transport."""
    env = _env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + ARGS, cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and "bench.py itself" in out["config"]["launched_by"]
    seen = out["config"]["ranks_seen"]
    assert [s["rank"] for s in seen] == list(range(n)) and len({s["pid"] for s in seen}) == n
    assert all(s["engine_world"] == n and s["engine_rank"] == s["rank"] and s["transport"] == "ipc" and s["transport_ranks"] == n for s in seen)
    assert out["config"]["distinct_devices"] == 1 and out["config"]["ranks_on_one_device"]   # (this box has one GPU; the line says so)
    # every search's sweep was dealt out to the ranks, every solve took the fused iteration: and the counts below are the one-rank run's
    assert all(s["contact_searches_with_the_sweep_dealt_out"] > 0 and s["linear_solves_five_launch_iteration"] == 0 for s in seen)
    assert sum(s["rows_owned"] for s in seen) == 13 ** 3 + 12 ** 3 + 2   # every block row has exactly one owner (nodes + hexahedron centres + the box's v, w)
    assert out["newton_iterations"] == single["newton_iterations"] == 8
    assert out["linear_solves"] == single["linear_solves"]
    assert abs(out["cg_iterations"] - single["cg_iterations"]) <= single["linear_solves"]
    # the N > 1 parts of the line: the windows' pre-flight with its peer-latency matrix, the RCCL leg (refused here: the ranks
    # share the one device, and the line says so instead of skipping it silently), measured stage times beside DESIGN.md's model
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert all(k in out for k in b.LINE_KEYS), [k for k in b.LINE_KEYS if k not in out]
    pf = out["preflight"]
    assert all(k in pf for k in b.PREFLIGHT_KEYS) and pf["ok"] and pf["timeout_s"] <= 2.0 and pf["wall_s"] < 10.0
    lat = out["peer_latency_us"]
    assert len(lat) == n and all(len(row) == n for row in lat)
    assert all(lat[a][a] == 0.0 and all(0.0 < lat[a][c] < 5e4 for c in range(n) if c != a) for a in range(n)), lat
    assert out["config"]["transport"] == "ipc" and out["config"]["transport_fallback_reason"] is None
    assert all(k in out["rccl"] for k in b.RCCL_LEG_KEYS) and out["rccl"]["ranks"] is None and "share a device" in out["rccl"]["refused"]
    st = out["stages_ms_per_newton_iteration"]
    assert all(k in st for k in b.STAGE_TABLE_KEYS) and all(k in st[c] for c in ("measured", "model", "model_one_gpu") for k in b.STAGE_KEYS)
    assert st["measured"]["total"] == pytest.approx(out["ms_per_step"], rel=1e-2) and 0 < st["measured"]["linear_solve"] < st["measured"]["total"]
    assert 1.0 / n <= st["largest_element_share"] < 1.0


def test_preflight_that_times_out_falls_back_and_says_so():
    """A window that never delivers (MISTARK_IPC_FAULT=drop_preflight: rank 1's pre-flight stores go nowhere) must cost the run no more than the
    pre-flight's 2 s time-out and be REPORTED: every rank takes RCCL — which refuses ranks sharing a device, so on this box the run ends with that
    error rather than with numbers from a transport that does not work."""
    env = _env()
    env["MISTARK_IPC_FAULT"] = "drop_preflight"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = __import__("time").perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, cwd=ROOT, env=env, capture_output=True, timeout=600)
    err = r.stderr.decode()
    assert "falling back to RCCL" in err and "pre-flight" in err, err[-3000:]
    assert r.returncode != 0 and ("Duplicate GPU" in err or "RCCL" in err), err[-3000:]   # (one device: RCCL refuses; on N devices the run continues on it)
    assert __import__("time").perf_counter() - t0 < 240


def test_plain_bench_command_refuses_more_ranks_than_gpus():
    """without MISTARK_BENCH_DEVICE a box with fewer GPUs than --gpus is an error, not a one-rank run labelled N"""
    env = _env()
    env.pop("MISTARK_BENCH_DEVICE")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # (the count comes from a child process: importing torch HERE would load its bundled ROCm libraries into the test process, and the RCCL
    # round trip of tests/test_gpu_sharded.py, which dlopens librccl later in the same process, then fails)
    n = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, timeout=300).stdout.decode().strip().splitlines()[-1]) + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + ARGS, cwd=ROOT, env=env, capture_output=True, timeout=300)
    assert r.returncode != 0 and b"GPU(s) visible" in r.stderr and not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


# ---- the cases of tests/test_gpu_sharded.py between real processes: the fused PCG iteration (kernels exchanging through the windows) -------
@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("name", ["tetbeam_full_4x1x1", "tetbeam_eo_4x1x1_big", "cloth_shells_6", "contactmix_t1", "rbchain"])
def test_stages_between_processes_equal_single_rank(name, n):
    _worker(n, ["stages", name])


def test_stages_between_processes_with_the_unfused_iteration_too():
    """option no_fused_pcg: the five-launch iteration with two all-gathers through the windows' general region"""
    _worker(2, ["stages", "contactmix_t1"], {"MP_ENGINE_OPTS": "no_fused_pcg=1"})


@pytest.mark.parametrize("n", [2, 4])
def test_newton_between_processes(n):
    _worker(n, ["newton"])


@pytest.mark.parametrize("n", [3, 8])
def test_contact_scene_between_processes(n):
    """2 331 block rows, recursive coordinate bisection, frictional contact against a rigid box: Newton counts per step (+-1) and end state
    against one rank; identical bits on all ranks"""
    _worker(n, ["scene"])


def test_full_size_headline_scene_between_processes():
    """configs[3] at full size (998 976 tets) on 4 processes sharing the one GPU"""
    _worker(4, ["fullsize"])
