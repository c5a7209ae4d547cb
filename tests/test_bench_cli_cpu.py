"""CPU: the launch contract of bench.py that needs no GPU — `python bench.py --gpus N` (no launcher around it) must start N ranks itself or
refuse; it must never run one rank and label the line N (missing #1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MISTARK_BENCH_DEVICE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, timeout=300)


def test_more_ranks_than_visible_gpus_is_refused():
    import torch

    n = torch.cuda.device_count() + 2
    r = _run(["--gpus", str(n), "--no-cpu-baseline"])
    assert r.returncode != 0
    assert b"GPU(s) visible" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _run(["--gpus", "4", "--no-cpu-baseline"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr


def _bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_model_column_is_designs_multi_gpu_table():
    """DESIGN.md "Multi-GPU": 106 CG iterations and 3.6 solves per Newton iteration, 158 k of 1 M elements per rank at 8 ranks ->
    1.7 + 0.4 + 0.34 + 0.4 ms; one GPU 4.1 + 0.4 + 1.5 + 0.8 ms; 2.3x."""
    b = _bench()
    m8 = b.model_ms_per_newton(8, 106.0, 3.6, 0.158)
    m1 = b.model_ms_per_newton(1, 106.0, 3.6, 1.0)
    assert abs(m8["linear_solve"] - 2.11) < 0.02 and abs(m8["evaluation_assembly_projection"] - 0.34) < 0.01 and abs(m8["contact_callbacks"] - 0.45) < 0.01
    assert abs(m1["linear_solve"] - 4.5) < 0.02 and m1["evaluation_assembly_projection"] == 1.5 and m1["contact_callbacks"] == 0.8
    assert 2.2 < m1["total"] / m8["total"] < 2.5
    # a world between the measured points interpolates monotonically
    assert b.model_ms_per_newton(2, 106.0, 3.6, 0.55)["total"] > b.model_ms_per_newton(3, 106.0, 3.6, 0.4)["total"] > b.model_ms_per_newton(4, 106.0, 3.6, 0.3)["total"]


def test_stage_table_and_refused_rccl_leg_have_the_pinned_keys():
    """The N > 1 parts of the JSON line, built from made-up rank reports (no GPU): the stage table beside the model and the RCCL leg of ranks
    that share a device (refused before any RCCL call, keys present and null)."""
    b = _bench()
    per_rank = [{"newton": 6.0, "linear_solve": 3.0 + r, "eval_pgh": 0.5, "eval_p": 0.1, "project": 0.2, "assembly": 0.3, "callback": 0.7, "step": 6.5} for r in range(4)]
    seen = [{"rank": r, "device": 0, "pci_bus_id": "0000:05:00.0", "elements_evaluated": 300_000, "elements_total": 1_000_000} for r in range(4)]
    t = b.stage_table(per_rank, 4, 20, 72, 2100, seen, 7.5)
    assert all(k in t for k in b.STAGE_TABLE_KEYS)
    for col in ("measured", "model", "model_one_gpu"):
        assert all(k in t[col] for k in b.STAGE_KEYS), col
    assert t["measured"]["linear_solve"] == 6.0 and t["measured"]["evaluation_assembly_projection"] == 1.1 and t["measured"]["total"] == 7.5   # slowest rank per stage
    assert t["largest_element_share"] == 0.3 and t["model_speedup"] > 1.0

    called = []
    leg, hung = b.rccl_allreduce_leg(None, None, 0, 4, 0, seen, lambda: called.append("uid"), lambda x: called.append("gather"), 517050)
    assert not hung and not called                      # refused before any collective or library call
    assert all(k in leg for k in b.RCCL_LEG_KEYS) and leg["ranks"] is None and "share a device" in leg["refused"]


def test_line_keys_are_what_the_source_prints():
    """Every key of LINE_KEYS is a top-level key of the dict bench.py prints (source check: the line itself needs a GPU)."""
    b = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("        out = {\n            \"metric\""):src.index("        print(json.dumps(out))")]
    for k in b.LINE_KEYS:
        assert ('            "%s":' % k) in body or ('out["%s"]' % k) in body, k


def test_one_gpu_line_has_a_measured_stage_column_and_value_windows(monkeypatch):
    """N = 1 fills `stages_ms_per_newton_iteration.measured` (the one-GPU column of the scaling table exists as a measurement,
    not only as a model) and carries `value_windows` — window 0 is `value`, further windows on freshly built scenes, and their median."""
    b = _bench()
    stage = {"newton": 0.12, "linear_solve": 0.08, "eval_pgh": 0.012, "eval_p": 0.004, "project": 0.003, "assembly": 0.002, "callback": 0.012, "step": 0.121}
    t = b.stage_table([{k: 1e3 * v / 20 for k, v in stage.items()}], 1, 20, 71, 2100, None, 6.0)
    assert all(k in t for k in b.STAGE_TABLE_KEYS) and all(k in t["measured"] for k in b.STAGE_KEYS)
    assert t["model_speedup"] == 1.0 and t["largest_element_share"] == 1.0 and abs(t["measured"]["linear_solve"] - 4.0) < 1e-9
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "if world == 1:\n        # the one-GPU column of the scaling table, measured" in src

    # value_windows from a stand-in scene (no GPU): three windows, the first is the headline's
    class Info:
        total_newton_iterations = 0
    calls = []

    class Sim:
        def close(self):
            calls.append("close")

    monkeypatch.setattr(b, "build_scene", lambda *a, **k: Sim())
    seq = iter([(20, 70, 2000, 0.07), (5, 1, 1, 0.0), (20, 72, 2100, 0.08)] * 2)
    monkeypatch.setattr(b, "run_newton_steps", lambda sim, S, capi, n: (n, 70, 2000, 0.07))
    import types
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda: None))
    monkeypatch.setitem(sys.modules, "torch", fake_torch)
    w = b.extra_windows(None, None, 44, 44, 43, 0, "contact", (0.0, 0.0), 20, 5, {"value": 160.0, "newton_iterations": 20, "ms_per_linear_solve": 1.13})
    assert all(k in w for k in b.VALUE_WINDOWS_KEYS)
    assert w["windows"] == 1 + b.EXTRA_WINDOWS == len(w["values"]) and w["values"][0] == 160.0 and w["min"] <= w["median"] <= w["max"]
    assert calls == ["close"] * b.EXTRA_WINDOWS
