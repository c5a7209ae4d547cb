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
