"""bench.py --gpus N starts its own ranks: without WORLD_SIZE in the environment the parent process launches torch.distributed.run as a CHILD before it
imports torch, relays the ranks' stdout and exits with the child's code.  --launch-check makes every rank report {rank, world} and leave before any GPU work, so the launch
path runs on a CPU-only box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr.decode(errors="replace")


def test_bench_gpus2_launches_two_ranks_by_itself():
    rc, lines, err = _run(["--gpus", "2", "--launch-check"])
    assert rc == 0, err[-2000:]
    assert sorted((l["rank"], l["world"]) for l in lines) == [(0, 2), (1, 2)], lines


def test_bench_under_torchrun_keeps_working():
    # the driver's own command shape: the ranks already exist (WORLD_SIZE set) -> no second launch
    rc, lines, err = _run(["--gpus", "2", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc == 0, err[-2000:]
    assert [(l["rank"], l["world"]) for l in lines] == [(1, 2)], lines


def test_bench_single_gpu_default_does_not_launch():
    rc, lines, err = _run(["--launch-check"])
    assert rc == 0 and [(l["rank"], l["world"], l["gpus"]) for l in lines] == [(0, 1, 1)], (lines, err[-500:])
