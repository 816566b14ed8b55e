"""`python bench.py --gpus N` started plainly (the way the driver runs it) must start its own N ranks: the parent spawns
`python -m torch.distributed.run ... bench.py --gpus N` as a child BEFORE it touches a device and relays output and return code.
Rehearsed here on the CPU with MTTS_BENCH_DRYRUN=1 (ranks rendezvous over gloo, no product code runs)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, extra_env=None):
    env = dict(os.environ, MTTS_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)


def test_plain_launch_spawns_its_ranks():
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                      # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 2, "max_rank": 1, "gpus_arg": 2}


def test_single_process_default_does_not_spawn():
    res = _run([])
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_child_failure_is_relayed():
    res = _run(["--gpus", "2", "--no-such-flag"])
    assert res.returncode != 0
