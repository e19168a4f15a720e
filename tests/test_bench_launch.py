"""`python bench.py --gpus N` must start N ranks by itself (child torchrun, no exec) and they must
rendezvous; the N=1 command shape the driver uses must never silently run one rank for N>1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_gpus_2_self_launches_two_ranks_that_rendezvous():
    r = _run(["--gpus", "2", "--dry-launch"], {"BBD_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["ranks"] == 2 and out["n_gpus"] == 2 and out["all_reduce_ok"] is True
    assert out["collective"] == "gloo"        # RCCL ("rccl") on the GPU box, gloo only under the test override


def test_single_rank_dry_launch_is_inline():
    r = _run(["--gpus", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["ranks"] == 1 and out["collective"] == "none"


def test_world_size_mismatch_is_refused():
    # a torchrun-style environment whose world size disagrees with --gpus must fail, not print an N=1 line
    r = _run(["--gpus", "4", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "world size" in (r.stderr + r.stdout)
