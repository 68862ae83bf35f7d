"""`python bench.py --gpus N` must start the N ranks itself (the driver's command shape), stay off the GPU in the
parent, relay rank 0's JSON line and exit with the children's return code.  Rehearsed here without a GPU: the
children rendezvous over gloo (MIVIT_BENCH_DRY=1), run one collective and print one line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"MIVIT_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0 and out["steps"] == 3 and out["warmup"] == 1


def test_bench_launcher_propagates_failure():
    # a world size that contradicts --gpus is an error in every rank -> non-zero exit from the parent
    r = _run(["--gpus", "2"], {"MIVIT_BENCH_DRY": "1", "WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0
